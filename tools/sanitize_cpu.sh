#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU ASan is not available on this pool): the oracle (test infrastructure)
# and the C++ host producer, driven by their CPU tests through LD_PRELOAD.  Restores the normal builds afterwards.
set -eu
cd "$(dirname "$0")/.."
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared"
PRE="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
restore() { rm -f oracle/liboracle.so; make -s -C oracle liboracle.so; rm -f gp_compressor_amd/libgpc_host.so; make -s -C gp_compressor_amd/host; }
trap restore EXIT
gcc -std=c99 -D_GNU_SOURCE $SAN -o oracle/liboracle.so oracle/gpc_oracle.c oracle/gpc_oracle_producer.c -lm
g++ -std=c++17 $SAN -o gp_compressor_amd/libgpc_host.so gp_compressor_amd/host/gp_compressor.cpp -Lgp_compressor_amd -lgpc_hip \
    -Wl,-rpath,"$PWD/gp_compressor_amd"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$PRE" python -m pytest tests/test_oracle.py tests/test_host_cpu.py -q -p no:cacheprovider
