// ref_noise_shim.cpp -- extern "C" doors onto the REFERENCE's own gaussian_noise / probit_noise objects.
//
// Test infrastructure only.  Compiled by oracle/Makefile together with
// /root/reference/src/gaussian_noise.cpp and /root/reference/src/probit_noise.cpp (the only two files of the
// hot path that build without Eigen/PCL) into oracle/_ref/libref_noise.so.  No reference source
// is copied into this repository: the headers are included from where they lie.
#include "gaussian_noise.h"   // -I/root/reference/src
#include "probit_noise.h"

extern "C" {
double ref_gaussian_dx_ln(double s20, double y, double x, double sigma_x) { gaussian_noise n(s20); return n.dx_ln(y, x, sigma_x); }
double ref_gaussian_dx2_ln(double s20, double y, double x, double sigma_x) { gaussian_noise n(s20); return n.dx2_ln(y, x, sigma_x); }
double ref_probit_dx_ln(double s20, double y, double x, double sigma_x) { probit_noise n(s20); return n.dx_ln(y, x, sigma_x); }
double ref_probit_dx2_ln(double s20, double y, double x, double sigma_x) { probit_noise n(s20); return n.dx2_ln(y, x, sigma_x); }
}
