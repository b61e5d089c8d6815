// gp_compressor.hpp -- host-side C++ mirror of the reference's class surface for the hot path.
//
// Same names, argument meaning and flow as /root/reference/src/gp_compressor.{h,cpp}: a cloud goes in, octree-leaf
// patches are cut and projected into their plane frames on the host (project_cloud / compute_rotation /
// project_points, src/gp_compressor.cpp:29-118,177-249), the per-patch GP loops (train_processes :121-175 and the
// patch loop of load_compressed :298-380) run BATCHED on the GPU through the C-ABI of include/gpc.h, and the
// predicted grids are re-projected to a coloured cloud (:335-373).  PCL and Eigen are not used: the spatial index is
// a plain voxel hash of side `res` (enough to feed the hot path; the octree itself is out of scope, SURVEY section 2
// row 9) and the 4x4 plane fit is a Jacobi eigen-solve.
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/gpc.h"

namespace gpc {

// pcl::PointXYZRGB, the fields the path touches (src/gp_compressor.h:20)
struct point {
    float x, y, z;
    uint8_t r, g, b;
};
using pointcloud = std::vector<point>;

// Duck-typed functors of the reference, host versions (what the kernels evaluate on the device).
class rbf_kernel {   // src/rbf_kernel.h:7-25
    double p_[2];
public:
    explicit rbf_kernel(double sigmaf_sq = 100e-0f, double l_sq = 1 * 1) : p_{sigmaf_sq, l_sq} {}
    int param_size() const { return 2; }
    const double* param() const { return p_; }
    double kernel_function(const double xi[2], const double xj[2]) const;   // src/rbf_kernel.cpp:15-18
};
class gaussian_noise {   // src/gaussian_noise.h:4-11
public:
    double s20;
    explicit gaussian_noise(double s20_) : s20(s20_) {}
    double dx_ln(double y, double x, double sigma_x) const { return (y - x) / (s20 + sigma_x); }
    double dx2_ln(double, double, double sigma_x) const { return -1.0f / (s20 + sigma_x); }
};

enum class gp_model {
    sparse,   // sparse_gp<rbf_kernel, gaussian_noise> + sparse_gp_field<rbf_kernel, gaussian_noise_3d>: what the reference runs
    dense     // gaussian_process (src/gaussian_process.h), the batched-Cholesky model of the north star
};

// The patch batch project_cloud() hands to the GPU: exactly the X, y, C of src/gp_compressor.cpp:146-155, batched.
struct patch_batch {
    std::vector<int32_t> off;            // P + 1
    std::vector<double> x0, x1;          // patch-frame coordinates pt(1), pt(2)
    std::vector<double> y;               // depth pt(0), mean-removed
    std::vector<double> rgb;             // 3 planes of N, mean-removed colours
    std::vector<std::array<double, 9>> rotations;   // R_i, column-major (columns = normal, u, v)
    std::vector<std::array<double, 3>> means;       // patch centres after the mean-depth shift (:116)
    std::vector<std::array<double, 3>> rgb_means;
    std::vector<uint8_t> W;              // sz*sz occupancy mask per patch (:117)
    int patches() const { return (int)off.size() - 1; }
};

class gp_compressor {
public:
    // gp_compressor(pointcloud::ConstPtr ncloud, double res = 0.1f, int sz = 10)   src/gp_compressor.h:65
    gp_compressor(const pointcloud& ncloud, double res = 0.1f, int sz = 10, gp_model model = gp_model::sparse,
                  int device = 0);
    ~gp_compressor();
    gp_compressor(const gp_compressor&) = delete;
    gp_compressor& operator=(const gp_compressor&) = delete;

    void save_compressed(const std::string& name);   // src/gp_compressor.cpp:21-27 (name unused upstream as well)
    pointcloud load_compressed();                    // src/gp_compressor.cpp:267-386

    // The wire format the reference never wrote (save_compressed ignores `name`): per patch the
    // frame (R_i, mean_i, RGB mean) and the two sparse GPs as (BV, alpha) -- all the decompressor's mean prediction needs.
    // C and Q are not stored: a loaded model reconstructs the same cloud bit for bit, but cannot be trained further.
    // Sparse model only.  Returns the number of bytes written.
    size_t save_model(const std::string& path);
    static gp_compressor* load_model(const std::string& path, int device = 0);

    // host-only part, usable without a GPU
    void project_cloud();                            // src/gp_compressor.cpp:177-249
    // the same batch, bit for bit, cut on the GPU (gpc_project_cloud); what save_compressed()
    // uses unless gpu_producer is cleared
    void project_cloud_device();
    bool gpu_producer = true;
    const patch_batch& patches() const { return batch_; }
    // statistics the reference prints ("Mean added" / "Max added", :173-174)
    double mean_added() const { return mean_added_; }
    int max_added() const { return max_added_; }
    // Multi-GPU: the reference is one process, so one process drives the GPUs of the node -- one gpc_ctx per
    // device, gpc_partition_patches (longest-processing-time) deals the patches, every device fits + predicts its slots and ONE
    // RCCL all-gather (gpc_comm_create_all + gpc_group bracket) reassembles the grids.  Host-cut patches (save_compressed() then
    // runs project_cloud() on the host); an empty list returns to the single-device flow.  Dense model: fit + predict per device
    // at training time.  Sparse model (what the reference runs): the partition is drawn with the sparse cost model and the two GPs
    // of a patch live on ITS device from train_processes() to load_compressed() (fixed affinity: the state never moves), which
    // predicts per device and gathers the grids once.
    void set_devices(const std::vector<int>& devices);
    // insertion-order source; default std::rand like sparse_gp::shuffle (src/sparse_gp.hpp:43-56)
    std::function<int()> rng;
    // hyper-parameters (defaults = the reference's compile-time constants)
    gpc_params depth_params, rgb_params, dense_params;

protected:
    void compute_rotation(double R[9], const std::vector<double>& pts4, int k) const;   // :29-64
    void train_processes();                                                            // :121-175
    static void flatten_colors(uint8_t out[3], const double c[3]);                      // :251-265
    void shuffle(std::vector<int32_t>& perm, int n);

    gp_compressor(double res, int sz, int device);   // empty shell for load_model
    pointcloud cloud_;
    double res_;
    int sz_;
    gp_model model_;
    int device_;
    patch_batch batch_;
    bool projected_ = false, trained_ = false;
    gpc_ctx* ctx_ = nullptr;
    gpc_sparse* gps_ = nullptr;
    gpc_sparse* rgb_gps_ = nullptr;
    std::vector<double> dense_f_, dense_c_;   // dense model: grids predicted at training time (fit + predict are fused)
    // device-resident flow (project_cloud_device): the batch stays in HBM, the GP kernels and the reprojection read it
    // there, and only `off`, the insertion orders and the final cloud cross PCIe
    gpc_patches* dev_patches_ = nullptr;
    double *d_dense_f_ = nullptr, *d_dense_c_ = nullptr;
    void release_device();
    void train_dense_sharded();
    void init_shards();
    void train_sparse_sharded(const std::vector<int32_t>& perm_d, const std::vector<int32_t>& perm_c);
    void predict_sparse_sharded(const std::vector<double>& xs0, const std::vector<double>& xs1, std::vector<double>& f_star,
                                std::vector<double>& c_star, std::vector<int32_t>& bv);
    std::vector<gpc_sparse*> shard_gps_, shard_rgb_;   // sparse model, sharded: the GPs of a device's slots
    std::vector<int32_t> shard_slots_;                 // world * S: slot -> patch (-1 padding), from gpc_partition_patches
    std::vector<int> devices_;                // set_devices(): non-empty = the sharded flow
    std::vector<gpc_ctx*> shard_ctx_;         // one per device (shard_ctx_[0] is its own context, not ctx_)
    std::vector<gpc_comm*> shard_comm_;
    std::vector<int32_t> status_;
    double mean_added_ = 0.0;
    int max_added_ = 0;
};

}  // namespace gpc
