// svnicp_hip_shim.hpp — header-only C++ mirror of the reference solver classes over the C ABI.
//
// Same class and method names, argument meaning and call order as svnicp::SVGDICP / svnicp::SVNICP
// (/root/reference/svn-icp/include/core/SVGDICP.h:64-110, SVNICP.h:29-43) as used by
// OdometryPipeline::ICP_processing (src/core/OdometryPipeline.cpp:573-607), with raw buffers in
// place of torch::Tensor and a row-major 3x3 + translation in place of gtsam::Pose3.  A ROS2 node
// keeps its call sequence and links libsvnicp_hip.so instead of libtorch.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "svnicp_hip.h"

namespace svnicp {

enum SteinICPState { ALIGN_SUCCESS = 1, NO_OPTIMIZER = 2 };  // SVGDICP.h:59-62

struct SteinICPParam {  // SVGDICP.h:41-57 (same names and defaults; solver-relevant fields)
  int iterations = 50;
  double lr = 0.02;
  double max_dist = 1.0;
  std::string optimizer = "Adam";
  bool check_early_stop = false;
  double convergence_threshold = 1e-5;
  int KNN_count = 100;
  bool SVN_full_grad = true;
};

struct ParticleWeightOpt { bool use_weight_mean = false; };  // SVNICP.h:25-27

class SVGDICP {
 public:
  // init_pose: [6][P] row-major (x.., y.., z.., rx.., ry.., rz..) — the reference's [6,P,1] tensor
  SVGDICP(const SteinICPParam& p, const std::vector<double>& init_pose, int device = 0, int mode = SVNICP_MODE_SVGD)
      : P_((int)(init_pose.size() / 6)), I_(p.iterations) {
    svnicp_params q{};
    q.struct_size = (int32_t)sizeof q;
    q.mode = mode;
    q.iterations = p.iterations; q.knn_count = p.KNN_count; q.lr = p.lr; q.max_dist = p.max_dist;
    q.convergence_threshold = p.convergence_threshold; q.check_early_stop = p.check_early_stop;
    q.svn_full_grad = p.SVN_full_grad;
    q.optimizer = p.optimizer == "Adam" ? SVNICP_OPT_ADAM : p.optimizer == "RMSprop" ? SVNICP_OPT_RMSPROP
                : p.optimizer == "SGD" ? SVNICP_OPT_SGD : p.optimizer == "Adagrad" ? SVNICP_OPT_ADAGRAD : SVNICP_OPT_NONE;
    if (svnicp_create(&q, device, init_pose.data(), P_, &h_) != 0) throw std::runtime_error(svnicp_last_error(nullptr));
  }
  virtual ~SVGDICP() { svnicp_destroy(h_); }
  SVGDICP(const SVGDICP&) = delete;
  SVGDICP& operator=(const SVGDICP&) = delete;

  // add_cloud(source[B,3], target[M,3], init_pose[6,P,1]) — SVGDICP.cpp:46-62
  void add_cloud(const double* source_xyz, int64_t B, const double* target_xyz, int64_t M, const double* init_pose6xP, int P) {
    chk(svnicp_set_clouds(h_, source_xyz, B, target_xyz, M, SVNICP_MEM_HOST));
    chk(svnicp_set_particles(h_, init_pose6xP, P));
    P_ = P;
  }
  // add_cloud with a host source scan and a target already in HBM (svnicp_map_query): only the source crosses PCIe
  void add_cloud_device_target(const double* source_xyz, int64_t B, const double* target_dev_xyz, int64_t M, const double* init_pose6xP, int P) {
    chk(svnicp_set_source(h_, source_xyz, B, SVNICP_MEM_HOST));
    chk(svnicp_set_target(h_, target_dev_xyz, M, SVNICP_MEM_DEVICE));
    chk(svnicp_synchronize(h_));
    chk(svnicp_set_particles(h_, init_pose6xP, P));
    P_ = P;
  }
  // add_cloud with both clouds already in HBM (float64 rows): two device-to-device copies
  void add_cloud_device(const double* source_dev_xyz, int64_t B, const double* target_dev_xyz, int64_t M, const double* init_pose6xP, int P) {
    chk(svnicp_set_source(h_, source_dev_xyz, B, SVNICP_MEM_DEVICE));
    chk(svnicp_set_target(h_, target_dev_xyz, M, SVNICP_MEM_DEVICE));
    chk(svnicp_synchronize(h_));
    chk(svnicp_set_particles(h_, init_pose6xP, P));
    P_ = P;
  }
  // set_initial_mean(gtsam::Pose3) — SVGDICP.h:102-110
  void set_initial_mean(const double R_rowmajor[9], const double t[3]) { chk(svnicp_set_initial_mean(h_, R_rowmajor, t)); }
  virtual SteinICPState stein_align() {  // SVNICP.cpp:41-114 / SVGDICP.cpp:66-140
    const int r = svnicp_align(h_);
    if (r < 0) fail();
    return (SteinICPState)r;
  }
  std::array<double, 6> get_transformation() { std::array<double, 6> o{}; chk(svnicp_get_transformation(h_, o.data())); return o; }
  std::array<double, 6> get_distribution() { std::array<double, 6> o{}; chk(svnicp_get_distribution(h_, o.data())); return o; }
  std::vector<double> get_cov_matrix() { std::vector<double> o(36); chk(svnicp_get_cov_matrix(h_, o.data())); return o; }
  std::vector<double> get_particles() { std::vector<double> o((size_t)6 * P_); chk(svnicp_get_particles(h_, o.data())); return o; }
  std::vector<double> get_particle_weight() { std::vector<double> o((size_t)P_); chk(svnicp_get_particle_weight(h_, o.data())); return o; }
  std::vector<std::vector<float>> get_particle_history() {
    std::vector<float> flat((size_t)I_ * 6 * P_);
    chk(svnicp_get_particle_history(h_, flat.data()));
    std::vector<std::vector<float>> o;
    for (int i = 0; i < I_; ++i) o.emplace_back(flat.begin() + (size_t)i * 6 * P_, flat.begin() + (size_t)(i + 1) * 6 * P_);
    return o;
  }
  std::vector<double> get_runtime() { std::vector<double> o(3); chk(svnicp_get_runtime(h_, o.data())); return o; }
  void set_k(int k) { chk(svnicp_set_k(h_, k)); }
  void set_threshold(double max_dist) { chk(svnicp_set_max_dist(h_, max_dist)); }
  svnicp_ctx* handle() { return h_; }

 protected:
  void chk(int rc) { if (rc != 0) fail(); }
  [[noreturn]] void fail() { throw std::runtime_error(svnicp_last_error(h_)); }
  svnicp_ctx* h_ = nullptr;
  int P_, I_;
};

class SVNICP final : public SVGDICP {
 public:
  SVNICP(const SteinICPParam& p, const std::vector<double>& init_pose, const ParticleWeightOpt& = {}, int device = 0)
      : SVGDICP(p, init_pose, device, SVNICP_MODE_SVN) {}
};

}  // namespace svnicp
