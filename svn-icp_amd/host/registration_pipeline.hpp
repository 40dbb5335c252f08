// registration_pipeline.hpp — header-only, ROS-free C++ restatement of the scan-to-map sequence that calls the solver.
//
// SURVEY.md §8(f)-1.  Mirrors OdometryPipeline::ICP_processing of the reference for the `estimator == ICP`
// configuration (/root/reference/svn-icp/src/core/OdometryPipeline.cpp:556-647): crop (:692-704) -> uniform
// down-sample, map cloud at 0.5·voxel and solver cloud at 1.5·voxel of that (:559-560, :684-690) -> constant-twist pose
// prediction (:706-737) -> particle prior (:661-667) -> local-map range query (:577-581, VoxelHashMap.cpp:48-58) ->
// solver (svnicp_hip_shim.hpp, :582-607) -> pose = prediction · correction (updater_, :37-46) -> map insert (:627,
// VoxelHashMap.cpp:22-42).  The reference does this with PCL / GTSAM / tsl::robin_map on the host; none of them exists
// in this image, so the few operations needed are written out here (float32 points like pcl::PointXYZ, float64 poses).
// The Python module svn-icp_amd/pipeline.py is the same sequence and serves as the cross-check in the tests.
//
// Where PCL / the hash map leave an order unspecified (iteration order of occupied leaves / voxels) this code emits
// ascending leaf / voxel index; the solver's result does not depend on the order of the target points except through
// exact ties.  Parity of the solver call itself is pinned by the tests against the CPU oracle on the very inputs this
// class hands to the solver (`Tap`).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <vector>

#include "svnicp_hip_shim.hpp"

namespace svnicp {

// ------------------------------------------------------------------------------------------------ SE(3), gtsam::Pose3 semantics
struct Pose3 {
  std::array<double, 9> R{1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major
  std::array<double, 3> t{0, 0, 0};

  Pose3 operator*(const Pose3& o) const {
    Pose3 r;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) r.R[3 * i + j] = R[3 * i] * o.R[j] + R[3 * i + 1] * o.R[3 + j] + R[3 * i + 2] * o.R[6 + j];
      r.t[i] = R[3 * i] * o.t[0] + R[3 * i + 1] * o.t[1] + R[3 * i + 2] * o.t[2] + t[i];
    }
    return r;
  }
  Pose3 inverse() const {
    Pose3 r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r.R[3 * i + j] = R[3 * j + i];
    for (int i = 0; i < 3; ++i) r.t[i] = -(r.R[3 * i] * t[0] + r.R[3 * i + 1] * t[1] + r.R[3 * i + 2] * t[2]);
    return r;
  }
};

namespace detail {
inline std::array<double, 9> hat(const double w[3]) { return {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}; }
inline std::array<double, 9> mul3(const std::array<double, 9>& a, const std::array<double, 9>& b) {
  std::array<double, 9> c{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  return c;
}
inline std::array<double, 9> lin3(double a, double b, const std::array<double, 9>& K, double c, const std::array<double, 9>& K2) {
  std::array<double, 9> r{};
  for (int i = 0; i < 9; ++i) r[i] = b * K[i] + c * K2[i];
  r[0] += a; r[4] += a; r[8] += a;
  return r;
}
}  // namespace detail

inline std::array<double, 9> so3_exp(const double w[3]) {  // Rot3::Expmap
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const auto K = detail::hat(w);
  const auto K2 = detail::mul3(K, K);
  if (th < 1e-10) return detail::lin3(1.0, 1.0, K, 0.5, K2);
  return detail::lin3(1.0, std::sin(th) / th, K, (1.0 - std::cos(th)) / (th * th), K2);
}
inline std::array<double, 3> so3_log(const std::array<double, 9>& R) {  // Rot3::Logmap
  const double c = std::max(-1.0, std::min(1.0, 0.5 * (R[0] + R[4] + R[8] - 1.0)));
  const double th = std::acos(c);
  const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const double f = th < 1e-10 ? 0.5 : th / (2.0 * std::sin(th));
  return {f * v[0], f * v[1], f * v[2]};
}
inline Pose3 se3_exp(const double xi[6]) {  // Pose3::Expmap, xi = [omega, v]
  const double* w = xi;
  const double* v = xi + 3;
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const auto K = detail::hat(w);
  const auto K2 = detail::mul3(K, K);
  const auto V = th < 1e-10 ? detail::lin3(1.0, 0.5, K, 0.0, K2)
                            : detail::lin3(1.0, (1.0 - std::cos(th)) / (th * th), K, (th - std::sin(th)) / (th * th * th), K2);
  Pose3 T;
  T.R = so3_exp(w);
  for (int i = 0; i < 3; ++i) T.t[i] = V[3 * i] * v[0] + V[3 * i + 1] * v[1] + V[3 * i + 2] * v[2];
  return T;
}
inline std::array<double, 6> se3_log(const Pose3& T) {  // Pose3::Logmap -> [omega, v]
  const auto w = so3_log(T.R);
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const auto K = detail::hat(w.data());
  const auto K2 = detail::mul3(K, K);
  const auto Vi = th < 1e-10 ? detail::lin3(1.0, -0.5, K, 0.0, K2)
                             : detail::lin3(1.0, -0.5, K, 1.0 / (th * th) - (1.0 + std::cos(th)) / (2.0 * th * std::sin(th)), K2);
  std::array<double, 6> xi{w[0], w[1], w[2], 0, 0, 0};
  for (int i = 0; i < 3; ++i) xi[3 + i] = Vi[3 * i] * T.t[0] + Vi[3 * i + 1] * T.t[1] + Vi[3 * i + 2] * T.t[2];
  return xi;
}
// svnicp::tensor2gtsamPose3 (src/core/ICPUtils.cpp:84-98): [x,y,z,rx,ry,rz] -> Pose3(Rot3::Expmap(r), t)
inline Pose3 correction_to_pose(const std::array<double, 6>& x) {
  Pose3 T;
  T.R = so3_exp(x.data() + 3);
  T.t = {x[0], x[1], x[2]};
  return T;
}

// ------------------------------------------------------------------------------------------------ clouds (pcl::PointXYZ = 3 x float32)
using Cloud = std::vector<std::array<float, 3>>;

// OdometryPipeline::crop_pointcloud (:692-704): keep min_range² < |p|² < max_range²; scan_max_range_ is updated to the
// largest SQUARED norm seen (:699) and is later used as a length (:578) — mirrored as is
inline Cloud crop_pointcloud(const Cloud& in, double min_range, double max_range, double* scan_max_range) {
  Cloud out;
  out.reserve(in.size());
  for (const auto& p : in) {
    // float32, left to right, as pt.x * pt.x + pt.y * pt.y + pt.z * pt.z on pcl::PointXYZ (:698); volatile keeps a host
    // compiler from contracting the sums into fused multiply-adds
    volatile float xx = p[0] * p[0], yy = p[1] * p[1], zz = p[2] * p[2];
    volatile float xy = xx + yy;
    const double n2 = (double)(float)(xy + zz);
    if (n2 > *scan_max_range) *scan_max_range = n2;
    if (n2 < max_range * max_range && n2 > min_range * min_range) out.push_back(p);
  }
  return out;
}

// pcl::UniformSampling with setRadiusSearch(radius) (:684-690): grid of leaf size `radius` anchored at floor(min/leaf);
// per occupied leaf the point closest to the leaf centre survives (first one wins ties); leaves in ascending index
inline Cloud downsample_uniform(const Cloud& in, double radius) {
  if (in.empty() || radius <= 0) return in;
  const double inv = 1.0 / radius;
  int64_t mn[3], mx[3];
  for (int d = 0; d < 3; ++d) { mn[d] = INT64_MAX; mx[d] = INT64_MIN; }
  for (const auto& p : in)
    for (int d = 0; d < 3; ++d) {
      const int64_t c = (int64_t)std::floor((double)p[d] * inv);
      mn[d] = std::min(mn[d], c); mx[d] = std::max(mx[d], c);
    }
  const int64_t dx = mx[0] - mn[0] + 1, dy = mx[1] - mn[1] + 1;
  struct Rec { int64_t leaf; double d2; size_t idx; };
  std::vector<Rec> recs(in.size());
  for (size_t i = 0; i < in.size(); ++i) {
    int64_t ijk[3];
    double d2 = 0;
    for (int d = 0; d < 3; ++d) {
      ijk[d] = (int64_t)std::floor((double)in[i][d] * inv) - mn[d];
      const double c = ((double)(ijk[d] + mn[d]) + 0.5) * radius;
      d2 += ((double)in[i][d] - c) * ((double)in[i][d] - c);
    }
    recs[i] = {ijk[0] + ijk[1] * dx + ijk[2] * dx * dy, d2, i};
  }
  std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {
    if (a.leaf != b.leaf) return a.leaf < b.leaf;
    if (a.d2 != b.d2) return a.d2 < b.d2;
    return a.idx < b.idx;
  });
  Cloud out;
  for (size_t i = 0; i < recs.size(); ++i)
    if (i == 0 || recs[i].leaf != recs[i - 1].leaf) out.push_back(in[recs[i].idx]);
  return out;
}

// ------------------------------------------------------------------------------------------------ local map
// svnicp::VoxelHashMap (src/core/VoxelHashMap.cpp:22-101): voxel -> at most max_points points in insertion order; voxel index
// = coordinates / voxel_size truncated TOWARD ZERO (Eigen cast<int>, :29); a voxel is dropped when its FIRST point is
// farther than max_range from the current position (:89-97) and selected by GetMap(pose, r) when that point is closer
// than r (:48-58).  An ordered map stands in for tsl::robin_map: deterministic iteration (ascending voxel index).
class VoxelHashMap {
 public:
  VoxelHashMap(double voxel_size, double max_range, int max_points) : voxel_size_(voxel_size), max_range_(max_range), max_points_(max_points) {}
  bool Empty() const { return map_.empty(); }
  size_t Size() const { return map_.size(); }
  void Clear() { map_.clear(); }

  void AddPointCloud(const Cloud& cloud, const Pose3& pose) {
    // pcl::transformPointCloud with gtsam's double Matrix4 (VoxelHashMap.cpp:23-25): formed in double from the widened float32
    // point, left to right, rounded once to float32 (the expression of voxel_map.hip and pipeline.py)
    const float vs = (float)voxel_size_;
    for (const auto& p : cloud) {
      std::array<float, 3> q;
      for (int i = 0; i < 3; ++i)
        q[i] = (float)(((pose.R[3 * i] * (double)p[0] + pose.R[3 * i + 1] * (double)p[1]) + pose.R[3 * i + 2] * (double)p[2]) + pose.t[i]);
      const Key k{(int64_t)std::trunc(q[0] / vs), (int64_t)std::trunc(q[1] / vs), (int64_t)std::trunc(q[2] / vs)};
      auto& v = map_[k];
      if ((int)v.size() < max_points_) v.push_back(q);
    }
    RemoveFarPointCloud(pose.t);
  }
  Cloud GetMap() const {
    Cloud out;
    for (const auto& kv : map_) out.insert(out.end(), kv.second.begin(), kv.second.end());
    return out;
  }
  Cloud GetMap(const Pose3& pose, double max_range) const {
    Cloud out;
    for (const auto& kv : map_)
      if (dist2(kv.second.front(), pose.t) < max_range * max_range) out.insert(out.end(), kv.second.begin(), kv.second.end());
    return out;
  }

 private:
  using Key = std::array<int64_t, 3>;
  static double dist2(const std::array<float, 3>& p, const std::array<double, 3>& c) {
    double s = 0;
    for (int i = 0; i < 3; ++i) s += ((double)p[i] - c[i]) * ((double)p[i] - c[i]);
    return s;
  }
  void RemoveFarPointCloud(const std::array<double, 3>& pos) {
    for (auto it = map_.begin(); it != map_.end();)
      it = dist2(it->second.front(), pos) > max_range_ * max_range_ ? map_.erase(it) : std::next(it);
  }
  double voxel_size_, max_range_;
  int max_points_;
  std::map<Key, Cloud> map_;
};

// The same map resident in HBM (svnicp_map_* of the C ABI, csrc/voxel_map.hip): AddPointCloud uploads only the new points, a
// query leaves float64 rows in device memory for SVNICP::add_cloud_device_target.  Same voxel order as the host map above.
class DeviceVoxelMap {
 public:
  DeviceVoxelMap(double voxel_size, double max_range, int max_points, int device = 0) {
    if (svnicp_map_create(device, voxel_size, max_range, max_points, 0, &m_) != 0) throw std::runtime_error(svnicp_map_last_error(nullptr));
  }
  ~DeviceVoxelMap() { svnicp_map_destroy(m_); }
  DeviceVoxelMap(const DeviceVoxelMap&) = delete;
  DeviceVoxelMap& operator=(const DeviceVoxelMap&) = delete;
  bool Empty() { return Size() == 0; }
  size_t Size() { int64_t n = 0; chk(svnicp_map_size(m_, &n)); return (size_t)n; }
  void AddPointCloud(const Cloud& cloud, const Pose3& pose) {
    chk(svnicp_map_add_cloud(m_, cloud.empty() ? nullptr : &cloud[0][0], (int64_t)cloud.size(), SVNICP_MEM_HOST, pose.R.data(), pose.t.data()));
  }
  void AddPointCloudDevice(const float* dev_xyz, int64_t n, const Pose3& pose) {   // float32 rows already in HBM (DevicePrep)
    chk(svnicp_map_add_cloud(m_, dev_xyz, n, SVNICP_MEM_DEVICE, pose.R.data(), pose.t.data()));
  }
  // -> number of points; the rows are at points_devptr()
  int64_t GetMap(const Pose3& pose, double max_range) { int64_t n = 0; chk(svnicp_map_query(m_, pose.t.data(), max_range, &n)); return n; }
  int64_t GetMap() { int64_t n = 0; chk(svnicp_map_query(m_, nullptr, -1.0, &n)); return n; }
  const double* points_devptr() { return static_cast<const double*>(svnicp_map_points_devptr(m_)); }
  std::vector<double> download() {
    int64_t n = 0;
    chk(svnicp_map_download(m_, nullptr, 0, &n));
    std::vector<double> o((size_t)3 * n);
    if (n) chk(svnicp_map_download(m_, o.data(), n, &n));
    return o;
  }

 private:
  void chk(int rc) { if (rc != 0) throw std::runtime_error(svnicp_map_last_error(m_)); }
  svnicp_map* m_ = nullptr;
};

// crop_pointcloud + the two uniform samplings on the device (svnicp_prep_*, csrc/scan_prep.hip): the raw scan is uploaded
// once; the cropped cloud, the map cloud (float32) and the source cloud (float64 rows) stay in HBM
class DevicePrep {
 public:
  explicit DevicePrep(int device = 0) { if (svnicp_prep_create(device, &p_) != 0) throw std::runtime_error(svnicp_prep_last_error(nullptr)); }
  ~DevicePrep() { svnicp_prep_destroy(p_); }
  DevicePrep(const DevicePrep&) = delete;
  DevicePrep& operator=(const DevicePrep&) = delete;
  void scan(const Cloud& points, double min_range, double max_range, double voxel_size, double* scan_max_range) {
    if (svnicp_prep_scan(p_, points.empty() ? nullptr : &points[0][0], (int64_t)points.size(), SVNICP_MEM_HOST, min_range, max_range, voxel_size,
                         scan_max_range, &n_cropped, &n_map, &n_source) != 0)
      throw std::runtime_error(svnicp_prep_last_error(p_));
  }
  const float* cropped() { return svnicp_prep_cropped_devptr(p_); }
  const float* map_cloud() { return svnicp_prep_map_cloud_devptr(p_); }
  const double* source() { return svnicp_prep_source_devptr(p_); }
  const float* source_f32() { return svnicp_prep_source_f32_devptr(p_); }
  std::vector<double> download_source() {   // test tap
    std::vector<float> f((size_t)3 * n_source);
    int64_t n = 0;
    if (n_source && svnicp_prep_download(p_, 2, f.data(), n_source, &n) != 0) throw std::runtime_error(svnicp_prep_last_error(p_));
    return std::vector<double>(f.begin(), f.end());
  }
  int64_t n_cropped = 0, n_map = 0, n_source = 0;

 private:
  svnicp_prep* p_ = nullptr;
};

// ------------------------------------------------------------------------------------------------ prediction
// OdometryPipeline::pose_prediction (:706-737): constant twist between the last two poses scaled by the time ratio;
// identity / last pose while fewer than two poses exist
inline Pose3 pose_prediction(const std::vector<Pose3>& poses, const std::vector<double>& times, double new_time) {
  if (poses.empty()) return Pose3{};
  if (poses.size() < 2) return poses.back();
  const Pose3& T0 = poses[poses.size() - 2];
  const Pose3& T1 = poses.back();
  const double dt = times.back() - times[times.size() - 2];
  const auto xi = se3_log(T0.inverse() * T1);
  const double ratio = (new_time - times.back()) / dt;
  double sx[6];
  for (int i = 0; i < 6; ++i) sx[i] = ratio * xi[i];
  return T1 * se3_exp(sx);
}

// ------------------------------------------------------------------------------------------------ the sequence
struct PipelineConfig {  // field names follow the node's parameters (OdometryPipeline.cpp parameter block; config/*.yaml)
  double min_range = 1.0, max_range = 100.0, voxel_size = 1.0;
  double map_voxel_size = 1.0, map_range = 100.0;
  int map_voxel_max_points = 20;
  int particle_count = 128;
  SteinICPParam solver;
  uint64_t seed = 0;
  int device = 0;
  bool gpu_map = false;  // keep the local map in HBM (DeviceVoxelMap): the target never crosses PCIe
  bool gpu_prep = false; // with gpu_map: crop and both uniform samplings on the device (DevicePrep): the raw scan is uploaded, no host pass over the points
};

struct ScanResult {
  double stamp = 0;
  Pose3 pose, initial_guess;
  bool aligned = false;
  int state = 0;
  std::array<double, 6> correction{}, variance{};
  std::vector<double> cov, particles, weights;
};

// what the pipeline handed to the solver for one scan (test tap: the oracle is run on exactly these)
struct Tap {
  std::vector<double> source, target, particles;  // [B][3], [M][3], [6][P]
  Pose3 initial_guess;
};

// particle prior bounds, OdometryPipeline.cpp:661-667
constexpr double kPriorUb[6] = {0.3, 0.2, 0.1, 0.004, 0.004, 0.012};

class RegistrationPipeline {
 public:
  explicit RegistrationPipeline(const PipelineConfig& cfg)
      : cfg_(cfg), map_(cfg.map_voxel_size, cfg.map_range, cfg.map_voxel_max_points), rng_(cfg.seed * 0x9e3779b97f4a7c15ull + 0x2545f4914f6cdd1dull) {
    if (cfg.gpu_map) dmap_ = std::make_unique<DeviceVoxelMap>(cfg.map_voxel_size, cfg.map_range, cfg.map_voxel_max_points, cfg.device);
    if (cfg.gpu_map && cfg.gpu_prep) dprep_ = std::make_unique<DevicePrep>(cfg.device);
  }

  // replaces the built-in uniform prior sampler (svnicp::initialize_particles, ICPUtils.cpp:45-58): fills [6][P] row-major
  void set_particle_source(std::function<void(int, double*)> f) { particle_source_ = std::move(f); }
  void set_tap(Tap* t) { tap_ = t; }
  const VoxelHashMap& map() const { return map_; }
  size_t map_voxels() { return dmap_ ? dmap_->Size() : map_.Size(); }
  size_t bytes_h2d() const { return bytes_h2d_; }   // cloud bytes sent to the GPU so far (source scans + map traffic)
  const std::vector<Pose3>& poses() const { return poses_; }

  // one pass of ICP_processing's loop body for one LiDAR frame (points: n x 3 float32, sensor frame)
  ScanResult process_scan(const Cloud& points, double stamp) {
    ScanResult res;
    res.stamp = stamp;
    Cloud cropped, to_map, source;
    if (dprep_) {
      dprep_->scan(points, cfg_.min_range, cfg_.max_range, cfg_.voxel_size, &scan_max_range_);         // :556-560 on the device
      bytes_h2d_ += points.size() * 12;
    } else {
      cropped = crop_pointcloud(points, cfg_.min_range, cfg_.max_range, &scan_max_range_);             // :556
      to_map = downsample_uniform(cropped, 0.5 * cfg_.voxel_size);                                     // :559
      source = downsample_uniform(to_map, 1.5 * cfg_.voxel_size);                                      // :560
    }
    const Pose3 guess = pose_prediction(poses_, times_, stamp);                                        // :563-564
    std::vector<double> init = sample_particles();                                                     // :573
    res.initial_guess = guess;
    if (dmap_ ? dmap_->Empty() : map_.Empty()) {                                                       // :585-593
      // the reference's downsample_uniform filters its input IN PLACE (:684-690): at :585 *cropped_cloud already holds the
      // 0.5-voxel sampling, and that is what seeds the map
      if (dprep_) dmap_->AddPointCloudDevice(dprep_->map_cloud(), dprep_->n_map, guess);
      else if (dmap_) { dmap_->AddPointCloud(to_map, guess); bytes_h2d_ += to_map.size() * 12; }
      else map_.AddPointCloud(to_map, guess);
      poses_.push_back(guess); times_.push_back(stamp);
      res.pose = guess;
      return res;
    }
    std::vector<double> src64;
    if (!dprep_) src64 = widen(source);                                                                // ICPUtils.cpp:27-43
    if (!solver_) solver_ = std::make_unique<SVNICP>(cfg_.solver, init, ParticleWeightOpt{}, cfg_.device);
    std::vector<double> tgt64;
    if (dmap_) {
      int64_t M = dmap_->GetMap(guess, scan_max_range_ + 10.0);                                        // :577-578
      if (M == 0) M = dmap_->GetMap();                                                                 // :579-581
      if (dprep_) {
        solver_->add_cloud_device(dprep_->source(), dprep_->n_source, dmap_->points_devptr(), M, init.data(), cfg_.particle_count);
        if (tap_) src64 = dprep_->download_source();
      } else {
        solver_->add_cloud_device_target(src64.data(), (int64_t)source.size(), dmap_->points_devptr(), M, init.data(), cfg_.particle_count);
        bytes_h2d_ += src64.size() * 8;
      }
      if (tap_) tgt64 = dmap_->download();
    } else {
      Cloud target = map_.GetMap(guess, scan_max_range_ + 10.0);                                       // :577-578
      if (target.empty()) target = map_.GetMap();                                                      // :579-581
      tgt64 = widen(target);
      solver_->add_cloud(src64.data(), (int64_t)source.size(), tgt64.data(), (int64_t)target.size(), init.data(), cfg_.particle_count);  // :582
      bytes_h2d_ += (src64.size() + tgt64.size()) * 8;
    }
    solver_->set_initial_mean(guess.R.data(), guess.t.data());                                         // :598
    if (tap_) { tap_->source = src64; tap_->target = tgt64; tap_->particles = init; tap_->initial_guess = guess; }
    res.state = (int)solver_->stein_align();                                                           // :599
    if (res.state != ALIGN_SUCCESS) { res.pose = guess; return res; }                                  // :599-601
    res.aligned = true;
    res.correction = solver_->get_transformation();                                                    // :602
    res.variance = solver_->get_distribution();
    res.cov = solver_->get_cov_matrix();
    res.particles = solver_->get_particles();
    res.weights = solver_->get_particle_weight();
    res.pose = guess * correction_to_pose(res.correction);                                             // updater_, :37-46
    // … and at :630 *voxelized_cloud_toMap holds the 1.5-voxel sampling (the second in-place filter, :560): the map is updated
    // with the same points the solver registered
    if (dprep_) dmap_->AddPointCloudDevice(dprep_->source_f32(), dprep_->n_source, res.pose);          // :630
    else if (dmap_) { dmap_->AddPointCloud(source, res.pose); bytes_h2d_ += source.size() * 12; }
    else map_.AddPointCloud(source, res.pose);
    poses_.push_back(res.pose); times_.push_back(stamp);
    return res;
  }

 private:
  static std::vector<double> widen(const Cloud& c) {
    std::vector<double> o(3 * c.size());
    for (size_t i = 0; i < c.size(); ++i)
      for (int d = 0; d < 3; ++d) o[3 * i + d] = (double)c[i][d];
    return o;
  }
  std::vector<double> sample_particles() {
    const int P = cfg_.particle_count;
    std::vector<double> init((size_t)6 * P, 0.0);
    if (particle_source_) { particle_source_(P, init.data()); return init; }
    if (P == 1) return init;  // ICPUtils.cpp:49-50
    for (int d = 0; d < 6; ++d)
      for (int p = 0; p < P; ++p) init[(size_t)d * P + p] = (2.0 * uniform01() - 1.0) * kPriorUb[d];
    return init;
  }
  double uniform01() {  // splitmix64 -> 53-bit uniform
    uint64_t z = (rng_ += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
  }

  PipelineConfig cfg_;
  VoxelHashMap map_;
  std::vector<Pose3> poses_;
  std::vector<double> times_;
  double scan_max_range_ = 0.0;
  uint64_t rng_;
  std::function<void(int, double*)> particle_source_;
  Tap* tap_ = nullptr;
  std::unique_ptr<SVNICP> solver_;
  std::unique_ptr<DeviceVoxelMap> dmap_;
  std::unique_ptr<DevicePrep> dprep_;
  size_t bytes_h2d_ = 0;
};

}  // namespace svnicp
