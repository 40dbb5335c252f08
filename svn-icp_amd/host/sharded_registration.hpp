// sharded_registration.hpp — C++ multi-GPU host of the Stein-ICP registration: one process per GPU, RCCL over xGMI for the
// one exchange step per iteration.  Header-only, on the C ABI (svnicp_hip.h) + <rccl/rccl.h>; the C++ sibling of
// svn-icp_amd/sharded.py (same two splits, same call sequence) for hosts that stay C++ like the reference's
// (OdometryPipeline.cpp:573-607 is the caller this replaces the solver calls of).  New functionality: the reference is one
// process on one GPU.
//
//   split Rows (default)   rank r is handed source rows [r·ceil(B/W), …): stage A, the candidate table and the per-iteration
//                          search + accumulation cover those rows for ALL particles; per iteration ONE ncclAllGather of
//                          W × P × 22 doubles (the ranks' partial sums) straight into the library's record array, which
//                          svnicp_iter_update adds in rank order on every rank — bit-identical replicas.
//   split Particles        rank r owns particles [r·P/W, (r+1)·P/W); clouds replicated; stage A sharded by rows with one
//                          ncclAllGather of the int32 candidate rows, then per iteration one ncclAllGather of 176 B per
//                          particle.  Needs P and B to be multiples of W (the Python driver pads ragged shards).
// Kernels and collectives share ONE queue: the library runs on the stream handed to svnicp_set_stream and the collectives
// are enqueued on the same stream, so nothing synchronises with the host inside an iteration; the early-stop flag is
// polled every `stop_poll` iterations (a 8-byte copy + stream sync), as in sharded.py.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "svnicp_hip_shim.hpp"

namespace svnicp {

enum class Split { Rows, Particles };

inline void shard_range(int64_t n, int world, int rank, int64_t* lo, int64_t* hi) {
  const int64_t per = (n + world - 1) / world;
  *lo = std::min<int64_t>(n, (int64_t)rank * per);
  *hi = std::min<int64_t>(n, *lo + per);
}

// SOLVER = svnicp::SVNICP or svnicp::SVGDICP (svnicp_hip_shim.hpp)
template <class SOLVER>
class Sharded {
 public:
  // comm: a communicator of `world` ranks, this process being `rank` (world == 1: comm may be null); stream: the HIP stream
  // the library and the collectives run on
  Sharded(const SteinICPParam& prm, const std::vector<double>& init_pose, int device, ncclComm_t comm, int rank, int world,
          hipStream_t stream, Split split = Split::Rows)
      : prm_(prm), solver_(make(prm, init_pose, device)), comm_(comm), rank_(rank), world_(world), stream_(stream), split_(split),
        P_((int)(init_pose.size() / 6)) {
    chk(svnicp_set_stream(solver_.handle(), (void*)stream_), "svnicp_set_stream");
  }

  // add_cloud(source[B,3], target[M,3], init_pose[6,P,1]) — SVGDICP.cpp:46-62; host buffers.  With rows sharded only this
  // rank's slice of the source is uploaded.
  void add_cloud(const double* source_xyz, int64_t B, const double* target_xyz, int64_t M, const double* init_pose6xP, int P) {
    B_total_ = B; P_ = P;
    int64_t lo = 0, hi = B;
    if (split_ == Split::Rows && world_ > 1) {
      if (B < world_) throw std::runtime_error("svnicp::Sharded: fewer source points than ranks");
      shard_range(B, world_, rank_, &lo, &hi);
    }
    B_ = hi - lo;
    solver_.add_cloud(source_xyz + 3 * lo, B_, target_xyz, M, init_pose6xP, P);
  }
  void set_initial_mean(const double R_rowmajor[9], const double t[3]) { solver_.set_initial_mean(R_rowmajor, t); }

  SteinICPState stein_align() {   // SVNICP.cpp:41-114 / SVGDICP.cpp:66-140 through the split-phase entry points
    svnicp_ctx* h = solver_.handle();
    if (std::is_same<SOLVER, SVGDICP>::value && prm_.optimizer != "Adam" && prm_.optimizer != "RMSprop" && prm_.optimizer != "SGD" &&
        prm_.optimizer != "Adagrad")
      return NO_OPTIMIZER;   // set_optimizer() found none: stein_align returns at once (SVGDICP.cpp:73-75)
    const bool rows = split_ == Split::Rows;
    int64_t p_lo = 0, p_hi = P_;
    if (!rows && world_ > 1) {
      if (P_ % world_ || B_ % world_) throw std::runtime_error("svnicp::Sharded: the particle split needs P and B to be multiples of the world size");
      shard_range(P_, world_, rank_, &p_lo, &p_hi);
    }
    chk(svnicp_set_shard(h, (int)p_lo, (int)p_hi), "svnicp_set_shard");
    chk(svnicp_set_row_shard(h, rows ? rank_ : 0, rows ? world_ : 1, rows ? B_total_ : B_), "svnicp_set_row_shard");
    chk(svnicp_align_begin(h), "svnicp_align_begin");
    if (rows || world_ == 1) {
      chk(svnicp_stage_candidates(h, 0, B_), "svnicp_stage_candidates");
    } else {   // rows replicated: every rank searches B / W rows, the int32 result rows are gathered once
      int64_t b_lo, b_hi;
      shard_range(B_, world_, rank_, &b_lo, &b_hi);
      chk(svnicp_stage_candidates(h, b_lo, b_hi), "svnicp_stage_candidates");
      int32_t* cand = static_cast<int32_t*>(svnicp_candidates_devptr(h));
      const size_t cnt = (size_t)(b_hi - b_lo) * prm_.KNN_count;
      nccl(ncclAllGather(cand + (size_t)b_lo * prm_.KNN_count, cand, cnt, ncclInt32, comm_, stream_), "ncclAllGather(candidates)");
    }
    chk(svnicp_build_candidate_table(h), "svnicp_build_candidate_table");
    for (int it = 0; it < prm_.iterations; ++it) {
      chk(svnicp_iter_accumulate(h, it), "svnicp_iter_accumulate");
      if (world_ > 1) {
        if (rows) {   // slot `rank` of [W][P][22] is this rank's partial record: in-place all-gather
          double* rec = static_cast<double*>(svnicp_rank_sums_devptr(h));
          const size_t cnt = (size_t)P_ * SVNICP_NSUMS;
          nccl(ncclAllGather(rec + (size_t)rank_ * cnt, rec, cnt, ncclDouble, comm_, stream_), "ncclAllGather(partial sums)");
        } else {      // rows [p_lo, p_hi) of [P][22] are this rank's particles' complete records
          double* rec = static_cast<double*>(svnicp_sums_devptr(h));
          const size_t cnt = (size_t)(p_hi - p_lo) * SVNICP_NSUMS;
          nccl(ncclAllGather(rec + (size_t)p_lo * SVNICP_NSUMS, rec, cnt, ncclDouble, comm_, stream_), "ncclAllGather(sums)");
        }
      }
      chk(svnicp_iter_update(h, it), "svnicp_iter_update");
      // the stop flag lives on the device and later launches return at once when it is set: the host only looks every few
      // iterations; every rank sees the same value because the update ran on identical inputs
      if (prm_.check_early_stop && it % stop_poll == stop_poll - 1) {
        const int st = svnicp_stopped(h);
        if (st < 0) fail("svnicp_stopped");
        if (st > 0) break;
      }
    }
    chk(svnicp_finish(h), "svnicp_finish");
    chk(svnicp_synchronize(h), "svnicp_synchronize");
    return ALIGN_SUCCESS;
  }

  SOLVER& solver() { return solver_; }   // getters of the reference interface: get_transformation(), get_cov_matrix(), …
  int64_t local_rows() const { return B_; }
  int stop_poll = 4;

 private:
  static SOLVER make(const SteinICPParam& prm, const std::vector<double>& init, int device);
  void chk(int rc, const char* what) { if (rc != 0) fail(what); }
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string(what) + ": " + svnicp_last_error(solver_.handle())); }
  void nccl(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }

  SteinICPParam prm_;
  SOLVER solver_;
  ncclComm_t comm_;
  int rank_, world_;
  hipStream_t stream_;
  Split split_;
  int P_;
  int64_t B_ = 0, B_total_ = 0;
};

template <> inline SVNICP Sharded<SVNICP>::make(const SteinICPParam& prm, const std::vector<double>& init, int device) {
  return SVNICP(prm, init, ParticleWeightOpt{}, device);
}
template <> inline SVGDICP Sharded<SVGDICP>::make(const SteinICPParam& prm, const std::vector<double>& init, int device) {
  return SVGDICP(prm, init, device);
}

}  // namespace svnicp
