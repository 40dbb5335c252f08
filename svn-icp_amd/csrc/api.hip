// api.hip — the C ABI of libsvnicp_hip.so (include/svnicp_hip.h): context, device buffers,
// launch sequencing.  Host-side mirror of the reference's solver object state
// (include/core/SVGDICP.h:170-210): clouds, R0/t0, particles R_/t_, pose_particles_, history.
// No CPU fallback: every compute entry point needs a gfx950 device and fails loudly without one.
#include "../../include/svnicp_hip.h"

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"

using namespace svnicp;

namespace {

thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  hipError_t ensure(size_t n) {
    if (n <= cap && p) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (n == 0) n = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct svnicp_ctx {
  svnicp_params prm{};
  int device = 0;
  int num_cus = 256;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // second queue: the pair statistics of the Stein step (they need the poses only) run here, beside the stage-B kernels of
  // the same iteration; forked from and joined into `stream` with events, so the caller still sees one ordered queue
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_init = nullptr;
  // blocking svnicp_align with early stop: the stop flag follows every few iterations into pinned memory, so that the host
  // stops enqueuing soon after the device has stopped (three slots, the host runs two chunks ahead)
  hipEvent_t ev_chunk[3] = {nullptr, nullptr, nullptr};
  int* h_flags = nullptr;   // pinned [3]
  bool init_in_flight = false;
  bool median_pending = false;
  // pinned host staging: the initial particles go up and the result block (mean, variance, covariance, weights) comes down
  // without a stream synchronisation of their own
  double* h_init = nullptr; size_t h_init_cap = 0;
  double* h_stats = nullptr; size_t h_stats_cap = 0;
  bool host_stats_valid = false;   // h_stats holds the last registration's results (after the stream has been synchronised)
  int single_done_it = -1;   // iteration whose Stein step the accumulate kernel's last workgroup has already enqueued (P = 1)
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  std::string err;

  int64_t B = 0, M = 0, Mp = 0;
  int K = 0, S = 0, P = 0;
  int p_lo = 0, p_hi = 0;
  bool src_set = false, tgt_set = false;
  bool clouds_set = false, particles_set = false, particles_dirty = false, shard_set = false;
  bool began = false, have_candidates = false, have_result = false;
  Pose0 pose0{};
  AccumPlan plan{};
  int plan_P = -1; int64_t plan_B = -1; int plan_K = -1;

  DevBuf<double> eul, opt, uctl;
  DevBuf<double> src, tgt, tx, ty, tz, pool_d, cand_d2, table, init_pose, R, t, Rtot, pose_out, sums, partial, work,
      stats, trH, trb, trN, trphi, trh;
  DevBuf<int32_t> pool_i, cand_idx, trcorr, torig, pool2, fail_list;
  DevBuf<float> txf, tyf, tzf, cmaxb;
  DevBuf<float4> tablef, tablea;
  DevBuf<unsigned int> small_bar;   // persistent small-registration kernel: arrivals, generation, error word
  bool small_launched = false;      // the last svnicp_align_async ran the persistent kernel (its error word is checked at the next synchronisation)
  bool defer_fin = false;           // svnicp_align's own loop: iteration i's early-stop decision is taken by iteration i + 1's search kernel
  DevBuf<uint8_t> kbest;
  DevBuf<int32_t> kidx;
  // source-row sharding (svnicp_set_row_shard): this context holds rows of a larger scan; its per-iteration sums are one of
  // row_world partial records that the host all-gathers into rank_sums [row_world][P][22]
  int row_rank = 0, row_world = 1;
  int64_t B_total = 0;
  DevBuf<double> rank_sums;
  DevBuf<double> sl_d, fail_tau, anchor, qrec;
  DevBuf<int32_t> sl_i;
  int sliced_max = 0;  // set at align_begin (kFallbackSlicedMax or SVNICP_FALLBACK_SLICED_MAX)
  DevBuf<int> ambig;
  int accum_mode = 3;  // 0 f64 baseline, 1 f32 VALU search (fused), 3 bf16 matrix-pipe search + accumulate kernels
  DevBuf<unsigned long long> emax;
  DevBuf<int> fail_count;
  // correspondence = full reuses fail_list / fail_count for every per-particle search: stage A's own are kept here
  DevBuf<int> stage_fail_count;
  DevBuf<int32_t> stage_fail_list;
  // stage A variant: 0 = streaming only (knn_topk), 1 = seeded f32 scan (knn_scan), 2 = pruned tiles (knn_tiles)
  int knn_variant = 0;
  int target_layout = -1;  // what the SoA currently holds: 0 hashed order, 1 Morton tiles, -1 nothing
  DevBuf<unsigned int> keys_a, keys_b;
  DevBuf<int32_t> vals_a, order_t, qorder, stat_n;
  DevBuf<unsigned long long> bbox;
  DevBuf<float> tile_box;
  DevBuf<unsigned char> sort_tmp;
  size_t sort_tmp_bytes = 0;
  bool use_scan = false;   // stage A through knn_scan.hip (f32 pre-filter) with knn_topk.hip as fallback
  int64_t scan_Ms = 0; int scan_rank = 0, scan_S2 = 0;
  DevBuf<float> history;
  DevBuf<int> ctl;
  int hist_I = 0, hist_P = 0;
  Tuning tune{};
  DevBuf<double> full_q, full_d2;      // correspondence = full: one particle's transformed source, its nearest distances
  DevBuf<int32_t> arena, chunk_tab;    // stage A (Morton tiles): overflow chunks of the survivor pools and their table
  int arena_cap = 0;
  DevBuf<int32_t> full_idx;            // … and the nearest target of every (particle of the shard, source point): [P][B]
  unsigned long long* dbg_phase = nullptr;   // debug option: per-phase wave cycles of k_knn_tiles (per context, per device)
  unsigned long long* dbg_upd = nullptr;     // debug option: phase cycles of k_particle_update
  bool finish_seen = true;   // the stop flag of the last registration has been folded into finish_iter
  int finish_iter = 0;   // finish_iter_: constructor value, changed only by an SVGD-mode early stop (SVGDICP.cpp:42,128)
  double gpu_ms[3] = {0, 0, 0};
  bool timing_valid = false;
  // optional per-kernel-class timing (svnicp_set_profile): event pairs around every launch
  bool profile = false;
  unsigned profile_mask = 0;            // classes that are bracketed (bit = class index)
  int pcur = -1;                        // class of the open bracket, -1 = none
  std::vector<hipEvent_t> pev;          // pairs: [2*i] start, [2*i+1] stop
  std::vector<int> pcls;                // kernel class of pair i
  size_t pused = 0;
};

// Tuning::fused_update_max_p default 128: measured crossover (C3: equal, P=256: 4.5x); above it the Stein step runs as
// workgroup-parallel kernels
constexpr int kFallbackGrid = 256;  // workgroups of the streaming kernel when it only redoes failed queries
constexpr int kFallbackSlicedMax = 512;  // up to this many failed queries are redone by target slices (all CUs per query)
constexpr int kFallbackQW = 2;      // … two queries per wave, so a few hundred failures still run in parallel

enum { KC_KNN = 0, KC_TABLE = 1, KC_SEARCH = 2, KC_ACCUM = 3, KC_REDUCE = 4, KC_UPDATE = 5, KC_COUNT = SVNICP_KERNEL_CLASSES };

static hipError_t prof_begin(svnicp_ctx* c, int cls) {
  c->pcur = -1;
  if (!c->profile || !((c->profile_mask >> cls) & 1u)) return hipSuccess;
  c->pcur = cls;
  if (c->pused * 2 + 2 > c->pev.size()) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t e;
      hipError_t r = hipEventCreate(&e);
      if (r != hipSuccess) return r;
      c->pev.push_back(e);
    }
    c->pcls.push_back(cls);
  }
  c->pcls[c->pused] = cls;
  return hipEventRecord(c->pev[2 * c->pused], c->stream);
}
static hipError_t prof_end(svnicp_ctx* c) {
  if (!c->profile || c->pcur < 0) return hipSuccess;
  c->pcur = -1;
  hipError_t r = hipEventRecord(c->pev[2 * c->pused + 1], c->stream);
  c->pused += 1;
  return r;
}

#define CTX_CHECK(ctx)                         \
  do {                                         \
    if (!(ctx)) return SVNICP_ERR_INVALID;     \
  } while (0)

static int fail(svnicp_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}
#define HIPCHK(c, expr)                                                                          \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return fail((c), _e == hipErrorOutOfMemory ? SVNICP_ERR_NOMEM : SVNICP_ERR_HIP,           \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                            \
  } while (0)

static int bind(svnicp_ctx* c) {
  HIPCHK(c, hipSetDevice(c->device));
  return 0;
}

extern "C" {

int svnicp_abi_version(void) { return SVNICP_ABI_VERSION; }

const char* svnicp_last_error(const svnicp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int svnicp_create(const svnicp_params* params, int device, const double* init_pose6xP, int P, svnicp_ctx** out) {
  if (!params || !out) return fail(nullptr, SVNICP_ERR_INVALID, "svnicp_create: null argument");
  if (params->struct_size != (int32_t)sizeof(svnicp_params))
    return fail(nullptr, SVNICP_ERR_INVALID, "svnicp_create: svnicp_params.struct_size mismatch");
  if (params->iterations < 0 || params->knn_count < 1)
    return fail(nullptr, SVNICP_ERR_INVALID, "svnicp_create: iterations >= 0 and knn_count >= 1 required");
  if (params->mode != SVNICP_MODE_SVN && params->mode != SVNICP_MODE_SVGD)
    return fail(nullptr, SVNICP_ERR_INVALID, "svnicp_create: unknown mode");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, SVNICP_ERR_NO_DEVICE, "svnicp_create: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(nullptr, SVNICP_ERR_INVALID, "svnicp_create: bad device ordinal");
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess)
    return fail(nullptr, SVNICP_ERR_HIP, "svnicp_create: hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, SVNICP_ERR_NO_DEVICE,
                std::string("svnicp_create: device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
  svnicp_ctx* c = new svnicp_ctx();
  c->prm = *params;
  c->device = device;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->K = params->knn_count;
  c->finish_iter = params->iterations;
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(c->pose0.R0, I3, sizeof I3);  // SVGDICP.cpp:38-39
  c->pose0.t0[0] = c->pose0.t0[1] = c->pose0.t0[2] = 0.0;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return fail(nullptr, SVNICP_ERR_HIP, "svnicp_create: hipStreamCreate failed");
  }
  c->stream = c->own_stream;
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_init, hipEventDisableTiming) != hipSuccess) {
    svnicp_destroy(c);
    return fail(nullptr, SVNICP_ERR_HIP, "svnicp_create: hipStreamCreate / hipEventCreate failed");
  }
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete c; return fail(nullptr, SVNICP_ERR_HIP, "hipEventCreate failed"); }
  if (c->ctl.ensure(4) != hipSuccess) { delete c; return fail(nullptr, SVNICP_ERR_NOMEM, "hipMalloc failed"); }
  *out = c;
  if (const char* e = getenv("SVNICP_OPTIONS")) {   // read ONCE, at creation: "name=value;name=value" for profiling scripts
    std::string all(e);
    size_t pos = 0;
    while (pos < all.size()) {
      const size_t end = all.find(';', pos) == std::string::npos ? all.size() : all.find(';', pos);
      const std::string kv = all.substr(pos, end - pos);
      const size_t eq = kv.find('=');
      if (eq != std::string::npos && svnicp_set_option(c, kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str()) != 0) {
        g_create_error = c->err; svnicp_destroy(c); *out = nullptr; return SVNICP_ERR_INVALID;
      }
      pos = end + 1;
    }
  }
  if (init_pose6xP) {
    int rc = svnicp_set_particles(c, init_pose6xP, P);
    if (rc != 0) { g_create_error = c->err; svnicp_destroy(c); *out = nullptr; return rc; }
  }
  return SVNICP_OK;
}

void svnicp_destroy(svnicp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->h_init) (void)hipHostFree(c->h_init);
  if (c->h_stats) (void)hipHostFree(c->h_stats);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_init) (void)hipEventDestroy(c->ev_init);
  for (auto& e : c->ev_chunk) if (e) (void)hipEventDestroy(e);
  if (c->h_flags) (void)hipHostFree(c->h_flags);
  DevBuf<double>* dbl[] = {&c->src, &c->tgt, &c->tx, &c->ty, &c->tz, &c->pool_d, &c->cand_d2, &c->table,
                           &c->init_pose, &c->R, &c->t, &c->Rtot, &c->pose_out, &c->sums, &c->partial, &c->work,
                           &c->stats, &c->trH, &c->trb, &c->trN, &c->trphi, &c->trh};
  for (auto* b : dbl) b->release();
  c->eul.release(); c->opt.release(); c->uctl.release(); c->rank_sums.release(); c->stage_fail_count.release(); c->stage_fail_list.release(); c->small_bar.release();
  c->keys_a.release(); c->keys_b.release(); c->vals_a.release(); c->order_t.release(); c->qorder.release(); c->stat_n.release(); c->bbox.release(); c->tile_box.release(); c->sort_tmp.release();
  c->full_q.release(); c->full_d2.release(); c->full_idx.release(); c->arena.release(); c->chunk_tab.release();
  c->pool_i.release(); c->torig.release(); c->pool2.release(); c->fail_list.release(); c->txf.release(); c->tyf.release(); c->tzf.release(); c->cmaxb.release(); c->tablef.release(); c->tablea.release(); c->kbest.release(); c->kidx.release(); c->sl_d.release(); c->sl_i.release(); c->fail_tau.release(); c->qrec.release(); c->anchor.release(); c->ambig.release(); c->emax.release(); c->fail_count.release(); c->cand_idx.release(); c->trcorr.release(); c->history.release(); c->ctl.release();
  if (c->dbg_phase) (void)hipFree(c->dbg_phase);
  if (c->dbg_upd) (void)hipFree(c->dbg_upd);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->pev) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int svnicp_set_stream(svnicp_ctx* c, void* hip_stream) {
  CTX_CHECK(c);
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, hipStreamSynchronize(c->stream));  // nothing of ours may still be in flight on the old stream
  c->stream = hip_stream == SVNICP_OWN_STREAM ? c->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
  return SVNICP_OK;
}

static int check_small_kernel(svnicp_ctx* c);
int svnicp_synchronize(svnicp_ctx* c) {
  CTX_CHECK(c);
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (const int rc = check_small_kernel(c)) return rc;
  if (c->have_result && c->h_stats) c->host_stats_valid = true;   // svnicp_finish's copy of the result block has landed
  return SVNICP_OK;
}

int svnicp_set_source(svnicp_ctx* c, const double* src, int64_t B, int mem_kind) {
  CTX_CHECK(c);
  if (!src || B < 1) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_source: need B >= 1");
  if (B > 0x7fffffffLL) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_source: cloud too large");
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, c->src.ensure((size_t)B * 3));
  const hipMemcpyKind kind = mem_kind == SVNICP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIPCHK(c, hipMemcpyAsync(c->src.p, src, (size_t)B * 24, kind, c->stream));
  if (mem_kind != SVNICP_MEM_DEVICE) HIPCHK(c, hipStreamSynchronize(c->stream));  // caller may reuse its host buffer
  c->B = B;
  c->src_set = true;
  c->clouds_set = c->src_set && c->tgt_set;
  c->have_candidates = false;
  return SVNICP_OK;
}

int svnicp_set_target(svnicp_ctx* c, const double* tgt, int64_t M, int mem_kind) {
  CTX_CHECK(c);
  if (!tgt || M < 1) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_target: need M >= 1");
  if (M > 0x7fffffffLL) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_target: cloud too large");
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, c->tgt.ensure((size_t)M * 3));
  const hipMemcpyKind kind = mem_kind == SVNICP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIPCHK(c, hipMemcpyAsync(c->tgt.p, tgt, (size_t)M * 24, kind, c->stream));
  if (mem_kind != SVNICP_MEM_DEVICE) HIPCHK(c, hipStreamSynchronize(c->stream));
  c->M = M; c->Mp = knn_padded_targets(M);
  c->target_layout = -1;  // the SoA copies are (re)built in svnicp_align_begin, once K is final
  c->tgt_set = true;
  c->clouds_set = c->src_set && c->tgt_set;
  c->have_candidates = false;
  return SVNICP_OK;
}

int svnicp_set_clouds(svnicp_ctx* c, const double* src, int64_t B, const double* tgt, int64_t M, int mem_kind) {
  CTX_CHECK(c);
  if (!src || !tgt || B < 1 || M < 1) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_clouds: need B >= 1, M >= 1");
  if (M > 0x7fffffffLL || B > 0x7fffffffLL) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_clouds: cloud too large");
  int rc = svnicp_set_source(c, src, B, mem_kind);
  if (rc) return rc;
  return svnicp_set_target(c, tgt, M, mem_kind);
}

int svnicp_set_particles(svnicp_ctx* c, const double* init, int P) {
  CTX_CHECK(c);
  if (!init || P < 1) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_particles: need P >= 1");
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, c->init_pose.ensure((size_t)P * 6));
  HIPCHK(c, c->R.ensure((size_t)P * 9));
  HIPCHK(c, c->t.ensure((size_t)P * 3));
  HIPCHK(c, c->Rtot.ensure((size_t)P * 12));
  HIPCHK(c, c->pose_out.ensure((size_t)P * 6));
  HIPCHK(c, c->sums.ensure((size_t)P * kNSums));
  HIPCHK(c, c->stats.ensure((size_t)48 + P));
  HIPCHK(c, c->work.ensure(update_workspace_doubles(P)));
  HIPCHK(c, c->eul.ensure((size_t)P * 6));
  HIPCHK(c, c->opt.ensure((size_t)P * 18));
  HIPCHK(c, c->uctl.ensure(update_uctl_doubles(P)));
  // through pinned staging: the caller's buffer is free when this returns and the stream is not synchronised (a second
  // call before the first copy has run would overwrite the staging area: wait for the stream only then)
  if ((size_t)P * 6 > c->h_init_cap || (size_t)P + 48 > c->h_stats_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h_init) (void)hipHostFree(c->h_init);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    c->h_init = c->h_stats = nullptr; c->h_init_cap = c->h_stats_cap = 0;
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_init), (size_t)P * 48, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_stats), ((size_t)P + 48) * 8, hipHostMallocDefault));
    c->h_init_cap = (size_t)P * 6; c->h_stats_cap = (size_t)P + 48;
  } else if (c->init_in_flight) {
    HIPCHK(c, hipEventSynchronize(c->ev_init));
  }
  std::memcpy(c->h_init, init, (size_t)P * 48);
  HIPCHK(c, hipMemcpyAsync(c->init_pose.p, c->h_init, (size_t)P * 48, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_init, c->stream));
  c->init_in_flight = true;
  c->host_stats_valid = false;
  const bool first = !c->particles_set || P != c->P;
  if (first && c->row_world > 1) { c->row_world = 1; c->row_rank = 0; c->B_total = 0; }   // the record array is sized by P: set the row shard again
  c->P = P;
  if (!c->shard_set || first) { c->p_lo = 0; c->p_hi = P; c->shard_set = false; }
  // ctor semantics: pose_particles_ is formed from the initial pose (SVNICP.cpp:36-37, SVGDICP.cpp:33-35);
  // add_cloud semantics: R_, t_ are reset, pose_particles_ is left alone (SVGDICP.cpp:46-62)
  HIPCHK(c, launch_init_particles(c->init_pose.p, P, c->pose0, c->prm.mode, c->R.p, c->t.p, c->Rtot.p, c->pose_out.p,
                                  (first || c->prm.mode == SVNICP_MODE_SVN) ? 1 : 0, c->eul.p, c->stream));
  c->particles_set = true;
  c->particles_dirty = true;
  return SVNICP_OK;
}

int svnicp_set_initial_mean(svnicp_ctx* c, const double R0[9], const double t0[3]) {
  CTX_CHECK(c);
  if (!R0 || !t0) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_initial_mean: null argument");
  std::memcpy(c->pose0.R0, R0, 9 * sizeof(double));
  std::memcpy(c->pose0.t0, t0, 3 * sizeof(double));
  c->have_candidates = false;
  return SVNICP_OK;
}

int svnicp_set_k(svnicp_ctx* c, int k) {
  CTX_CHECK(c);
  if (k < 1) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_k: k >= 1 required");
  c->K = k;
  c->have_candidates = false;
  return SVNICP_OK;
}

int svnicp_set_max_dist(svnicp_ctx* c, double md) {
  CTX_CHECK(c);
  c->prm.max_dist = md;
  return SVNICP_OK;
}

int svnicp_set_option(svnicp_ctx* c, const char* name, const char* value) {
  CTX_CHECK(c);
  if (!name || !value) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_option: null argument");
  const std::string k(name), v(value);
  Tuning& t = c->tune;
  auto num = [&](int lo, int hi, int* out) { char* end = nullptr; const long x = strtol(v.c_str(), &end, 10);
                                             if (end == v.c_str() || *end || x < lo || x > hi) return false; *out = (int)x; return true; };
  bool ok = true;
  if (k == "knn") { if (v == "auto") t.knn = -1; else if (v == "v1") t.knn = 0; else if (v == "v2") t.knn = 1; else if (v == "brute") t.knn = 2; else if (v == "tiles") t.knn = 3; else ok = false; }
  else if (k == "fallback_sliced_max") ok = num(-1, 1 << 20, &t.fallback_sliced_max);
  else if (k == "accum") { if (v == "f64") t.accum = 0; else if (v == "valu") t.accum = 1; else if (v == "split") t.accum = 3; else ok = false; }
  else if (k == "update") { if (v == "auto") t.update_fused = 0; else if (v == "fused") t.update_fused = 1; else ok = false; }
  else if (k == "fused_update_max_p") ok = num(1, 700, &t.fused_update_max_p);   // P = 1 has no pair statistics; the fused kernel's LDS ends near P = 800
  else if (k == "wgpcu") { int x = 0, y = 0; ok = sscanf(v.c_str(), "%d,%d", &x, &y) == 2 && x >= 0 && x <= 64 && y >= 0 && y <= 16; if (ok) { t.wgpcu_search = x; t.wgpcu_accum = y; } }
  else if (k == "tp") ok = num(0, 1 << 16, &t.tp);
  else if (k == "debug") ok = num(0, 1, &t.debug);
  else if (k == "scan_split") ok = num(0, 16, &t.scan_split);
  else if (k == "group_stride") ok = num(0, 1 << 30, &t.group_stride);
  else if (k == "accum_min_steps") ok = num(0, 1 << 20, &t.accum_min_steps);
  else if (k == "brute_qb") { ok = num(0, 6, &t.brute_qb); }   // queries per workgroup of k_knn_brute, 0 = automatic
  else if (k == "single") { if (v == "fused") t.single_fused = 1; else if (v == "split") t.single_fused = 0; else ok = false; }
  else if (k == "chain") { if (v == "auto") { t.small_chain = 1; t.persistent = 0; } else if (v == "persistent") { t.small_chain = 1; t.persistent = 1; }
                           else if (v == "general") { t.small_chain = 0; t.persistent = 0; } else ok = false; }
  else if (k == "median") { if (v == "auto") t.median_inline = -1; else if (v == "stream") t.median_inline = 0; else if (v == "inline") t.median_inline = 1; else ok = false; }
  else if (k == "correspondence") { if (v == "fast") t.full_corr = 0; else if (v == "full") t.full_corr = 1; else ok = false; }
  else return fail(c, SVNICP_ERR_INVALID, "svnicp_set_option: unknown option '" + k + "'");
  if (!ok) return fail(c, SVNICP_ERR_INVALID, "svnicp_set_option: bad value '" + v + "' for option '" + k + "'");
  c->have_candidates = false;
  c->target_layout = -1;
  return SVNICP_OK;
}

int svnicp_set_shard(svnicp_ctx* c, int p_lo, int p_hi) {
  CTX_CHECK(c);
  if (!c->particles_set || p_lo < 0 || p_hi > c->P || p_lo > p_hi)
    return fail(c, SVNICP_ERR_INVALID, "svnicp_set_shard: need 0 <= p_lo <= p_hi <= P after svnicp_set_particles");
  c->p_lo = p_lo; c->p_hi = p_hi; c->shard_set = true;
  return SVNICP_OK;
}

int svnicp_set_row_shard(svnicp_ctx* c, int row_rank, int row_world, int64_t total_source_points) {
  CTX_CHECK(c);
  if (!c->particles_set || row_world < 1 || row_rank < 0 || row_rank >= row_world || (row_world > 1 && total_source_points < 1))
    return fail(c, SVNICP_ERR_INVALID, "svnicp_set_row_shard: need 0 <= row_rank < row_world and the whole scan's point count, after svnicp_set_particles");
  if (bind(c)) return SVNICP_ERR_HIP;
  c->row_rank = row_rank; c->row_world = row_world;
  c->B_total = row_world > 1 ? total_source_points : 0;
  if (row_world > 1) {
    HIPCHK(c, c->rank_sums.ensure((size_t)row_world * c->P * kNSums));
    // a rank whose particle shard is empty in a 2-D split never writes its record: keep it defined
    HIPCHK(c, hipMemsetAsync(c->rank_sums.p, 0, (size_t)row_world * c->P * kNSums * sizeof(double), c->stream));
  }
  return SVNICP_OK;
}

void* svnicp_rank_sums_devptr(svnicp_ctx* c) { return (c && c->row_world > 1) ? (void*)c->rank_sums.p : nullptr; }

// (re)build the target SoA copies in the order the chosen stage-A kernel wants
static int ensure_target_layout(svnicp_ctx* c) {
  int want = 0;
  if (knn_tiles_applicable(c->Mp, c->K)) want = 2;
  else if (knn_scan_plan(c->Mp, c->K, &c->scan_Ms, &c->scan_rank, &c->scan_S2)) want = 1;
  if (c->tune.knn == 0) want = 0;   // option "knn": v1 | v2 | brute | tiles (A/B for tests and profiling)
  if (c->tune.knn == 1) want = knn_scan_plan(c->Mp, c->K, &c->scan_Ms, &c->scan_rank, &c->scan_S2) ? 1 : 0;
  // small registrations (the scan-to-map loop's sizes, BASELINE C1): brute force in one launch, no target layout at all
  if ((c->tune.knn == -1 && knn_brute_applicable(c->B, c->M, c->K)) || (c->tune.knn == 2 && c->K <= 128 && c->M < (1ll << 31))) want = 3;
  c->knn_variant = want;
  c->use_scan = want == 1;
  if (want == 3) return 0;
  const int layout = want == 2 ? 1 : 0;
  if (c->target_layout == layout) return 0;
  const int64_t M = c->M, Mp = c->Mp;
  HIPCHK(c, c->tx.ensure((size_t)Mp)); HIPCHK(c, c->ty.ensure((size_t)Mp)); HIPCHK(c, c->tz.ensure((size_t)Mp));
  HIPCHK(c, c->txf.ensure((size_t)Mp)); HIPCHK(c, c->tyf.ensure((size_t)Mp)); HIPCHK(c, c->tzf.ensure((size_t)Mp));
  HIPCHK(c, c->torig.ensure((size_t)Mp));
  HIPCHK(c, c->emax.ensure(1));
  if (layout == 0) {
    HIPCHK(c, launch_targets_soa2(c->tgt.p, M, Mp, c->tx.p, c->ty.p, c->tz.p, c->txf.p, c->tyf.p, c->tzf.p, c->torig.p,
                                  c->emax.p, c->stream));
  } else {
    const size_t nmax = (size_t)(M > c->B ? M : c->B);
    HIPCHK(c, c->keys_a.ensure(nmax)); HIPCHK(c, c->keys_b.ensure(nmax)); HIPCHK(c, c->vals_a.ensure(nmax));
    HIPCHK(c, c->order_t.ensure((size_t)M)); HIPCHK(c, c->qorder.ensure((size_t)c->B));
    HIPCHK(c, c->bbox.ensure(6));
    HIPCHK(c, c->tile_box.ensure((size_t)6 * (Mp / 512)));
    const size_t tb = sort_temp_bytes(nmax);
    if (tb > c->sort_tmp_bytes) { HIPCHK(c, c->sort_tmp.ensure(tb)); c->sort_tmp_bytes = tb; }
    HIPCHK(c, launch_bbox(c->tgt.p, M, c->bbox.p, c->stream));
    HIPCHK(c, launch_morton_order(c->tgt.p, 0, M, 0, c->pose0, c->bbox.p, c->keys_a.p, c->keys_b.p, c->vals_a.p,
                                  c->order_t.p, c->sort_tmp.p, c->sort_tmp_bytes, c->stream));
    HIPCHK(c, launch_targets_sorted(c->tgt.p, M, Mp, c->order_t.p, c->tx.p, c->ty.p, c->tz.p, c->txf.p, c->tyf.p, c->tzf.p,
                                    c->torig.p, c->tile_box.p, c->emax.p, c->stream));
  }
  c->target_layout = layout;
  return 0;
}

int svnicp_align_begin(svnicp_ctx* c) {
  CTX_CHECK(c);
  if (!c->clouds_set || !c->particles_set)
    return fail(c, SVNICP_ERR_INVALID, "svnicp_align: svnicp_set_clouds and svnicp_set_particles must come first");
  if (bind(c)) return SVNICP_ERR_HIP;
  const int I = c->prm.iterations, P = c->P;
  const int64_t B = c->B;
  c->S = knn_pool_size(c->K);
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if (ensure_target_layout(c)) return c->err.empty() ? SVNICP_ERR_HIP : SVNICP_ERR_HIP;
  if (c->knn_variant == 3) {
    HIPCHK(c, c->fail_count.ensure(1));   // svnicp_get_knn_fallbacks: the brute-force kernel has none (cleared by the begin kernel below)
  } else if (c->knn_variant != 0) {
    if (c->knn_variant == 2) {
      // survivors of the f32 pre-filter: 512 slots per query (median 127 at C3) + a shared arena of 512-slot chunks for the
      // heavy tail (C3: 0.4 % of the queries, 0.17 M entries; C5: 4 %, 3.1 M entries, up to 9016 per query)
      c->scan_S2 = kTilesBase + kTilesChunks * kTilesChunk;
      c->arena_cap = (int)std::max<int64_t>(32768, B / 4);
      HIPCHK(c, c->pool2.ensure((size_t)B * kTilesBase));
      HIPCHK(c, c->arena.ensure((size_t)c->arena_cap * kTilesChunk));
      HIPCHK(c, c->chunk_tab.ensure((size_t)B * kTilesChunks + 16 + (size_t)(B + 63) / 64 + 1));
    } else {
      HIPCHK(c, c->pool2.ensure((size_t)B * c->scan_S2));
    }
    HIPCHK(c, c->fail_list.ensure((size_t)B));
    HIPCHK(c, c->fail_count.ensure(1));
    HIPCHK(c, c->fail_tau.ensure((size_t)B));
    HIPCHK(c, c->qrec.ensure((size_t)B * 6));  // 48-byte records
    HIPCHK(c, c->pool_d.ensure((size_t)kFallbackGrid * 4 * kFallbackQW * c->S));   // fallback rows only
    HIPCHK(c, c->pool_i.ensure((size_t)kFallbackGrid * 4 * kFallbackQW * c->S));
    c->sliced_max = c->tune.fallback_sliced_max >= 0 ? c->tune.fallback_sliced_max : kFallbackSlicedMax;  // 0 forces the list-mode fallback
    if (c->sliced_max > kFallbackSlicedMax) c->sliced_max = kFallbackSlicedMax;
    if (c->sliced_max > 0) {
      HIPCHK(c, c->sl_d.ensure((size_t)c->sliced_max * knn_slice_count(c->K) * c->K));
      HIPCHK(c, c->sl_i.ensure((size_t)c->sliced_max * knn_slice_count(c->K) * c->K));
    }
  } else {
    HIPCHK(c, c->pool_d.ensure((size_t)B * c->S));
    HIPCHK(c, c->pool_i.ensure((size_t)B * c->S));
  }
  HIPCHK(c, c->cand_idx.ensure((size_t)B * c->K));
  HIPCHK(c, c->cand_d2.ensure((size_t)B * c->K));
  c->accum_mode = c->tune.accum;   // option "accum": f64 | valu | split
  HIPCHK(c, c->cmaxb.ensure((size_t)B));
  HIPCHK(c, c->ambig.ensure(2));   // [0] wave steps with an undecided lane, [1] undecided (point, particle) pairs (cleared by the begin kernel below)
  HIPCHK(c, c->history.ensure((size_t)(I > 0 ? I : 1) * 6 * P));
  c->hist_I = I; c->hist_P = P;
  const int nshard = c->p_hi - c->p_lo;
  if (nshard > 0) {
    // small chain (few pairs, one context holds everything, 2 <= P <= 128): decided here, carried by the plan
    Tuning tn = c->tune;
    tn.small_chain = tn.small_chain && P >= 2 && P <= 128 && P <= tn.fused_update_max_p && !tn.update_fused && c->row_world == 1 &&
                     c->p_lo == 0 && c->p_hi == P && !tn.full_corr;
    c->plan = plan_accumulate(nshard, B, c->K, c->num_cus, c->accum_mode, tn);
    if (c->tune.debug)
      fprintf(stderr, "[svnicp] stage-B plan: mode=%d PW=%d WP=%d TP=%d grid=%dx%d tiles/block=%d smem=%zu sgrid=%d pts/block=%d/%d\n", c->plan.f32,
              c->plan.PW, c->plan.WP, c->plan.TP, c->plan.grid_x, c->plan.grid_y, c->plan.tiles_per_block, c->plan.smem,
              c->plan.sgrid_x, c->plan.spts_per_block, c->plan.pts_per_block);
    if (c->plan.smem > 160u * 1024)   // K > 128 runs the LDS-tile VALU search: its smallest tile must fit one CU's LDS
      return fail(c, SVNICP_ERR_INVALID, "svnicp_align: knn_count " + std::to_string(c->K) + " needs " + std::to_string(c->plan.smem) +
                  " bytes of LDS per workgroup (limit 163840): the candidate count is too large for this particle count");
    HIPCHK(c, c->partial.ensure((size_t)std::max(c->plan.grid_x, P == 1 ? single_particle_grid(B) : 0) * c->plan.Ppad * kNSums));
  } else {
    c->plan = AccumPlan{};
    c->plan.f32 = c->accum_mode == 3 ? 1 : c->accum_mode;
  }
  if (c->plan.f32 != 3) HIPCHK(c, c->table.ensure((size_t)B * c->K * 3));  // the split variant gathers from the target cloud
  if (c->plan.f32 == 3) { HIPCHK(c, c->tablea.ensure((size_t)B * 128)); HIPCHK(c, c->anchor.ensure((size_t)B * 3)); }
  if (c->plan.f32 == 3) { HIPCHK(c, c->kbest.ensure((size_t)B * c->plan.Ppad)); HIPCHK(c, c->kidx.ensure((size_t)B * c->plan.Ppad)); }
  else HIPCHK(c, c->tablef.ensure((size_t)B * c->K));
  if (c->prm.record_trace) {
    HIPCHK(c, c->trcorr.ensure((size_t)I * P * B));
    HIPCHK(c, c->trH.ensure((size_t)I * P * 36));
    HIPCHK(c, c->trb.ensure((size_t)I * P * 6));
    HIPCHK(c, c->trN.ensure((size_t)I * P * 6));
    HIPCHK(c, c->trphi.ensure((size_t)I * P * 6));
    HIPCHK(c, c->trh.ensure((size_t)I + 1));
    HIPCHK(c, hipMemsetAsync(c->trcorr.p, 0xff, (size_t)I * P * B * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->trH.p, 0, (size_t)I * P * 36 * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->trb.p, 0, (size_t)I * P * 6 * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->trN.p, 0, (size_t)I * P * 6 * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->trphi.p, 0, (size_t)I * P * 6 * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->trh.p, 0, ((size_t)I + 1) * 8, c->stream));
  }
  c->pused = 0;
  c->single_done_it = -1;
  if (c->median_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));   // a registration that was abandoned between its two per-iteration calls
  c->median_pending = false;
  if (!c->finish_seen && c->prm.mode == SVNICP_MODE_SVGD && c->prm.check_early_stop) {
    // finish_iter_ is sticky across registrations (SVGDICP.cpp:42,128): fold the previous registration's stop flag in before
    // the control words are reset, in case nobody asked for svnicp_get_runtime in between (SVGD mode with early stop only)
    int v[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(v, c->ctl.p, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (v[0]) c->finish_iter = v[1];
    c->finish_seen = true;
  }
  // ONE launch for everything small that a registration starts from: control words {stop flag, finish_iter (SVGDICP.cpp:42)},
  // tickets / counters / pair histogram, the float32 history (SVGDICP.cpp:172-174), statistics counters, a fresh optimizer
  // state (SVGDICP.cpp:73,142-170) — and the total pose of iteration 0 from the CURRENT R_, t_ and R0, t0 (SVNICP.cpp:58-59).
  // (Five fill / copy launches before: 5-8 us each, a tenth of a scan-to-map registration.)
  {
    BeginZero z{};
    auto add = [&](void* p, size_t bytes) { if (p && bytes) { z.ptr[z.n] = static_cast<unsigned int*>(p); z.dwords[z.n] = (unsigned int)(bytes / 4); ++z.n; } };
    add(c->uctl.p, update_uctl_doubles(P) * sizeof(double));
    add(c->history.p, (size_t)(I > 0 ? I : 1) * 6 * P * sizeof(float));
    add(c->ambig.p, 2 * sizeof(int));
    if (c->knn_variant == 3) add(c->fail_count.p, sizeof(int));
    HIPCHK(c, c->small_bar.ensure(8));
    add(c->small_bar.p, 8 * sizeof(unsigned int));
    if (c->prm.mode == SVNICP_MODE_SVGD) add(c->opt.p, (size_t)P * 18 * sizeof(double));
    z.ctl = c->ctl.p; z.iterations = I;
    HIPCHK(c, launch_init_particles(c->init_pose.p, P, c->pose0, 2, c->R.p, c->t.p, c->Rtot.p, c->pose_out.p, 0,
                                    nullptr, c->stream, &z));
  }
  c->particles_dirty = false;
  c->began = true;
  c->finish_seen = false;
  c->have_result = false;
  c->timing_valid = false;
  return SVNICP_OK;
}

// redo the queries listed in fail_list: few -> target-sliced scan + merge, many -> one wave per two queries
static hipError_t launch_fallback(svnicp_ctx* c, KnnArgs a) {
  a.qlist = c->fail_list.p; a.qlist_count = c->fail_count.p; a.list_grid = kFallbackGrid; a.list_qw = kFallbackQW;
  a.slice_max_queries = c->sliced_max; a.slices = 0;
  hipError_t e = launch_knn_topk(a, c->stream);  // returns at once unless the list is longer than slice_max_queries
  if (e != hipSuccess || c->sliced_max <= 0) return e;
  a.slices = knn_slice_count(a.K);
  a.merge_n = 1;
  while (a.merge_n < a.slices * a.K) a.merge_n <<= 1;
  a.sl_d = c->sl_d.p; a.sl_i = c->sl_i.p;
  e = launch_knn_topk(a, c->stream);
  if (e != hipSuccess) return e;
  return launch_knn_merge_slices(a, c->stream);
}

// exact top-K of pose·qsrc[b_lo, b_hi) against the whole target into out_idx / out_d2 ([rows][K]); K = the context's K for
// stage A, K = 1 for the per-particle search of correspondence = full (there the seeded-scan variant, whose plan is built
// for the context's K, is not used)
static int stage_a(svnicp_ctx* c, const double* qsrc, const Pose0& pose, int K, int32_t* out_idx, double* out_d2, int64_t b_lo,
                   int64_t b_hi) {
  KnnArgs a{};
  a.src = qsrc; a.pose = pose; a.tx = c->tx.p; a.ty = c->ty.p; a.tz = c->tz.p; a.torig = c->torig.p;
  a.M = c->M; a.Mp = c->Mp; a.b_lo = b_lo; a.b_hi = b_hi; a.K = K; a.S = knn_pool_size(K);
  a.pool_d = c->pool_d.p; a.pool_i = c->pool_i.p; a.out_idx = out_idx; a.out_d2 = out_d2;
  if (c->knn_variant == 3) {
    KnnBruteArgs k{};
    k.src = qsrc; k.pose = pose; k.tgt = c->tgt.p; k.M = c->M; k.b_lo = b_lo; k.b_hi = b_hi; k.K = K; k.out_idx = out_idx; k.out_d2 = out_d2;
    if (c->tune.debug) {
      if (!c->dbg_phase) HIPCHK(c, hipMalloc(&c->dbg_phase, (8 + 8 * 65536) * sizeof(unsigned long long)));
      HIPCHK(c, hipMemsetAsync(c->dbg_phase, 0, 8 * sizeof(unsigned long long), c->stream));
      k.phase_cycles = c->dbg_phase;
    }
    HIPCHK(c, launch_knn_brute(k, c->num_cus, c->tune.brute_qb, c->stream));
    if (k.phase_cycles) {
      unsigned long long h[8];
      HIPCHK(c, hipMemcpyAsync(h, c->dbg_phase, sizeof h, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const int qb = c->tune.brute_qb > 0 ? c->tune.brute_qb : knn_brute_queries_per_block(b_hi - b_lo, c->num_cus);
      const double nwg = (double)((b_hi - b_lo + qb - 1) / qb);
      fprintf(stderr, "[svnicp] k_knn_brute (%d queries per workgroup) thread-0 cycles per workgroup: pass A %.0f, bound %.0f, pass B %.0f, general path %.0f, rank + write %.0f\n",
              qb, h[0] / nwg, h[1] / nwg, h[2] / nwg, h[3] / nwg, h[4] / nwg);
    }
  } else if (c->knn_variant == 2) {
    const int64_t n = b_hi - b_lo;
    if (n > 0) {
      HIPCHK(c, launch_morton_order(qsrc, b_lo, n, 1, pose, c->bbox.p, c->keys_a.p, c->keys_b.p, c->vals_a.p,
                                    c->qorder.p + b_lo, c->sort_tmp.p, c->sort_tmp_bytes, c->stream));
      KnnTilesArgs k{};
      k.src = qsrc; k.pose = pose; k.qorder = c->qorder.p + b_lo;
      k.tx = c->tx.p; k.ty = c->ty.p; k.tz = c->tz.p; k.txf = c->txf.p; k.tyf = c->tyf.p; k.tzf = c->tzf.p;
      k.torig = c->torig.p; k.tile_box = c->tile_box.p; k.emax_bits = c->emax.p;
      k.M = c->M; k.Mp = c->Mp; k.n_tiles = (int)(c->Mp / 512); k.b_lo = b_lo; k.b_hi = b_hi; k.K = K; k.S2 = c->scan_S2;
      k.pool = c->pool2.p; k.out_idx = out_idx; k.out_d2 = out_d2;
      k.arena = c->arena.p; k.chunk_tab = c->chunk_tab.p; k.arena_cap = c->arena_cap; k.tab_rows = c->B;
      k.scan_split = c->tune.scan_split == 4 ? 4 : 8;
      {  // a small stride coprime to n_groups: 17 sweeps over the curve (C3 1.04 -> 0.98 ms, C5 1.83 -> 1.65 ms against natural order)
        const unsigned int ng = (unsigned int)((n + 63) / 64);
        unsigned int st = c->tune.group_stride > 0 ? (unsigned int)c->tune.group_stride : 17u;
        auto gcd = [](unsigned int x, unsigned int y) { while (y) { const unsigned int t = x % y; x = y; y = t; } return x; };
        while (st > 1 && gcd(st, ng) != 1) st += 2;
        if (ng <= 2 || st >= ng) st = 1;
        k.group_stride = st;
      }
      HIPCHK(c, hipMemsetAsync(c->chunk_tab.p, 0xff, ((size_t)c->B * kTilesChunks + 16 + (size_t)(c->B + 63) / 64 + 1) * sizeof(int32_t), c->stream));
      k.fail_list = c->fail_list.p; k.fail_count = c->fail_count.p; k.fail_tau = c->fail_tau.p; k.qrec = c->qrec.p;
      a.qthr = c->fail_tau.p;
      if (c->prm.record_trace) { HIPCHK(c, c->stat_n.ensure((size_t)c->B)); k.stat_n = c->stat_n.p; }
      HIPCHK(c, hipMemsetAsync(c->fail_count.p, 0, sizeof(int), c->stream));
      unsigned long long*& dbg_phase = c->dbg_phase;
      const size_t dbg_waves = (size_t)((n + 63) / 64) * 4;   // seed kernel: four waves per 64-query group
      if (c->tune.debug && dbg_waves <= 65536) {   // larger launches are simply not instrumented
        if (!dbg_phase) HIPCHK(c, hipMalloc(&dbg_phase, (8 + 8 * 65536) * sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(dbg_phase, 0, (8 + 8 * dbg_waves) * sizeof(unsigned long long), c->stream));
        k.phase_cycles = dbg_phase;
      }
      HIPCHK(c, launch_knn_tiles(k, c->stream));
      if (k.phase_cycles) {
        std::vector<unsigned long long> h(8 + 8 * dbg_waves);
        HIPCHK(c, hipMemcpyAsync(h.data(), dbg_phase, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        fprintf(stderr, "[svnicp] k_knn_tiles wave cycles: rank %llu seed %llu scan %llu barrier waits + hand-over %llu | counts: seed tiles %llu scan tiles %llu scan (query, tile) pairs %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
        // per-wave records of k_knn_seed: [0] rank (wave 0 of a group only) [1] seed [3] barrier waits [5] tile loop [6] K-th bisection
        auto column = [&](const char* tag, int col, unsigned long long floor_) {
          std::vector<unsigned long long> v;
          for (size_t w = 0; w < dbg_waves; ++w) { const unsigned long long x = h[8 + 8 * w + col]; if (x >= floor_) v.push_back(x); }
          if (v.empty()) return;
          std::sort(v.begin(), v.end());
          fprintf(stderr, "[svnicp]   seed kernel %-12s n %6zu  p50 %8llu  p90 %8llu  p99 %8llu  max %8llu cycles\n", tag, v.size(), v[v.size() / 2],
                  v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back());
        };
        column("rank", 0, 5000); column("seed", 1, 1); column("tile loop", 5, 1); column("bisection", 6, 1); column("waits", 3, 0);
      }
      HIPCHK(c, launch_fallback(c, a));
    }
  } else if (c->use_scan && K == c->K) {
    KnnScanArgs k{};
    k.src = qsrc; k.pose = pose; k.tx = c->tx.p; k.ty = c->ty.p; k.tz = c->tz.p;
    k.txf = c->txf.p; k.tyf = c->tyf.p; k.tzf = c->tzf.p; k.torig = c->torig.p; k.emax_bits = c->emax.p;
    k.M = c->M; k.Mp = c->Mp; k.Ms = c->scan_Ms; k.b_lo = b_lo; k.b_hi = b_hi; k.K = K; k.S2 = c->scan_S2;
    k.seed_rank = c->scan_rank; k.pool = c->pool2.p; k.out_idx = out_idx; k.out_d2 = out_d2;
    k.fail_list = c->fail_list.p; k.fail_count = c->fail_count.p;
    HIPCHK(c, hipMemsetAsync(c->fail_count.p, 0, sizeof(int), c->stream));
    HIPCHK(c, launch_knn_scan(k, c->stream));
    // redo the (rare) queries whose seeded threshold was too tight: streaming kernel, list mode
    HIPCHK(c, launch_fallback(c, a));
  } else {
    if (c->knn_variant != 0) return fail(c, SVNICP_ERR_INVALID, "correspondence = full needs knn_count <= 128 (Morton-tile stage A) or knn = v1");
    HIPCHK(c, launch_knn_topk(a, c->stream));
  }
  return SVNICP_OK;
}

int svnicp_stage_candidates(svnicp_ctx* c, int64_t b_lo, int64_t b_hi) {
  CTX_CHECK(c);
  if (!c->began) return fail(c, SVNICP_ERR_INVALID, "svnicp_stage_candidates: call svnicp_align_begin first");
  if (b_lo < 0 || b_hi > c->B || b_lo > b_hi) return fail(c, SVNICP_ERR_INVALID, "svnicp_stage_candidates: bad row range");
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, prof_begin(c, KC_KNN));
  const int rc = stage_a(c, c->src.p, c->pose0, c->K, c->cand_idx.p, c->cand_d2.p, b_lo, b_hi);
  if (rc) return rc;
  if (c->tune.full_corr && c->knn_variant != 0 && c->knn_variant != 3) {   // svnicp_get_knn_fallbacks / _rows describe STAGE A, not the last particle's K = 1 search
    HIPCHK(c, c->stage_fail_count.ensure(1)); HIPCHK(c, c->stage_fail_list.ensure((size_t)c->B));
    HIPCHK(c, hipMemcpyAsync(c->stage_fail_count.p, c->fail_count.p, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->stage_fail_list.p, c->fail_list.p, (size_t)c->B * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(c, prof_end(c));
  return SVNICP_OK;
}


int svnicp_build_candidate_table(svnicp_ctx* c) {
  CTX_CHECK(c);
  if (!c->began) return fail(c, SVNICP_ERR_INVALID, "svnicp_build_candidate_table: call svnicp_align_begin first");
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, prof_begin(c, KC_TABLE));
  if (c->plan.f32 == 3)
    HIPCHK(c, launch_build_table3(c->cand_idx.p, c->B, c->K, c->tgt.p, c->M, c->plan.f32 == 3 ? nullptr : c->table.p,
                                  c->anchor.p, c->tablea.p, c->cmaxb.p, c->stream));
  else
    HIPCHK(c, launch_build_table2(c->cand_idx.p, c->B, c->K, c->tgt.p, c->M, c->table.p, c->tablef.p, c->cmaxb.p, c->stream));
  HIPCHK(c, prof_end(c));
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  c->have_candidates = true;
  return SVNICP_OK;
}

// argument block of the Stein-step kernels for iteration `it`
static UpdateArgs update_args(svnicp_ctx* c, int it) {
  UpdateArgs u{};
  u.sums = c->row_world > 1 ? c->rank_sums.p : c->sums.p; u.n_ranks = c->row_world; u.R = c->R.p; u.t = c->t.p; u.Rtot = c->Rtot.p; u.pose = c->pose0;
  u.P = c->P; u.iteration = it; u.iterations = c->prm.iterations;
  u.lr = c->prm.lr; u.conv_thr = c->prm.convergence_threshold;
  u.check_early_stop = c->prm.check_early_stop; u.full_grad = c->prm.svn_full_grad;
  u.work = c->work.p; u.history = c->history.p; u.pose_out = c->pose_out.p; u.ctl = c->ctl.p;
  if (c->prm.record_trace) {
    u.trH = c->trH.p + (size_t)it * c->P * 36; u.trb = c->trb.p + (size_t)it * c->P * 6;
    u.trN = c->trN.p + (size_t)it * c->P * 6; u.trphi = c->trphi.p + (size_t)it * c->P * 6; u.trh = c->trh.p + it;
  }
  u.eul = c->eul.p; u.opt = c->opt.p; u.optimizer = c->prm.optimizer;
  u.n_src = (double)(c->row_world > 1 ? c->B_total : c->B);   // gradient_scaling_factor_ = the whole scan's size (SVGDICP.cpp:58)
  u.uctl = c->uctl.p;
  u.dbg = c->tune.debug ? c->dbg_upd : nullptr;
  u.svgd = c->prm.mode == SVNICP_MODE_SVGD ? 1 : 0;
  return u;
}
// P = 1 (no pair statistics) and option update=fused: the whole Stein step is one one-workgroup kernel on the main stream
static bool update_one_kernel(const svnicp_ctx* c) { return c->P < 2 || (c->tune.update_fused && c->P <= c->tune.fused_update_max_p); }

// the pair statistics of iteration `it` (bandwidth h from the exact median of the pair distances): they depend on the
// poses only, so they are forked onto the second stream at the START of the iteration and run beside the search and
// accumulate kernels; svnicp_iter_update joins before the Stein direction
// few (point, particle) pairs: the accumulate kernel runs at most kSmallChainBlocks workgroups, nothing reduces their records
// (the prepare lanes add them), and the pair statistics share the prepare kernel's launch on the main stream
static bool small_chain(const svnicp_ctx* c) { return c->plan.f32 == 3 && c->plan.small && c->plan.grid_x <= kSmallChainBlocks; }

// general chain, up to 128 particles (the one-workgroup pair statistics): they run as the last workgroup of the prepare
// kernel's launch on the main stream.  The second stream hid their 12 us behind the search kernel, but its fork and join
// (event record / wait on both sides, one more launch) cost more: C3 7.25 -> 6.99 ms, C2 2.72 -> 2.56 ms per registration,
// and 0.25 ms less host time to enqueue a registration.  Above 128 particles the three-kernel chain stays on the second stream.
constexpr int kMedianInlineMaxP = 128;
static bool median_inline(const svnicp_ctx* c) {
  if (c->P < 2 || c->P > 128 || c->P > c->tune.fused_update_max_p || update_one_kernel(c)) return false;
  return c->tune.median_inline == 1 || (c->tune.median_inline == -1 && c->P <= kMedianInlineMaxP);
}

static int fork_median(svnicp_ctx* c, int it) {
  if (update_one_kernel(c) || c->median_pending || small_chain(c) || median_inline(c)) return SVNICP_OK;
  if (c->tune.debug && !c->dbg_upd) { HIPCHK(c, hipMalloc(&c->dbg_upd, 8 * sizeof(unsigned long long))); HIPCHK(c, hipMemset(c->dbg_upd, 0, 8 * sizeof(unsigned long long))); }
  const UpdateArgs u = update_args(c, it);
  HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
  HIPCHK(c, launch_update_median(u, c->num_cus, c->tune.fused_update_max_p, c->side));
  HIPCHK(c, hipEventRecord(c->ev_join, c->side));
  c->median_pending = true;
  return SVNICP_OK;
}

int svnicp_iter_accumulate(svnicp_ctx* c, int it) {
  CTX_CHECK(c);
  if (!c->began || !c->have_candidates)
    return fail(c, SVNICP_ERR_INVALID, "svnicp_iter_accumulate: candidates not staged");
  if (it < 0 || it >= c->prm.iterations) return fail(c, SVNICP_ERR_INVALID, "svnicp_iter_accumulate: bad iteration");
  if (bind(c)) return SVNICP_ERR_HIP;
  if (const int rc = fork_median(c, it)) return rc;
  const int nshard = c->p_hi - c->p_lo;
  if (nshard <= 0) return SVNICP_OK;
  AccumArgs a{};
  a.src = c->src.p; a.table = c->table.p; a.tablef = c->tablef.p; a.tablea = c->tablea.p; a.kbest = c->kbest.p; a.kidx = c->kidx.p; a.tgt = c->tgt.p; a.cand = c->cand_idx.p; a.anchor = c->anchor.p; a.M = c->M; a.cmax = c->cmaxb.p; a.ambig_count = c->ambig.p;
  a.Rtot = c->Rtot.p; a.B = c->B; a.K = c->K;
  a.p_lo = c->p_lo; a.p_hi = c->p_hi; a.max_dist = c->prm.max_dist; a.partial = c->partial.p; a.ctl = c->ctl.p;
  a.corr = c->prm.record_trace ? c->trcorr.p + (size_t)it * c->P * c->B : nullptr;
  a.svgd = c->prm.mode == SVNICP_MODE_SVGD ? 1 : 0;
  a.fin_iteration = -1;
  if (c->defer_fin && it >= 1) {   // the previous iteration's early-stop decision rides on this iteration's search launch
    const UpdateArgs up = update_args(c, it - 1);
    a.fin_iteration = it - 1; a.fin_P = c->P; a.fin_thr = c->prm.convergence_threshold; a.fin_norms = update_step_norms(up);
    a.fin_pose = c->pose_out.p; a.fin_history = c->history.p; a.fin_ctl = c->ctl.p;
  }
  if (c->tune.full_corr) {
    // correspondence = full (the reference's get_correspondence, SVGDICP.cpp:274-298): every particle's transformed source
    // against the WHOLE target, K = 1 — P exact nearest-neighbour searches per iteration through the stage-A machinery
    if (c->plan.f32 != 3) return fail(c, SVNICP_ERR_INVALID, "correspondence = full needs the split stage B (accum = split, more than 8 particles or knn_count <= 128)");
    HIPCHK(c, c->full_q.ensure((size_t)c->B * 3)); HIPCHK(c, c->full_d2.ensure((size_t)c->B));
    HIPCHK(c, c->full_idx.ensure((size_t)c->P * c->B));
    Pose0 ident{};
    ident.R0[0] = ident.R0[4] = ident.R0[8] = 1.0;
    HIPCHK(c, prof_begin(c, KC_SEARCH));
    for (int p = c->p_lo; p < c->p_hi; ++p) {
      HIPCHK(c, launch_transform_cloud(c->src.p, c->B, c->Rtot.p + 12 * (size_t)p, c->full_q.p, c->ctl.p, c->stream));
      const int rc = stage_a(c, c->full_q.p, ident, 1, c->full_idx.p + (size_t)p * c->B, c->full_d2.p, 0, c->B);
      if (rc) return rc;
    }
    HIPCHK(c, prof_end(c));
    a.full_idx = c->full_idx.p;
  } else if (c->plan.f32 == 3) {
    HIPCHK(c, prof_begin(c, KC_SEARCH));
    HIPCHK(c, launch_search_split(c->plan, a, c->stream));
    HIPCHK(c, prof_end(c));
  }
  // ONE particle, no exchange between ranks ahead, the fused f32 kernel: its last workgroup reduces the partial sums and
  // runs the Stein step (for P = 1 the Newton step and the pose update) — the iteration is this one launch
  const bool single = c->P == 1 && c->prm.mode == SVNICP_MODE_SVN && c->row_world == 1 && c->p_lo == 0 && c->p_hi == 1 &&
                      !c->tune.full_corr && c->tune.single_fused && accumulate_can_fuse_single(c->plan);
  const UpdateArgs us = update_args(c, it);
  a.ticket = reinterpret_cast<unsigned int*>(c->uctl.p + 41);
  HIPCHK(c, prof_begin(c, KC_ACCUM));
  HIPCHK(c, launch_accumulate(c->plan, a, single ? &us : nullptr, c->stream));
  HIPCHK(c, prof_end(c));
  if (single) { c->single_done_it = it; return SVNICP_OK; }
  if (small_chain(c)) return SVNICP_OK;   // the update kernels add the workgroups' records themselves
  HIPCHK(c, prof_begin(c, KC_REDUCE));
  // one rank: the particle's record; source-row sharding: this rank's slot of the [row_world][P][22] array
  double* rec = c->row_world > 1 ? c->rank_sums.p + (size_t)c->row_rank * c->P * kNSums : c->sums.p;
  HIPCHK(c, launch_reduce_partials(c->partial.p, c->plan.grid_x, c->plan.Ppad, c->p_lo, nshard, rec, c->ctl.p, c->stream));
  HIPCHK(c, prof_end(c));
  return SVNICP_OK;
}

int svnicp_iter_update(svnicp_ctx* c, int it) {
  CTX_CHECK(c);
  if (!c->began) return fail(c, SVNICP_ERR_INVALID, "svnicp_iter_update: call svnicp_align_begin first");
  if (it < 0 || it >= c->prm.iterations) return fail(c, SVNICP_ERR_INVALID, "svnicp_iter_update: bad iteration");
  if (bind(c)) return SVNICP_ERR_HIP;
  if (c->single_done_it == it) { c->single_done_it = -1; return SVNICP_OK; }   // done by the accumulate kernel's last workgroup
  if (c->tune.debug && !c->dbg_upd) { HIPCHK(c, hipMalloc(&c->dbg_upd, 8 * sizeof(unsigned long long))); HIPCHK(c, hipMemset(c->dbg_upd, 0, 8 * sizeof(unsigned long long))); }
  UpdateArgs u = update_args(c, it);
  HIPCHK(c, prof_begin(c, KC_UPDATE));
  if (c->tune.debug && it == c->prm.iterations - 1) {   // debug option: phase cycles of the one-workgroup kernels so far
    unsigned long long h[8];
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->side));
    HIPCHK(c, hipMemcpy(h, c->dbg_upd, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[svnicp] Stein step thread-0 cycles (summed over launches so far; fused kernel: prepare / median / direction / pose / tail; k_upd_median: state / histogram / bin scan / collect / rank): %llu %llu %llu %llu %llu\n", h[0], h[1], h[2], h[3], h[4]);
  }
  if (update_one_kernel(c)) {
    u.svgd = 0;   // the one-workgroup kernels are per mode
    if (c->prm.mode == SVNICP_MODE_SVGD) HIPCHK(c, launch_update_svgd(u, c->stream));
    else HIPCHK(c, launch_update(u, c->stream));
  } else if (small_chain(c)) {
    u.sums = c->partial.p; u.n_ranks = c->plan.grid_x; u.sums_stride = c->plan.Ppad * kNSums; u.sums_out = c->sums.p;
    HIPCHK(c, launch_update_prepare_median(u, c->stream));
    HIPCHK(c, launch_update_direction(u, c->stream, !(c->defer_fin && it < c->prm.iterations - 1)));
  } else if (median_inline(c)) {
    HIPCHK(c, launch_update_prepare_median(u, c->stream));
    HIPCHK(c, launch_update_direction(u, c->stream, !(c->defer_fin && it < c->prm.iterations - 1)));
  } else {
    // pair statistics: forked at the start of the iteration; a caller that skipped svnicp_iter_accumulate gets them here
    if (const int rc = fork_median(c, it)) return rc;
    HIPCHK(c, launch_update_prepare(u, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    c->median_pending = false;
    HIPCHK(c, launch_update_direction(u, c->stream, !(c->defer_fin && it < c->prm.iterations - 1)));
  }
  HIPCHK(c, prof_end(c));
  return SVNICP_OK;
}

int svnicp_finish(svnicp_ctx* c) {
  CTX_CHECK(c);
  if (!c->began) return fail(c, SVNICP_ERR_INVALID, "svnicp_finish: call svnicp_align_begin first");
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  StatsArgs s{c->pose_out.p, c->P, c->prm.mode, c->stats.p};
  HIPCHK(c, launch_stats(s, c->stream));
  // the result block follows the kernels down the stream into pinned memory: the getters then cost no GPU round trip
  c->host_stats_valid = false;
  if (c->h_stats && (size_t)c->P + 48 <= c->h_stats_cap)
    HIPCHK(c, hipMemcpyAsync(c->h_stats, c->stats.p, ((size_t)c->P + 48) * 8, hipMemcpyDeviceToHost, c->stream));
  c->have_result = true;
  c->timing_valid = true;
  return SVNICP_OK;
}

int svnicp_stopped(svnicp_ctx* c) {
  CTX_CHECK(c);
  if (bind(c)) return SVNICP_ERR_HIP;
  int v[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(v, c->ctl.p, sizeof v, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return v[0] ? 1 : 0;
}

void* svnicp_candidates_devptr(svnicp_ctx* c) { return c ? (void*)c->cand_idx.p : nullptr; }
void* svnicp_sums_devptr(svnicp_ctx* c) { return c ? (void*)c->sums.p : nullptr; }

// follow_stop: the caller is going to wait for the result anyway (svnicp_align) — with early stop on, the host then enqueues
// the iterations in chunks and waits for the stop flag of the chunk before the previous one before it goes on: after the
// device has stopped it enqueues at most two more chunks of launches that return at once, instead of all the remaining
// iterations (the shipped configurations run 100 iterations with early stop and stop after 30-50: 350 empty launches were
// 1.1 ms of a 2.3 ms registration)
static int align_enqueue(svnicp_ctx* c, bool follow_stop) {
  CTX_CHECK(c);
  if (c->prm.mode == SVNICP_MODE_SVGD && (c->prm.optimizer < 0 || c->prm.optimizer > 3))
    return SVNICP_NO_OPTIMIZER;  // set_optimizer() found no optimizer: stein_align returns at once (SVGDICP.cpp:73-75)
  int rc = svnicp_align_begin(c);
  if (rc) return rc;
  if ((rc = svnicp_stage_candidates(c, 0, c->B))) return rc;
  if ((rc = svnicp_build_candidate_table(c))) return rc;
  c->small_launched = false;
  if (small_chain(c) && c->tune.persistent && !c->prm.record_trace && !c->profile && c->prm.iterations > 0 &&
      small_registration_supported(c->plan.PW, c->plan.WP, c->K)) {
    // all iterations in ONE cooperative launch (particle_update.hip: k_small_registration); anything the runtime refuses
    // (no cooperative launch, grid not resident) falls back to the four launches per iteration
    AccumArgs a{};
    a.src = c->src.p; a.tablea = c->tablea.p; a.kbest = c->kbest.p; a.kidx = c->kidx.p; a.tgt = c->tgt.p; a.cand = c->cand_idx.p;
    a.anchor = c->anchor.p; a.M = c->M; a.cmax = c->cmaxb.p; a.ambig_count = c->ambig.p;
    a.Rtot = c->Rtot.p; a.B = c->B; a.K = c->K; a.p_lo = c->p_lo; a.p_hi = c->p_hi; a.max_dist = c->prm.max_dist;
    a.partial = c->partial.p; a.ctl = c->ctl.p; a.svgd = c->prm.mode == SVNICP_MODE_SVGD ? 1 : 0;
    if (c->tune.debug && !c->dbg_upd) { HIPCHK(c, hipMalloc(&c->dbg_upd, 8 * sizeof(unsigned long long))); }
    if (c->tune.debug) HIPCHK(c, hipMemsetAsync(c->dbg_upd, 0, 8 * sizeof(unsigned long long), c->stream));
    UpdateArgs u = update_args(c, 0);
    u.sums_out = c->sums.p;
    const hipError_t e = launch_small_registration(c->plan, a, u, c->prm.iterations, c->small_bar.p, c->num_cus, c->stream);
    if (e == hipSuccess && c->tune.debug) {
      unsigned long long h[8];
      HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipMemcpy(h, c->dbg_upd, sizeof h, hipMemcpyDeviceToHost));
      fprintf(stderr, "[svnicp] k_small_registration, workgroup 0, cycles over %d iterations: search %llu | barrier %llu | accumulate %llu | barrier %llu | prepare %llu | barrier %llu | direction %llu | barrier %llu\n",
              c->prm.iterations, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    if (e == hipSuccess) { c->small_launched = true; return svnicp_finish(c); }
    (void)hipGetLastError();
  }
  constexpr int kChunk = 4;   // (2: 1.29 ms at the shipped settings against 1.26 — the host then waits more often than it saves launches)
  const bool follow = follow_stop && c->prm.check_early_stop && c->prm.iterations > 2 * kChunk;
  if (follow && !c->h_flags) {
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_flags), 3 * sizeof(int), hipHostMallocDefault));
    for (auto& e : c->ev_chunk) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  // early stop: iteration i's decision is taken by iteration i + 1's search kernel (the last iteration keeps k_upd_finish)
  c->defer_fin = c->prm.check_early_stop && !c->prm.record_trace && c->plan.f32 == 3 && !c->tune.full_corr && c->P >= 2 && !update_one_kernel(c);
  struct Reset { bool& f; ~Reset() { f = false; } } reset_defer{c->defer_fin};
  for (int it = 0; it < c->prm.iterations; ++it) {
    if ((rc = svnicp_iter_accumulate(c, it))) return rc;
    if ((rc = svnicp_iter_update(c, it))) return rc;
    if (follow && (it + 1) % kChunk == 0) {
      const int chunk = it / kChunk, slot = chunk % 3;
      HIPCHK(c, hipMemcpyAsync(&c->h_flags[slot], c->ctl.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipEventRecord(c->ev_chunk[slot], c->stream));
      if (chunk >= 1) {
        const int prev = (chunk - 1) % 3;
        HIPCHK(c, hipEventSynchronize(c->ev_chunk[prev]));
        if (c->h_flags[prev]) break;   // stopped: what is already enqueued returns at once, nothing more is needed
      }
    }
  }
  return svnicp_finish(c);
}
int svnicp_align_async(svnicp_ctx* c) { return align_enqueue(c, false); }

// the persistent kernel's barrier gives up after a bounded wait and says so in its error word
static int check_small_kernel(svnicp_ctx* c) {
  if (!c->small_launched) return SVNICP_OK;
  c->small_launched = false;
  unsigned int w[2] = {0u, 0u};
  HIPCHK(c, hipMemcpyAsync(w, c->small_bar.p, sizeof w, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (w[1] != 0u) { c->have_result = false; return fail(c, SVNICP_ERR_HIP, "svnicp_align: the small-registration kernel's grid barrier timed out (chain=persistent is an option; the default launches do not wait on each other)"); }
  return SVNICP_OK;
}

int svnicp_align(svnicp_ctx* c) {
  CTX_CHECK(c);
  if (c->shard_set && (c->p_lo != 0 || c->p_hi != c->P))
    return fail(c, SVNICP_ERR_INVALID, "svnicp_align: a particle shard is set; drive the split-phase calls instead");
  if (c->row_world > 1)
    return fail(c, SVNICP_ERR_INVALID, "svnicp_align: a source-row shard is set (this context holds a partial record); drive the split-phase calls instead");
  int rc = align_enqueue(c, true);
  if (rc) return rc;  // negative status, or SVNICP_NO_OPTIMIZER
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if ((rc = check_small_kernel(c))) return rc;
  c->host_stats_valid = c->h_stats != nullptr;
  return SVNICP_ALIGN_SUCCESS;
}

static int fetch(svnicp_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SVNICP_OK;
}
#define NEED_RESULT(c)                                                                   \
  do {                                                                                   \
    CTX_CHECK(c);                                                                        \
    if (!(c)->have_result) return fail((c), SVNICP_ERR_INVALID, "no registration result yet"); \
  } while (0)

// mean[6] var[6] cov[36] weights[P]: from the pinned copy svnicp_finish queued when it has landed, else from the device
static int fetch_stats(svnicp_ctx* c, void* dst, size_t off_doubles, size_t n_doubles) {
  if (c->host_stats_valid) { std::memcpy(dst, c->h_stats + off_doubles, n_doubles * 8); return SVNICP_OK; }
  return fetch(c, dst, c->stats.p + off_doubles, n_doubles * 8);
}
int svnicp_get_transformation(svnicp_ctx* c, double out6[6]) { NEED_RESULT(c); return fetch_stats(c, out6, 0, 6); }
int svnicp_get_distribution(svnicp_ctx* c, double out6[6]) { NEED_RESULT(c); return fetch_stats(c, out6, 6, 6); }
int svnicp_get_cov_matrix(svnicp_ctx* c, double out36[36]) { NEED_RESULT(c); return fetch_stats(c, out36, 12, 36); }
int svnicp_get_particle_weight(svnicp_ctx* c, double* outP) {
  NEED_RESULT(c);
  return fetch_stats(c, outP, 48, (size_t)c->P);
}
int svnicp_get_particles(svnicp_ctx* c, double* out6P) {
  CTX_CHECK(c);
  if (!c->particles_set) return fail(c, SVNICP_ERR_INVALID, "no particles set");
  return fetch(c, out6P, c->pose_out.p, (size_t)c->P * 48);
}
int svnicp_get_particle_history(svnicp_ctx* c, float* out) {
  NEED_RESULT(c);
  return fetch(c, out, c->history.p, (size_t)c->hist_I * 6 * c->hist_P * sizeof(float));
}

int svnicp_get_gpu_ms(svnicp_ctx* c, double out3[3]) {
  NEED_RESULT(c);
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float a = 0, b = 0;
  HIPCHK(c, hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
  HIPCHK(c, hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
  out3[0] = a; out3[1] = b; out3[2] = (double)a + b;
  return SVNICP_OK;
}

int svnicp_get_runtime(svnicp_ctx* c, double out3[3]) {
  NEED_RESULT(c);
  double ms[3];
  int rc = svnicp_get_gpu_ms(c, ms);   // synchronises the stream
  if (rc) return rc;
  if (c->prm.mode == SVNICP_MODE_SVGD && c->prm.check_early_stop && !c->finish_seen) {
    // finish_iter_ = epoch + 1 on an SVGD-mode early stop (SVGDICP.cpp:128).  Read here rather than in svnicp_align so that
    // svnicp_align_async + svnicp_synchronize and the split-phase sequence (svnicp_iter_update … svnicp_finish) report it too
    int v[2] = {0, 0};
    HIPCHK(c, hipMemcpy(v, c->ctl.p, sizeof v, hipMemcpyDeviceToHost));
    if (v[0]) c->finish_iter = v[1];
    c->finish_seen = true;
  }
  // finish_iter_: SVNICP::stein_align never touches it (SVNICP.cpp:95-101 only breaks), SVGDICP::stein_align sets it on
  // an early stop and nothing resets it (SVGDICP.cpp:42,128)
  out3[0] = ms[0] * 1e-3; out3[1] = ms[1] * 1e-3; out3[2] = (double)c->finish_iter;
  return SVNICP_OK;
}

int svnicp_get_iterations_run(svnicp_ctx* c, int* out) {
  NEED_RESULT(c);
  if (!out) return SVNICP_ERR_INVALID;
  int v[2];
  const int rc = fetch(c, v, c->ctl.p, sizeof v);
  if (rc) return rc;
  *out = v[1];
  return SVNICP_OK;
}

int svnicp_get_knn_fallbacks(svnicp_ctx* c, int* out) {
  CTX_CHECK(c);
  if (!c->have_candidates) return fail(c, SVNICP_ERR_INVALID, "no candidates yet");
  if (c->knn_variant == 0) { *out = -1; return SVNICP_OK; }
  return fetch(c, out, (c->tune.full_corr && c->stage_fail_count.p) ? c->stage_fail_count.p : c->fail_count.p, sizeof(int));
}

int svnicp_get_knn_fallback_rows(svnicp_ctx* c, int32_t* out, int cap, int* n_out) {
  CTX_CHECK(c);
  if (!c->have_candidates || !n_out) return fail(c, SVNICP_ERR_INVALID, "no candidates yet");
  *n_out = 0;
  if (c->knn_variant == 0) return SVNICP_OK;
  const bool snap = c->tune.full_corr && c->stage_fail_count.p;
  int n = 0;
  int rc = fetch(c, &n, snap ? c->stage_fail_count.p : c->fail_count.p, sizeof(int));
  if (rc) return rc;
  *n_out = n;
  if (n > cap) n = cap;
  if (n > 0 && out) return fetch(c, out, snap ? c->stage_fail_list.p : c->fail_list.p, (size_t)n * 4);
  return SVNICP_OK;
}

int svnicp_get_knn_survivors(svnicp_ctx* c, int32_t* outB) {
  CTX_CHECK(c);
  if (!c->have_candidates || c->knn_variant != 2 || !c->prm.record_trace || !c->stat_n.p)
    return fail(c, SVNICP_ERR_INVALID, "svnicp_get_knn_survivors: needs record_trace and the pruned stage-A kernel");
  return fetch(c, outB, c->stat_n.p, (size_t)c->B * 4);
}

int svnicp_get_ambiguous_steps(svnicp_ctx* c, int* out) {
  CTX_CHECK(c);
  if (!c->have_result) return fail(c, SVNICP_ERR_INVALID, "no registration result yet");
  if (c->accum_mode == 0) { *out = -1; return SVNICP_OK; }
  return fetch(c, out, c->ambig.p, sizeof(int));
}

int svnicp_get_ambiguous_pairs(svnicp_ctx* c, int64_t* out) {
  CTX_CHECK(c);
  if (!c->have_result || !out) return fail(c, SVNICP_ERR_INVALID, "no registration result yet");
  if (c->accum_mode != 3 || c->plan.f32 != 3) { *out = -1; return SVNICP_OK; }   // counted by the bf16 search kernel only
  int v[2] = {0, 0};
  const int rc = fetch(c, v, c->ambig.p, sizeof v);
  *out = v[1];
  return rc;
}

int svnicp_set_profile(svnicp_ctx* c, int on) {
  CTX_CHECK(c);
  c->profile = on != 0;
  c->profile_mask = on == 1 ? ~0u : ((unsigned)on >> 1);  // 1 = every class, else bit (class + 1) selects a class
  return SVNICP_OK;
}

int svnicp_get_kernel_ms(svnicp_ctx* c, double* ms5, int32_t* launches5) {  // SVNICP_KERNEL_CLASSES entries each
  NEED_RESULT(c);
  if (!c->profile) return fail(c, SVNICP_ERR_INVALID, "svnicp_get_kernel_ms: profiling is off (svnicp_set_profile)");
  if (bind(c)) return SVNICP_ERR_HIP;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < KC_COUNT; ++i) { ms5[i] = 0.0; launches5[i] = 0; }
  for (size_t i = 0; i < c->pused; ++i) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->pev[2 * i], c->pev[2 * i + 1]));
    ms5[c->pcls[i]] += ms;
    launches5[c->pcls[i]] += 1;
  }
  return SVNICP_OK;
}

int svnicp_get_candidates(svnicp_ctx* c, int32_t* out) {
  CTX_CHECK(c);
  if (!c->have_candidates) return fail(c, SVNICP_ERR_INVALID, "no candidates yet");
  return fetch(c, out, c->cand_idx.p, (size_t)c->B * c->K * 4);
}
int svnicp_get_candidate_dist2(svnicp_ctx* c, double* out) {
  CTX_CHECK(c);
  if (!c->have_candidates) return fail(c, SVNICP_ERR_INVALID, "no candidates yet");
  return fetch(c, out, c->cand_d2.p, (size_t)c->B * c->K * 8);
}

int svnicp_get_trace(svnicp_ctx* c, int32_t* corr, double* H, double* b, double* N, double* phi, double* h) {
  NEED_RESULT(c);
  if (!c->prm.record_trace) return fail(c, SVNICP_ERR_INVALID, "svnicp_get_trace: params.record_trace was 0");
  const size_t I = (size_t)c->prm.iterations, P = (size_t)c->P;
  int rc = 0;
  if (corr && (rc = fetch(c, corr, c->trcorr.p, I * P * (size_t)c->B * 4))) return rc;
  if (H && (rc = fetch(c, H, c->trH.p, I * P * 36 * 8))) return rc;
  if (b && (rc = fetch(c, b, c->trb.p, I * P * 6 * 8))) return rc;
  if (N && (rc = fetch(c, N, c->trN.p, I * P * 6 * 8))) return rc;
  if (phi && (rc = fetch(c, phi, c->trphi.p, I * P * 6 * 8))) return rc;
  if (h && (rc = fetch(c, h, c->trh.p, I * 8))) return rc;
  return SVNICP_OK;
}

}  // extern "C"
