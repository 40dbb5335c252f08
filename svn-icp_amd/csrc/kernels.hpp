// kernels.hpp — argument blocks and host launchers of the HIP kernels (internal to libsvnicp_hip.so)
#pragma once
#include "device_math.hpp"

namespace svnicp {

// ---------------- Stage A (knn_topk.hip) ----------------
struct KnnArgs {
  const double* src;  // [B][3] source cloud (un-transformed)
  Pose0 pose;         // R0, t0
  const double *tx, *ty, *tz;  // target SoA in permuted order, padded to Mp with NaN
  const int32_t* torig;        // [Mp] original target index of each slot
  int64_t M, Mp;
  int64_t b_lo, b_hi;  // query rows handled by this launch
  int K, S;            // S = pool capacity (power of two >= K + 128)
  double* pool_d;      // [B][S]
  int32_t* pool_i;     // [B][S]
  int32_t* out_idx;    // [B][K]
  double* out_d2;      // [B][K]
  const int32_t* qlist;     // list mode (fallback of k_knn_scan): query indices, else nullptr
  const int* qlist_count;   // device scalar: number of entries of qlist
  int list_grid;            // list mode: number of workgroups
  int list_qw;              // list mode: queries per wave chunk (pool rows = list_grid*4*list_qw)
  // sliced list mode (lists of at most slice_max_queries): every listed query is scanned by `slices` waves over
  // disjoint target ranges; per-slice top-K lists in sl_d/sl_i [slice_max_queries][slices][K], merged by
  // k_knn_merge_slices (merge_n = power of two >= slices*K).  slices = 0: plain list mode.
  int slices, slice_max_queries, merge_n;
  double* sl_d;
  int32_t* sl_i;
  const double* qthr;  // sliced list mode, optional: starting threshold of each listed query (from k_knn_tiles)
};
int knn_pool_size(int K);
int knn_slice_count(int K);
hipError_t launch_knn_merge_slices(const KnnArgs& a, hipStream_t st);
int64_t knn_padded_targets(int64_t M);

// ---------------- Stage A for small registrations (knn_brute.hip) ----------------
struct KnnBruteArgs {
  const double* src;   // [B][3] source cloud (un-transformed)
  Pose0 pose;          // R0, t0
  const double* tgt;   // [M][3] target cloud as given (no re-ordered copy)
  int64_t M;
  int64_t b_lo, b_hi;  // query rows of this launch
  int K;
  int32_t* out_idx;    // [B][K]
  double* out_d2;      // [B][K]
  unsigned long long* phase_cycles;   // optional [8]: thread-0 cycles per phase, summed over the workgroups (option debug)
};
bool knn_brute_applicable(int64_t B, int64_t M, int K);
int knn_brute_queries_per_block(int64_t n, int num_cus);
// queries_per_block: 0 = knn_brute_queries_per_block(n, num_cus); 1..6 = as given (option brute_qb, A/B)
hipError_t launch_knn_brute(const KnnBruteArgs& a, int num_cus, int queries_per_block, hipStream_t st);

// ---------------- Stage A fast variant (knn_scan.hip) ----------------
struct KnnScanArgs {
  const double* src;
  Pose0 pose;
  const double *tx, *ty, *tz;     // f64 permuted SoA (exact distances of the survivors)
  const float *txf, *tyf, *tzf;   // f32 copies (pre-filter)
  const int32_t* torig;
  const unsigned long long* emax_bits;  // max |target coordinate| as the bits of a non-negative double
  int64_t M, Mp, Ms;              // Ms = slots scanned in the seed phase
  int64_t b_lo, b_hi;
  int K, S2, seed_rank;
  int32_t* pool;                  // [B][S2] slots
  int32_t* out_idx;               // [B][K]
  double* out_d2;                 // [B][K]
  int32_t* fail_list;             // [B]
  int* fail_count;
};
bool knn_scan_plan(int64_t Mp, int K, int64_t* Ms, int* seed_rank, int* S2);
hipError_t launch_targets_soa2(const double* tgt, int64_t M, int64_t Mp, double* tx, double* ty, double* tz,
                               float* txf, float* tyf, float* tzf, int32_t* torig, unsigned long long* emax_bits,
                               hipStream_t st);
hipError_t launch_knn_scan(const KnnScanArgs& a, hipStream_t st);
hipError_t launch_knn_topk(const KnnArgs& a, hipStream_t st);
hipError_t launch_build_table(const int32_t* idx, int64_t n_entries, const double* tgt, double* table,
                              hipStream_t st);

// ---------------- Stage A pruned variant (spatial_prep.hip, knn_tiles.hip) ----------------
struct KnnTilesArgs {
  const double* src;
  Pose0 pose;
  const int32_t* qorder;          // [b_hi-b_lo] original query rows in Morton order
  const double *tx, *ty, *tz;     // Morton-ordered target SoA (f64), NaN padded to Mp
  const float *txf, *tyf, *tzf;   // f32 copies
  const int32_t* torig;
  const float* tile_box;          // [6][n_tiles]: lo xyz, hi xyz (outward rounded)
  const unsigned long long* emax_bits;
  int64_t M, Mp;
  int n_tiles;
  int64_t b_lo, b_hi;
  int K, S2;                      // S2 = most survivors a query may hold: kTilesBase + kTilesChunks * kTilesChunk
  int32_t* pool;                  // [B][kTilesBase]: the first survivors of each query position
  int32_t* arena;                 // [arena_cap][kTilesChunk]: overflow chunks, handed out on demand to the few heavy queries
  int32_t* chunk_tab;             // [B][kTilesChunks] chunk id per (query position, chunk), -1 = none; [B*kTilesChunks] = allocation counter (starts at -1)
  int arena_cap;
  int64_t tab_rows;               // B of the allocation (rows of chunk_tab); after the table: the chunk allocation counter
  int scan_split;                 // waves per 64-query workgroup of the scan kernel: 4 or 8
  unsigned int group_stride;      // scan kernel: workgroup i serves group (i * group_stride) mod n_groups (coprime to n_groups)
  int32_t* out_idx;
  double* out_d2;
  int32_t* fail_list;
  int* fail_count;
  int32_t* stat_n;                // optional [B]: survivors of the f32 filter per query (first pass)
  unsigned long long* phase_cycles;  // optional [8]: summed wave cycles per phase (debug, SVNICP_DEBUG)
  void* qrec;         // [B] 48-byte per-query records (position, threshold, survivor count) between the two kernels
  double* fail_tau;   // optional [B]: a valid threshold (>= K-th distance) of each failed query, +inf if none
};
constexpr int kTilesBase = 512, kTilesChunk = 512, kTilesChunks = 31;   // 512 + 31 * 512 = 16384 survivors per query at most
bool knn_tiles_applicable(int64_t Mp, int K);
hipError_t launch_knn_tiles(const KnnTilesArgs& a, hipStream_t st);
size_t sort_temp_bytes(size_t n);
hipError_t launch_bbox(const double* pts, int64_t n, unsigned long long* bbox, hipStream_t st);
hipError_t launch_morton_order(const double* pts, int64_t i0, int64_t n, int transform, const Pose0& pose,
                               const unsigned long long* bbox, unsigned int* keys_a, unsigned int* keys_b,
                               int32_t* vals_a, int32_t* order, void* temp, size_t temp_bytes, hipStream_t st);
hipError_t launch_targets_sorted(const double* tgt, int64_t M, int64_t Mp, const int32_t* order, double* tx, double* ty,
                                 double* tz, float* txf, float* tyf, float* tzf, int32_t* torig, float* tile_box,
                                 unsigned long long* emax_bits, hipStream_t st);

// ---------------- Stage B (stein_iter.hip) ----------------
struct AccumArgs {
  const double* src;    // [B][3]
  const double* table;  // [B][K][3] candidate coordinates (f64, absolute)
  const float4* tablef; // [B][K] float32 local coordinates + |c'|² (fast variant)
  const float4* tablea; // [B][2][64] float32 local rows in MFMA A-operand order (stein_split.hip: k_build_table3)
  const float* cmax;    // [B] max |c'| per source point (fast variants)
  int* ambig_count;     // optional statistic: wave steps that took the exact path, or nullptr
  const double* Rtot;   // [P][12]: R_total row-major (9) + t_total (3)
  int64_t B;
  int K, RS;            // RS = LDS row stride in doubles (odd)
  int p_lo, p_hi;       // particle shard of this GPU
  int TP;               // source points per LDS tile
  int tiles_per_block;
  int64_t n_tiles;
  int Ppad;             // padded shard size = gridDim.y * particles per workgroup
  double max_dist;
  double* partial;      // [gridDim.x][Ppad][kNSums]
  const int* ctl;       // ctl[0] = stop flag
  int32_t* corr;        // optional trace [P][B] (this iteration), or nullptr
  int svgd;             // SVGD-ICP mode: slot 4 of the sums counts non-zero rows (SVGDICP.cpp:404)
  // split variant: no f64 candidate table — winners are gathered as tgt[cand[b][k]] (6 MB + 4 B/candidate instead of
  // 24 B/candidate of HBM traffic), the local frame's origin comes from anchor[b]
  const double* tgt;    // [M][3]
  const int32_t* cand;  // [B][K] candidate indices (stage A)
  const double* anchor; // [B][3] = tgt[cand[b][0]]
  int64_t M;
  uint8_t* kbest;       // split variant: winner slot per (source point, particle of the shard), [B][Ppad]
  int32_t* kidx;        // … and the winner's TARGET index cand[b][slot], [B][Ppad]: the accumulate kernel's gather then starts one
                        // dependent load later (byte -> index -> coordinates becomes index -> coordinates)
  const int32_t* full_idx;  // correspondence = full: nearest target index of every (particle, source point), [P][B]; else nullptr
  int pts_per_block, spts_per_block;  // split variant: source points per workgroup (accumulate / search kernel)
  unsigned int* ticket;               // fused one-particle iteration: arrival counter of the accumulate kernel's workgroups
  // the PREVIOUS iteration's early-stop decision, taken by workgroup (0, 0) of the search kernel instead of a k_upd_finish
  // launch of its own (svnicp_align's loop; fin_iteration < 0: nothing to decide)
  int fin_iteration, fin_P;
  double fin_thr;
  const double* fin_norms;            // [P] step norms of that iteration (uctl + UCTL_NORM)
  const double* fin_pose;             // [6][P] pose_out
  float* fin_history;                 // [I][6][P]
  int* fin_ctl;                       // [0] stop flag, [1] finish_iter
};
struct AccumPlan { int PW, WP, TP, grid_x, grid_y, tiles_per_block, Ppad, RS, f32, K, sgrid_x, pts_per_block, spts_per_block; int64_t n_tiles; size_t smem;
                   int small; /* split variant, few (point, particle) pairs: at most kSmallChainBlocks accumulate workgroups, see api.hip small_chain */ };
constexpr int kSmallChainBlocks = 32;
// f32: 0 = float64 baseline, 1 = float32 VALU search (fused with the accumulation), 3 = bf16 matrix-pipe search kernel +
// accumulation kernel (falls back to 1 when K > 128 or the shard has <= 8 particles)
// test / profiling knobs of a context (svnicp_set_option); the defaults are the product configuration
struct Tuning {
  int knn = -1;                  // stage A kernel: -1 automatic, 0 streaming only (v1), 1 seeded scan (v2), 2 brute force (small sizes), 3 Morton tiles even where brute force applies
  int fallback_sliced_max = -1;  // stage A: failed queries redone by target slices up to this many (-1 default)
  int accum = 3;                 // stage B: 0 f64 baseline, 1 f32 VALU search (fused), 3 search + accumulate kernels
  int update_fused = 0;          // Stein update: 1 = one fused kernel for 2 <= P <= fused_update_max_p
  int fused_update_max_p = 128;  // above this the Stein step runs as workgroup-parallel kernels
  int wgpcu_search = 0, wgpcu_accum = 0;   // workgroups per CU the stage-B grids are sized for (0 = automatic)
  int tp = 0;                    // fused stage-B variants: source points per LDS tile (0 = automatic)
  int accum_min_steps = 0;       // least wave steps per accumulate workgroup (0 = default 4)
  int group_stride = 0;          // stage A scan: group order stride (0 = default, 1 = natural order)
  int scan_split = 0;            // stage A scan: waves per 64-query workgroup, 4 or 8 (0 = default)
  int debug = 0;                 // print plans and per-phase cycle counters to stderr
  int single_fused = 1;          // one particle: reduce + Stein step in the accumulate kernel's last workgroup (0: three launches, A/B)
  int small_chain = 1;           // small registrations: no k_reduce_partials, Stein-step front in one launch on the main stream (0: the general chain, A/B)
  int persistent = 0;            // 1: svnicp_align runs all iterations of a small-chain registration in ONE cooperative launch (k_small_registration;
                                 // measured SLOWER than the four launches per iteration on this eight-XCD part: off by default, option chain=persistent)
  int median_inline = -1;        // pair statistics in the prepare kernel's launch on the main stream also in the general chain: -1 automatic (P <= 128), 0 never (second stream), 1 the same as automatic
  int brute_qb = 0;              // brute-force stage A: queries per workgroup, 0 automatic (knn_brute_queries_per_block)
  int full_corr = 0;             // 1: correspondence = full — per-particle exact NN over the whole target (SVGDICP.cpp:274-298)
};
AccumPlan plan_accumulate(int n_particles, int64_t B, int K, int num_cus, int f32, const Tuning& tune);
hipError_t launch_search_split(const AccumPlan& plan, AccumArgs a, hipStream_t st);         // split variant, kernel 1
hipError_t launch_accumulate_split(const AccumPlan& plan, const AccumArgs& a, hipStream_t st);  // split variant, kernel 2 (via launch_accumulate)
void split_occupancy_blocks(int PW, int WP, int K, size_t smem, int* search, int* accum);
// table may be nullptr (split variant): then only anchor / tablea / cmax are written
hipError_t launch_build_table3(const int32_t* idx, int64_t B, int K, const double* tgt, int64_t M, double* table,
                               double* anchor, float4* tablea, float* cmax, hipStream_t st);
hipError_t launch_build_table2(const int32_t* idx, int64_t B, int K, const double* tgt, int64_t M, double* table,
                               float4* tablef, float* cmax, hipStream_t st);
// q[b] = (s·R^T) + t with the stage-B expression (SVNICP.cpp:62-64); pose12 = device [R row-major | t]
hipError_t launch_transform_cloud(const double* src, int64_t B, const double* pose12, double* q, const int* ctl, hipStream_t st);
// ---------------- particle update (particle_update.hip) ----------------
struct UpdateArgs {
  const double* sums;  // [P][kNSums] (all particles); source-row sharding: [n_ranks][P][kNSums], summed in rank order on load
  int n_ranks;         // 1, or the number of row-shard records behind `sums`
  int sums_stride;     // doubles between two records of `sums` (0: P * kNSums); small chain: the accumulate kernel's partial rows
  double* sums_out;    // small chain: k_upd_prepare stores each particle's reduced record here, or nullptr
  double* R;           // [P][9]
  double* t;           // [P][3]
  double* Rtot;        // [P][12]
  Pose0 pose;
  int P, iteration, iterations;
  double lr, conv_thr;
  int check_early_stop, full_grad;
  double* work;        // workspace, see update_workspace_doubles()
  float* history;      // [I][6][P]
  double* pose_out;    // [6][P]
  int* ctl;            // [0] stop flag, [1] finish_iter
  double *trH, *trb, *trN, *trphi, *trh;  // optional traces (per-iteration slices) or nullptr
  int h_in_lds;        // set by launch_update: per-particle H fits in LDS
  // SVGD-ICP mode (k_particle_update_svgd)
  double* eul;         // [P][6] optimizer parameters x,y,z,roll,pitch,yaw
  double* opt;         // [3][P][6] optimizer state
  int svgd;            // 1: SVGD-ICP mode through the workgroup-parallel chain (k_upd_prepare … k_upd_finish)
  int optimizer;       // SVNICP_OPT_*
  double n_src;        // gradient_scaling_factor_ = B (SVGDICP.cpp:58)
  double* uctl;        // multi-workgroup update path: small control / norm area
  unsigned long long* dbg;  // optional [8]: cycle stamps of the fused kernel's phases (SVNICP_DEBUG)
};
size_t update_workspace_doubles(int P);
// areas k_init_particles clears at the start of a registration (whole 32-bit words), and the control words it resets
struct BeginZero { unsigned int* ptr[6]; unsigned int dwords[6]; int n; int* ctl; int iterations; };
hipError_t launch_init_particles(const double* init6xP, int P, const Pose0& pose, int mode, double* R, double* t,
                                 double* Rtot, double* pose_out, int refresh_pose, double* eul, hipStream_t st, const BeginZero* zero = nullptr);
// stage B accumulate (fused variants: whole stage B); single: see stein_iter.hip — the one-particle iteration in one launch
bool accumulate_can_fuse_single(const AccumPlan& plan);
int single_particle_grid(int64_t B);   // workgroups of the one-particle iteration kernel (rows of `partial` it writes)
hipError_t launch_accumulate(const AccumPlan& plan, AccumArgs a, const UpdateArgs* single, hipStream_t st);
hipError_t launch_update_svgd(const UpdateArgs& a, hipStream_t st);
hipError_t launch_update(const UpdateArgs& a, hipStream_t st);
// the Stein step of P >= 2 particles in three pieces (particle_update.hip): pair statistics (second stream), sums -> H, b,
// Newton steps, direction + pose update
hipError_t launch_update_median(const UpdateArgs& a, int num_cus, int max_p_one_workgroup, hipStream_t st);
hipError_t launch_update_prepare(const UpdateArgs& a, hipStream_t st);
hipError_t launch_update_prepare_median(const UpdateArgs& a, hipStream_t st);   // small chain: both in one launch (2 <= P <= 128)
// small chain, all iterations in one cooperative launch (particle_update.hip: k_small_registration)
bool small_registration_supported(int PW, int WP, int K);
hipError_t launch_small_registration(const AccumPlan& plan, AccumArgs a, const UpdateArgs& u, int iterations, unsigned int* bar,
                                     int num_cus, hipStream_t st);
hipError_t launch_update_direction(const UpdateArgs& a, hipStream_t st, bool finish = true);   // finish: k_upd_finish behind it when the step needs one
const double* update_step_norms(const UpdateArgs& a);   // [P] norms the early-stop decision averages
hipError_t launch_reduce_partials(const double* partial, int nblk, int Ppad, int p_lo, int n_particles, double* sums, const int* ctl,
                                  hipStream_t st);
size_t update_uctl_doubles(int P);
struct StatsArgs { const double* pose; int P; int mode; double* out; /* mean6,var6,cov36,weightsP */ };
hipError_t launch_stats(const StatsArgs& a, hipStream_t st);

}  // namespace svnicp
