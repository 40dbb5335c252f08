/*
 * svnicp_hip.h — C ABI of libsvnicp_hip.so: MI355X (gfx950) Stein-variational ICP registration.
 *
 * This is the drop-in boundary for the reference's solver classes svnicp::SVGDICP (virtual) /
 * svnicp::SVNICP (final).  The reference has no FFI: its only caller hands libtorch tensors on
 * kCUDA and one gtsam::Pose3 to those classes (svn-icp/src/core/OdometryPipeline.cpp:282-288,
 * 573-607, 1021).  Every entry point below names the reference interface it replaces
 * (file:line relative to /root/reference/svn-icp/).  Plain pointers and sizes only; no torch,
 * no C++ types.  All floating point is float64 (reference: include/core/SVGDICP.h:207), row-major.
 *
 * Threading: like the reference (one steinicp_thread_, OdometryPipeline.cpp:106-110) a context
 * is used by one host thread at a time; calls are sequential per scan.  Ownership: the context
 * owns all device memory; the caller owns every host buffer it passes in or out.
 *
 * Return convention: 0 (or a SteinICPState for svnicp_align) on success, a negative
 * svnicp_status on failure; nothing throws across this ABI.  svnicp_last_error() gives text.
 */
#ifndef SVNICP_HIP_H
#define SVNICP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVNICP_ABI_VERSION 1

typedef struct svnicp_ctx svnicp_ctx;

/* enum SteinICPState — include/core/SVGDICP.h:59-62 */
#define SVNICP_ALIGN_SUCCESS 1
#define SVNICP_NO_OPTIMIZER 2

typedef enum {
  SVNICP_OK = 0,
  SVNICP_ERR_INVALID = -1,   /* bad argument / call order            */
  SVNICP_ERR_HIP = -2,       /* a HIP runtime call failed             */
  SVNICP_ERR_NO_DEVICE = -3, /* no gfx950 device / code object        */
  SVNICP_ERR_NOMEM = -4
} svnicp_status;

/* class_type — OdometryPipeline.cpp:282-288 ("SVNICP" | "SVGDICP") */
#define SVNICP_MODE_SVN 0
#define SVNICP_MODE_SVGD 1

/* optimizer names of SVGDICP::set_optimizer — src/core/SVGDICP.cpp:142-170 */
#define SVNICP_OPT_ADAM 0
#define SVNICP_OPT_RMSPROP 1
#define SVNICP_OPT_SGD 2
#define SVNICP_OPT_ADAGRAD 3
#define SVNICP_OPT_NONE (-1)

#define SVNICP_MEM_HOST 0
#define SVNICP_MEM_DEVICE 1

/* svnicp::SteinICPParam — include/core/SVGDICP.h:41-57 (solver-relevant fields; batch_size,
 * normalize_cloud, convergence_steps, cov_filter_type are ignored by the reference solver) */
typedef struct svnicp_params {
  int32_t struct_size;           /* = sizeof(svnicp_params)                                   */
  int32_t mode;                  /* SVNICP_MODE_*                                              */
  int32_t iterations;            /* SteinICPParam::iterations                                  */
  int32_t knn_count;             /* SteinICPParam::KNN_count  (K_source_).  K <= 128: matrix-pipe
                                  * search and Morton-tile stage A (the tuned path); larger K runs
                                  * the LDS-tile kernels and is refused with SVNICP_ERR_INVALID in
                                  * svnicp_align / svnicp_align_begin once the smallest tile no longer
                                  * fits one CU's 160 KB of LDS (about K >= 330 for shards of <= 8
                                  * particles, K >= 620 otherwise) */
  double lr;                     /* SteinICPParam::lr                                          */
  double max_dist;               /* SteinICPParam::max_dist                                    */
  double convergence_threshold;  /* SteinICPParam::convergence_threshold                       */
  int32_t check_early_stop;      /* SteinICPParam::check_early_stop                            */
  int32_t svn_full_grad;         /* SteinICPParam::SVN_full_grad                               */
  int32_t optimizer;             /* SteinICPParam::optimizer as SVNICP_OPT_*                   */
  int32_t record_trace;          /* test hook: keep per-iteration H,b,N,phi,h,correspondences  */
} svnicp_params;

/* ctor svnicp::SVNICP(param, init_pose[6,P,1], opt) / svnicp::SVGDICP(param, init_pose)
 * — src/core/SVNICP.cpp:20-38, src/core/SVGDICP.cpp:22-44.  init_pose6xP may be NULL (then
 * svnicp_set_particles must be called before svnicp_align).  `device` is the HIP ordinal. */
int svnicp_create(const svnicp_params *params, int device, const double *init_pose6xP, int P,
                  svnicp_ctx **out);
void svnicp_destroy(svnicp_ctx *ctx);
const char *svnicp_last_error(const svnicp_ctx *ctx); /* ctx may be NULL: last create() error */
int svnicp_abi_version(void);

/* run on this hipStream_t instead of the context's private stream: lets a host that owns streams
 * (torch, a ROS executor) keep kernels, copies and collectives in ONE queue.  NULL is a valid
 * handle — HIP's default (null) stream, which is what torch.cuda.current_stream() usually is;
 * pass SVNICP_OWN_STREAM to return to the private stream. */
#define SVNICP_OWN_STREAM ((void *)(intptr_t)-1)
int svnicp_set_stream(svnicp_ctx *ctx, void *hip_stream);
int svnicp_synchronize(svnicp_ctx *ctx);

/* SVGDICP::add_cloud(source[B,3], target[M,3], init_pose[6,P,1]) — src/core/SVGDICP.cpp:46-62.
 * Split in two because the clouds and the particles are independent buffers; both are copied.  SVNICP_MEM_HOST: the
 * caller's buffer is free when the call returns.  SVNICP_MEM_DEVICE: the copy is QUEUED on the context's stream — the
 * device buffer must stay unchanged until svnicp_align has returned (or svnicp_synchronize).  svnicp_set_particles stages
 * the poses through pinned memory and does not wait for the stream either. */
int svnicp_set_clouds(svnicp_ctx *ctx, const double *src_xyz, int64_t B, const double *tgt_xyz,
                      int64_t M, int mem_kind);
int svnicp_set_particles(svnicp_ctx *ctx, const double *init_pose6xP, int P);
/* the two halves of svnicp_set_clouds, for callers whose clouds live on different sides (host source scan, device
 * target from svnicp_map_query): each copies its cloud; both must have been given before svnicp_align */
int svnicp_set_source(svnicp_ctx *ctx, const double *src_xyz, int64_t B, int mem_kind);
int svnicp_set_target(svnicp_ctx *ctx, const double *tgt_xyz, int64_t M, int mem_kind);

/* SVGDICP::set_initial_mean(gtsam::Pose3) — include/core/SVGDICP.h:102-110.
 * R0 is the rotation matrix row-major (the reference's R0_ after its transpose), t0 the translation. */
int svnicp_set_initial_mean(svnicp_ctx *ctx, const double R0_rowmajor[9], const double t0[3]);
/* SVGDICP::set_k / set_threshold — include/core/SVGDICP.h:98,100 */
int svnicp_set_k(svnicp_ctx *ctx, int k);
int svnicp_set_max_dist(svnicp_ctx *ctx, double max_dist);

/* SVNICP::stein_align() / SVGDICP::stein_align() — src/core/SVNICP.cpp:41-114,
 * src/core/SVGDICP.cpp:66-140.  Synchronous.  Returns SVNICP_ALIGN_SUCCESS / SVNICP_NO_OPTIMIZER
 * or a negative svnicp_status. */
int svnicp_align(svnicp_ctx *ctx);
/* same work, enqueued only (no host sync): for benchmarking / pipelining; results are valid after
 * svnicp_synchronize().  With check_early_stop every iteration is enqueued (those behind the stop return at once on the
 * device); the blocking svnicp_align instead follows the device's stop flag and stops enqueuing a few iterations after it
 * (the reference breaks out of its loop, SVNICP.cpp:95-101) — same result, less idle launching when the run stops early. */
int svnicp_align_async(svnicp_ctx *ctx);

/* results — caller-allocated outputs, copied device -> host */
int svnicp_get_transformation(svnicp_ctx *ctx, double out6[6]);      /* SVNICP.cpp:286-290 / SVGDICP.cpp:497-499 */
int svnicp_get_distribution(svnicp_ctx *ctx, double out6[6]);        /* SVNICP.cpp:292-297 / SVGDICP.cpp:501-503 */
int svnicp_get_cov_matrix(svnicp_ctx *ctx, double out36[36]);        /* SVNICP.cpp:299-308 / SVGDICP.cpp:505-513 */
int svnicp_get_particles(svnicp_ctx *ctx, double *out6P);            /* SVGDICP.cpp:515-520: x..,y..,z..,rx..,ry..,rz.. */
int svnicp_get_particle_weight(svnicp_ctx *ctx, double *outP);       /* SVNICP.cpp:281-284 / SVGDICP.cpp:522-524 */
int svnicp_get_particle_history(svnicp_ctx *ctx, float *outIx6P);    /* SVGDICP.cpp:526-534: I rows of 6P float32 */
int svnicp_get_runtime(svnicp_ctx *ctx, double out3[3]);             /* SVGDICP.h:94-96 {knn_s, update_s, finish_iter} */

/* test / profiling knobs of a context, by name (the product configuration is the default of every one):
 *   knn = auto|v1|v2|brute|tiles   brute_qb = 0..6   fallback_sliced_max = <n>      accum = split|valu|f64
 *   update = auto|fused     fused_update_max_p = <P>       wgpcu = <search>,<accumulate>     tp = <points>    debug = 0|1
 *   single = fused|split    chain = auto|general|persistent   median = auto|stream|inline          correspondence = fast|full
 * The environment variable SVNICP_OPTIONS ("name=value;name=value") is read once, in svnicp_create. */
int svnicp_set_option(svnicp_ctx *ctx, const char *name, const char *value);

/* ---- split-phase entry points: one process per GPU, particles sharded across ranks ----------
 * (new functionality; the reference is single-GPU).  Sequence per registration:
 *   svnicp_set_shard -> svnicp_stage_candidates(b_lo,b_hi) -> [host all-gathers rows of
 *   svnicp_candidates_devptr] -> svnicp_build_candidate_table -> per iteration:
 *   svnicp_iter_accumulate -> [host all-gathers svnicp_sums_devptr, 22 doubles per particle] ->
 *   svnicp_iter_update ; finally svnicp_finish.  svnicp_align == all of it with one shard. */
int svnicp_set_shard(svnicp_ctx *ctx, int p_lo, int p_hi);
/* ---- the other split: SOURCE ROWS sharded across ranks (the one bench.py --gpus N uses) ---------------------------
 * Rank r is given only ITS rows of the source scan (svnicp_set_source / svnicp_set_clouds with the row slice) and the
 * whole target; it runs stage A, the candidate table and the per-iteration search + accumulation on those rows for ALL
 * particles, so nothing of size [B] is replicated or gathered.  Its 22 sums per particle are a partial record; the ranks
 * exchange them (all-gather of row_world x P x 22 doubles into svnicp_rank_sums_devptr, slot = row_rank) and
 * svnicp_iter_update adds the records in rank order before the Stein step — every rank the same values in the same
 * order, so the replicas stay bit-identical.  total_source_points = the whole scan's B (SVGD mode scales by it,
 * SVGDICP.cpp:58).  row_world = 1 returns to the unsharded behaviour.  May be combined with svnicp_set_shard
 * (2-D split): the record slot [row_rank][p_lo, p_hi) is then this rank's contribution.
 * Sequence: svnicp_set_row_shard -> svnicp_align_begin -> svnicp_stage_candidates(0, rows) ->
 * svnicp_build_candidate_table -> per iteration { svnicp_iter_accumulate -> [all-gather] -> svnicp_iter_update } ->
 * svnicp_finish. */
int svnicp_set_row_shard(svnicp_ctx *ctx, int row_rank, int row_world, int64_t total_source_points);
void *svnicp_rank_sums_devptr(svnicp_ctx *ctx);  /* double [row_world][P][SVNICP_NSUMS]; NULL unless row_world > 1 */
int svnicp_align_begin(svnicp_ctx *ctx);
int svnicp_stage_candidates(svnicp_ctx *ctx, int64_t b_lo, int64_t b_hi);
int svnicp_build_candidate_table(svnicp_ctx *ctx);
int svnicp_iter_accumulate(svnicp_ctx *ctx, int iteration);
int svnicp_iter_update(svnicp_ctx *ctx, int iteration);
int svnicp_finish(svnicp_ctx *ctx);
int svnicp_stopped(svnicp_ctx *ctx);     /* 1 once the early-stop flag is set (syncs the stream) */
void *svnicp_candidates_devptr(svnicp_ctx *ctx); /* int32 [B][K] */
void *svnicp_sums_devptr(svnicp_ctx *ctx);       /* double [P][SVNICP_NSUMS] */
#define SVNICP_NSUMS 22

/* ---- local map in HBM: svnicp::VoxelHashMap — src/core/VoxelHashMap.cpp:22-101, include/core/VoxelHashMap.h ----------
 * voxel -> at most max_points points (float32, insertion order); the query result is float64 rows in device memory that
 * svnicp_set_target(..., SVNICP_MEM_DEVICE) copies device-to-device, so the target never crosses PCIe. */
typedef struct svnicp_map svnicp_map;
/* VoxelHashMap(voxel_size, max_range, max_pointscount) — VoxelHashMap.h:39-42; capacity_voxels 0 = default (2^20, grows) */
int svnicp_map_create(int device, double voxel_size, double max_range, int max_points, int64_t capacity_voxels,
                      svnicp_map **out);
void svnicp_map_destroy(svnicp_map *map);
const char *svnicp_map_last_error(const svnicp_map *map);
int svnicp_map_clear(svnicp_map *map);                        /* VoxelHashMap::Clear — VoxelHashMap.h:55 */
int svnicp_map_size(svnicp_map *map, int64_t *voxels);        /* VoxelHashMap::Size / Empty — VoxelHashMap.h:56-57 */
/* VoxelHashMap::AddPointCloud(cloud, pose) incl. RemoveFarPointCloud — VoxelHashMap.cpp:22-42, 89-97.
 * xyz: n x 3 float32 (pcl::PointXYZ) in the sensor frame, host or device; pose = rotation (row-major) + translation */
int svnicp_map_add_cloud(svnicp_map *map, const float *xyz, int64_t n, int mem_kind, const double R_rowmajor[9],
                         const double t[3]);
/* points svnicp_map_add_cloud has not stored since creation / clear because they lie outside +-2^20 voxels or are NaN
 * (the reference would index a voxel for them; here they are counted and the call still succeeds) */
int svnicp_map_skipped_points(svnicp_map *map, int64_t *out);
/* VoxelHashMap::GetMap(pose, max_range) — VoxelHashMap.cpp:48-58; center NULL or max_range < 0: GetMap() (:44-46).
 * The points are written as float64 [count][3] rows into a device buffer owned by the map (valid until the next query),
 * voxels in ascending (x, y, z) index, points of a voxel in insertion order. */
int svnicp_map_query(svnicp_map *map, const double center[3], double max_range, int64_t *count_out);
void *svnicp_map_points_devptr(svnicp_map *map);              /* double [count][3] of the last query */
int svnicp_map_download(svnicp_map *map, double *out_xyz, int64_t cap_points, int64_t *n_out); /* test tap */

/* ---- test-only taps (parity tests; not part of the reference interface) -------------------- */
int svnicp_get_candidates(svnicp_ctx *ctx, int32_t *outBK);          /* sourceKNN_idx_  SVGDICP.cpp:214 */
int svnicp_get_candidate_dist2(svnicp_ctx *ctx, double *outBK);
/* valid when params.record_trace != 0; any pointer may be NULL.
 * corr: [I][P][B] int32 (-1 where not run), H [I][P][36], b [I][P][6], newton [I][P][6],
 * phi [I][P][6], h [I] */
int svnicp_get_trace(svnicp_ctx *ctx, int32_t *corr, double *H, double *b, double *newton,
                     double *phi, double *h);
/* iterations the last align executed (= iterations unless the early stop fired); svnicp_get_runtime()[2] is the
 * reference's finish_iter_, which SVN mode never updates */
int svnicp_get_iterations_run(svnicp_ctx *ctx, int *out);
/* elapsed GPU milliseconds of the last align, by phase: {stage A (candidates + table),
 * iterations (accumulate + update), total} — measured with hipEvents on the context's stream */
int svnicp_get_gpu_ms(svnicp_ctx *ctx, double out3[3]);

/* number of queries of the last stage A that the pre-filtered kernel handed to the streaming
 * fallback (-1 when the streaming kernel ran alone) */
int svnicp_get_knn_fallbacks(svnicp_ctx *ctx, int *out);
/* the source rows behind that count (at most `cap` of them, unordered); *n_out = the count */
int svnicp_get_knn_fallback_rows(svnicp_ctx *ctx, int32_t *out, int cap, int *n_out);
/* per source point: how many targets survived the float32 pre-filter of the pruned stage-A kernel
 * (needs params.record_trace) */
int svnicp_get_knn_survivors(svnicp_ctx *ctx, int32_t *outB);
/* wave steps of the last align whose float32 nearest-candidate search was not decisive and were
 * redone in float64 (-1 when the float64 kernel ran alone) */
int svnicp_get_ambiguous_steps(svnicp_ctx *ctx, int *out);
/* (source point, particle) pairs of the last align that the bf16 matrix-pipe search could not certify and handed to its
 * exact float64 pass (-1 when another search kernel ran) */
int svnicp_get_ambiguous_pairs(svnicp_ctx *ctx, int64_t *out);
/* bench hook: when on, every kernel launch of an align is bracketed by hipEvents on the
 * context's stream; svnicp_get_kernel_ms then returns the summed milliseconds and launch counts
 * per kernel class of the LAST align, SVNICP_KERNEL_CLASSES entries in this order:
 *   0 stage A (ordering + k_knn_tiles/k_knn_scan + fallback)   1 k_build_table*
 *   2 k_stein_search_bf16 (split stage B only)                 3 k_stein_accumulate* (fused variants: whole stage B)
 *   4 k_reduce_partials                                        5 k_particle_update / k_upd_* */
#define SVNICP_KERNEL_CLASSES 6
/* on: 0 = off, 1 = every class, otherwise a mask with bit (class + 1) set for each class to bracket (the event
 * pairs cost ~5 us of stream time each, so a timed run brackets only what it reports) */
int svnicp_set_profile(svnicp_ctx *ctx, int on);
int svnicp_get_kernel_ms(svnicp_ctx *ctx, double *ms6, int32_t *launches6);

/* ---- per-scan pre-processing on the device ------------------------------------------------
 * What OdometryPipeline::ICP_processing does to a scan before the solver sees it (src/core/OdometryPipeline.cpp):
 * crop_pointcloud (:692-704), pcl::UniformSampling at 0.5 * voxel_size (:559, :684-690) and at 1.5 * voxel_size of that
 * cloud (:560).  Mind the reference's aliasing: downsample_uniform filters the cloud it is handed IN PLACE
 * (uniform_sampling_.filter(*cloud) on its own input, :684-690), so after :559 *cropped_cloud holds the 0.5-voxel sampling
 * and after :560 *voxelized_cloud_toMap holds the 1.5-voxel one: the map is SEEDED with the 0.5-voxel cloud (:585,
 * svnicp_prep_map_cloud_devptr) and UPDATED with the 1.5-voxel cloud (:630, svnicp_prep_source_f32_devptr) — the same
 * points the solver registers.  The raw float32 scan is uploaded once; the three clouds stay in
 * device memory for svnicp_map_add_cloud(..., SVNICP_MEM_DEVICE) and svnicp_set_source(..., SVNICP_MEM_DEVICE).
 * Leaves are emitted in ascending linear index, the point closest to a leaf centre survives, first in input order on
 * ties (svn-icp_amd/host/registration_pipeline.hpp: downsample_uniform).  scan_max_range: in/out, the largest SQUARED
 * norm seen so far (:699, kept as the reference keeps it).  The counts are valid until the next svnicp_prep_scan. */
typedef struct svnicp_prep svnicp_prep;
int svnicp_prep_create(int device, svnicp_prep **out);
void svnicp_prep_destroy(svnicp_prep *prep);
const char *svnicp_prep_last_error(const svnicp_prep *prep);
int svnicp_prep_scan(svnicp_prep *prep, const float *xyz, int64_t n, int mem_kind, double min_range, double max_range,
                     double voxel_size, double *scan_max_range, int64_t *n_cropped, int64_t *n_map, int64_t *n_source);
const float *svnicp_prep_cropped_devptr(svnicp_prep *prep);    /* float32 [n_cropped][3] */
const float *svnicp_prep_map_cloud_devptr(svnicp_prep *prep);  /* float32 [n_map][3]     */
const double *svnicp_prep_source_devptr(svnicp_prep *prep);    /* float64 [n_source][3]: the solver's source cloud */
const float *svnicp_prep_source_f32_devptr(svnicp_prep *prep); /* the same points as float32 rows: what the map is UPDATED with */
int svnicp_prep_download(svnicp_prep *prep, int which /* 0 cropped, 1 map cloud, 2 source */, float *out_xyz,
                         int64_t cap_points, int64_t *n_out);   /* test tap */

#ifdef __cplusplus
}
#endif
#endif /* SVNICP_HIP_H */
