/*
 * svnicp_oracle.h — CPU restatement of the reference Stein-ICP registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (svn-icp_amd/, include/) may include,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * Parity status: the KNN stage is PINNED against the reference's own, unmodified
 * svn-icp/src/core/knn/knn_cpu.cpp compiled into oracle/_ref (see oracle/ref_knn_harness.cpp
 * and tests/test_oracle_knn_ref.py).  The solver stages (SVNICP.cpp / SVGDICP.cpp) are
 * "PARITY UNPINNED": the reference ships no tests / golden vectors, and its solver TUs cannot
 * be built here without stand-ins for absent headers (PCL, Eigen, GTSAM, rclcpp), which the
 * build rules forbid.  They are cross-checked instead against an independent op-by-op
 * libtorch(ATen, CPU, f64) restatement in oracle/torch_restatement.py.
 *
 * All file:line citations are relative to /root/reference/svn-icp/.
 */
#ifndef SVNICP_ORACLE_H
#define SVNICP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors svnicp::SteinICPParam (include/core/SVGDICP.h:41-57), solver-relevant fields only */
typedef struct {
  int iterations;               /* config_.iterations                       */
  double lr;                    /* config_.lr                               */
  double max_dist;              /* config_.max_dist                         */
  int check_early_stop;         /* config_.check_early_stop                 */
  double convergence_threshold; /* config_.convergence_threshold            */
  int knn_count;                /* config_.KNN_count -> K_source_           */
  int svn_full_grad;            /* config_.SVN_full_grad                    */
  int optimizer;                /* SVGD mode only: 0 Adam,1 RMSprop,2 SGD,3 Adagrad, -1 none */
} orc_params;

enum { ORC_ALIGN_SUCCESS = 1, ORC_NO_OPTIMIZER = 2 }; /* SVGDICP.h:59-62 */
enum { ORC_MODE_SVN = 0, ORC_MODE_SVGD = 1 };

/* optional per-iteration trace (caller-allocated; any pointer may be NULL) */
typedef struct {
  int32_t *corr;   /* [I][P][B]  position k in [0,K) chosen by get_correspondence_fast          */
  uint8_t *mask;   /* [I][P][B]  1 if dist2 < max_dist (point_filter)                            */
  double *H;       /* [I][P][36]                                                                 */
  double *b;       /* [I][P][6]                                                                  */
  double *newton;  /* [I][P][6]                                                                  */
  double *phi;     /* [I][P][6]  stein_grad                                                      */
  double *h;       /* [I]        RBF bandwidth                                                   */
  double *pose;    /* [I][6][P]  pose_particles_ after the update (f64, before the f32 cast)     */
} orc_trace;

typedef struct orc_solver orc_solver;

/* ---- stage functions (also used stand-alone by the tests) ---- */

/* knn_cpu.cpp:35-67 semantics on double: K smallest by (dist2, idx), strict '<', ascending output;
 * slots >= M keep idx 0 / dist 0 (torch::full(...,0), knn_cpu.cpp:25-26). */
void orc_knn_topk(const double *q, int64_t B, const double *tgt, int64_t M, int K,
                  int64_t *idx, double *dist2);
/* float32 twin of the above, bit-for-bit the reference loop (used to pin against oracle/_ref) */
void orc_knn_topk_f32(const float *q, int64_t B, const float *tgt, int64_t M, int K,
                      int64_t *idx, float *dist2);
/* SVGDICP.cpp:204  src * R0^T + t0 */
void orc_transform(const double *src, int64_t B, const double R[9], const double t[3], double *out);
/* SVNICP.cpp:166-194 : R = Exp(r); Jl = left Jacobian (side effect J_l_) */
void orc_so3_exp(const double r[3], double R[9], double Jl[9]);
/* SVNICP.cpp:196-215 */
void orc_so3_log(const double R[9], double w[3]);
/* SVGDICP.cpp:226-260 : R = Rz(yaw) Ry(pitch) Rx(roll) */
void orc_euler_to_R(double roll, double pitch, double yaw, double R[9]);
/* LU with partial pivoting (LAPACK dgesv as used by at::linalg_solve on CPU); returns 0 on success */
int orc_solve6(const double A[36], const double b[6], double x[6]);
int orc_inv6(const double A[36], double Ainv[36]);

/* ---- solver object mirroring svnicp::SVNICP / svnicp::SVGDICP ---- */
orc_solver *orc_create(int mode, const orc_params *prm, const double *init_pose6xP, int P);
void orc_destroy(orc_solver *s);
/* SVGDICP.cpp:46-62 */
void orc_add_cloud(orc_solver *s, const double *src, int64_t B, const double *tgt, int64_t M,
                   const double *init_pose6xP, int P);
/* SVGDICP.h:102-110 ; R0 row-major = true rotation */
void orc_set_initial_mean(orc_solver *s, const double R0[9], const double t0[3]);
void orc_set_k(orc_solver *s, int k);               /* SVGDICP.h:98  */
void orc_set_threshold(orc_solver *s, double md);   /* SVGDICP.h:100 */
void orc_set_trace(orc_solver *s, const orc_trace *t);
void orc_set_threads(int n);                         /* OpenMP threads used by the heavy loops */
int orc_get_threads(void);
/* SVNICP.cpp:41-114 / SVGDICP.cpp:66-140 */
int orc_stein_align(orc_solver *s);

void orc_get_transformation(orc_solver *s, double out[6]);      /* SVNICP.cpp:286-290 */
void orc_get_distribution(orc_solver *s, double out[6]);        /* SVNICP.cpp:292-297 */
void orc_get_cov_matrix(orc_solver *s, double out[36]);         /* SVNICP.cpp:299-308 */
void orc_get_particles(orc_solver *s, double *out6P);           /* SVGDICP.cpp:515-520 */
void orc_get_particle_weight(orc_solver *s, double *outP);      /* SVNICP.cpp:281-284 */
void orc_get_particle_history(orc_solver *s, float *outIx6P);   /* SVGDICP.cpp:526-534 */
int orc_get_finish_iter(orc_solver *s);      /* finish_iter_ as the reference keeps it (SVGDICP.cpp:42,128) */
int orc_get_iterations_run(orc_solver *s);   /* test tap: iterations the last align executed */
/* 1: correspondences by the reference's get_correspondence (SVGDICP.cpp:274-298, K = 1 over the whole target per
 * particle; dead code upstream, kept as the optional mode of SURVEY.md §8 a20); trace corr then holds target indices */
void orc_set_correspondence_full(orc_solver *s, int on);
/* candidate indices of the last align: [B][K] int64 (sourceKNN_idx_, SVGDICP.cpp:214) */
const int64_t *orc_get_candidates(orc_solver *s);
const double *orc_get_candidate_dist2(orc_solver *s);

/* split-phase form of the SVN path, for the CPU (gloo) test of the sharded driver only */
void orc_sp_begin(orc_solver *s);
void orc_sp_candidate_rows(orc_solver *s, int64_t b_lo, int64_t b_hi);
int64_t *orc_sp_candidates(orc_solver *s);
void orc_sp_build_table(orc_solver *s);
void orc_sp_accumulate(orc_solver *s, int epoch, int p_lo, int p_hi, double *recPx42);
/* the same for a solver that holds a SLICE of the source rows: damping = 0 leaves the 1e-6 diagonal to the caller, who adds
 * the slices' records first (row-sharded multi-GPU layout, tests/oracle_backend.py) */
void orc_sp_accumulate_rows(orc_solver *s, int epoch, int p_lo, int p_hi, double *recPx42, int damping);
int orc_sp_update(orc_solver *s, int epoch, const double *recPx42);
void orc_sp_finish(orc_solver *s);

#ifdef __cplusplus
}
#endif
#endif
