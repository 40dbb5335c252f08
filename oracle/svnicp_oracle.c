/*
 * svnicp_oracle.c — plain-C (C11 + optional OpenMP) CPU restatement of the reference's
 * Stein-ICP registration path: svnicp::SVNICP::stein_align / svnicp::SVGDICP::stein_align and
 * everything under them.  See svnicp_oracle.h for the parity status and the usage rules
 * (TEST INFRASTRUCTURE ONLY — never linked into the product).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/svn-icp/).  The arithmetic is written the way the reference's tensor program
 * evaluates it (operation order, masking-by-multiplication, float32 weights/history, lower
 * median, LU solve); it does NOT use the algebraic shortcuts of the HIP product path, so that
 * it can check them.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include "svnicp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0; /* 0 = OpenMP default */

void orc_set_threads(int n) { g_threads = n; }
int orc_get_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}
#ifdef _OPENMP
#define NTHREADS() (g_threads > 0 ? g_threads : omp_get_max_threads())
#else
#define NTHREADS() 1
#endif

/* ------------------------------------------------------------------------------------------
 * KNN (src/core/knn/knn_cpu.cpp:35-67)
 * The reference keeps a std::priority_queue<std::tuple<float,int>> (max-heap on (dist, idx)),
 * inserts when size<K or dist < top.dist (strict), pops the max tuple after an insert at
 * capacity, and finally empties the heap back-to-front => ascending by (dist, idx).
 * Below: the same thing with an explicit binary max-heap ordered lexicographically.
 * ---------------------------------------------------------------------------------------- */
#define DEFINE_KNN(NAME, T)                                                                  \
  typedef struct { T d; int i; } NAME##_ent;                                                 \
  static inline int NAME##_less(NAME##_ent a, NAME##_ent b) {                                \
    return (a.d < b.d) || (a.d == b.d && a.i < b.i);                                         \
  }                                                                                          \
  static void NAME##_sift_up(NAME##_ent *h, int n) {                                         \
    int c = n - 1;                                                                           \
    while (c > 0) {                                                                          \
      int p = (c - 1) / 2;                                                                   \
      if (NAME##_less(h[p], h[c])) { NAME##_ent t = h[p]; h[p] = h[c]; h[c] = t; c = p; }     \
      else break;                                                                            \
    }                                                                                        \
  }                                                                                          \
  static void NAME##_pop(NAME##_ent *h, int n) { /* remove max from heap of size n */        \
    h[0] = h[n - 1];                                                                         \
    n -= 1;                                                                                  \
    int p = 0;                                                                               \
    for (;;) {                                                                               \
      int l = 2 * p + 1, r = l + 1, m = p;                                                   \
      if (l < n && NAME##_less(h[m], h[l])) m = l;                                           \
      if (r < n && NAME##_less(h[m], h[r])) m = r;                                           \
      if (m == p) break;                                                                     \
      NAME##_ent t = h[p]; h[p] = h[m]; h[m] = t; p = m;                                     \
    }                                                                                        \
  }                                                                                          \
  static void NAME##_one(const T *q, const T *tgt, int64_t M, int K, NAME##_ent *heap,       \
                         int64_t *idx, T *dist2) {                                           \
    int size = 0;                                                                            \
    for (int64_t i2 = 0; i2 < M; ++i2) {                                                     \
      T dist = 0;                                                                            \
      for (int d = 0; d < 3; ++d) { /* knn_cpu.cpp:43-50 */                                  \
        T diff = q[d] - tgt[3 * i2 + d];                                                     \
        dist += diff * diff;                                                                 \
      }                                                                                      \
      if (size < K || dist < heap[0].d) { /* knn_cpu.cpp:52 */                               \
        heap[size].d = dist; heap[size].i = (int)i2;                                         \
        NAME##_sift_up(heap, size + 1);                                                      \
        if (size >= K) NAME##_pop(heap, size + 1); else size += 1;                           \
      }                                                                                      \
    }                                                                                        \
    for (int k = 0; k < K; ++k) { idx[k] = 0; dist2[k] = 0; } /* knn_cpu.cpp:25-26 */        \
    while (size > 0) { /* knn_cpu.cpp:59-65 */                                               \
      NAME##_ent t = heap[0];                                                                \
      NAME##_pop(heap, size);                                                                \
      size -= 1;                                                                             \
      dist2[size] = t.d; idx[size] = t.i;                                                    \
    }                                                                                        \
  }

DEFINE_KNN(knn64, double)
DEFINE_KNN(knn32, float)

void orc_knn_topk(const double *q, int64_t B, const double *tgt, int64_t M, int K,
                  int64_t *idx, double *dist2) {
#pragma omp parallel num_threads(NTHREADS())
  {
    knn64_ent *heap = (knn64_ent *)malloc(sizeof(knn64_ent) * (size_t)(K + 1));
#pragma omp for schedule(dynamic, 16)
    for (int64_t b = 0; b < B; ++b)
      knn64_one(q + 3 * b, tgt, M, K, heap, idx + b * K, dist2 + b * K);
    free(heap);
  }
}

void orc_knn_topk_f32(const float *q, int64_t B, const float *tgt, int64_t M, int K,
                      int64_t *idx, float *dist2) {
  knn32_ent *heap = (knn32_ent *)malloc(sizeof(knn32_ent) * (size_t)(K + 1));
  for (int64_t b = 0; b < B; ++b)
    knn32_one(q + 3 * b, tgt, M, K, heap, idx + b * K, dist2 + b * K);
  free(heap);
}

/* SVGDICP.cpp:204  transformed = source.matmul(R0^T) + t0   (row . row of R0, then + t0) */
void orc_transform(const double *src, int64_t B, const double R[9], const double t[3],
                   double *out) {
  for (int64_t b = 0; b < B; ++b) {
    const double *s = src + 3 * b;
    for (int i = 0; i < 3; ++i)
      out[3 * b + i] = (s[0] * R[3 * i + 0] + s[1] * R[3 * i + 1] + s[2] * R[3 * i + 2]) + t[i];
  }
}

/* ------------------------------------------------------------------------------------------
 * small dense helpers
 * ---------------------------------------------------------------------------------------- */
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

/* LAPACK dgetrf-style LU with partial pivoting on a 6x6 (row-major copy) */
static int lu6(double A[36], int piv[6]) {
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double mx = fabs(A[6 * k + k]);
    for (int r = k + 1; r < 6; ++r)
      if (fabs(A[6 * r + k]) > mx) { mx = fabs(A[6 * r + k]); p = r; }
    piv[k] = p;
    if (p != k)
      for (int c = 0; c < 6; ++c) { double t = A[6 * k + c]; A[6 * k + c] = A[6 * p + c]; A[6 * p + c] = t; }
    if (A[6 * k + k] == 0.0) return 1;
    double inv = 1.0 / A[6 * k + k];
    for (int r = k + 1; r < 6; ++r) A[6 * r + k] *= inv;
    for (int r = k + 1; r < 6; ++r)
      for (int c = k + 1; c < 6; ++c) A[6 * r + c] -= A[6 * r + k] * A[6 * k + c];
  }
  return 0;
}
static void lu6_solve(const double LU[36], const int piv[6], double x[6]) {
  for (int k = 0; k < 6; ++k)
    if (piv[k] != k) { double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; }
  for (int r = 1; r < 6; ++r)
    for (int c = 0; c < r; ++c) x[r] -= LU[6 * r + c] * x[c];
  for (int r = 5; r >= 0; --r) {
    for (int c = r + 1; c < 6; ++c) x[r] -= LU[6 * r + c] * x[c];
    x[r] /= LU[6 * r + r];
  }
}
/* torch::linalg::solve(H, b) -> at::linalg_solve -> LAPACK gesv (SVNICP.cpp:162) */
int orc_solve6(const double A[36], const double b[6], double x[6]) {
  double LU[36]; int piv[6];
  memcpy(LU, A, sizeof LU);
  int info = lu6(LU, piv);
  memcpy(x, b, 6 * sizeof(double));
  if (info) { for (int i = 0; i < 6; ++i) x[i] = NAN; return info; }
  lu6_solve(LU, piv, x);
  return 0;
}
/* torch::linalg::inv -> solve(A, I) (SVNICP.cpp:225,250) */
int orc_inv6(const double A[36], double Ainv[36]) {
  double LU[36]; int piv[6];
  memcpy(LU, A, sizeof LU);
  int info = lu6(LU, piv);
  if (info) { for (int i = 0; i < 36; ++i) Ainv[i] = NAN; return info; }
  for (int c = 0; c < 6; ++c) {
    double e[6] = {0, 0, 0, 0, 0, 0};
    e[c] = 1.0;
    lu6_solve(LU, piv, e);
    for (int r = 0; r < 6; ++r) Ainv[6 * r + c] = e[r];
  }
  return 0;
}

/* SVNICP.cpp:166-194  to_rotation_tensor (Rodrigues) with the J_l_ side effect (:188-192) */
void orc_so3_exp(const double r[3], double R[9], double Jl[9]) {
  const double angle = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]); /* :170 */
  double a[3];
  if (angle < 1e-12) { a[0] = a[1] = a[2] = 0.0; }                      /* :171-173 */
  else { a[0] = r[0] / angle; a[1] = r[1] / angle; a[2] = r[2] / angle; }
  const double c = cos(angle), s = sin(angle);
  const double ah[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0}; /* :176-180 */
  const double soa = s / angle;            /* NaN when angle == 0, as in the reference (:188) */
  const double omc_a = (1 - c) / angle;    /* :192 */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double I = (i == j) ? 1.0 : 0.0;
      const double aa = a[i] * a[j];
      R[3 * i + j] = (c * I + (1 - c) * aa) + s * ah[3 * i + j];                 /* :182-186 */
      if (Jl) Jl[3 * i + j] = (soa * I + (1 - soa) * aa) + omc_a * ah[3 * i + j]; /* :188-192 */
    }
}

/* SVNICP.cpp:196-215  rotm_to_ypr_tensor == SO(3) log */
void orc_so3_log(const double R[9], double w[3]) {
  double c = 0.5 * (R[0] + R[4] + R[8] - 1);  /* :199 */
  if (c < -1) c = -1;
  if (c > 1) c = 1;                           /* clip, :198-200 (NaN propagates) */
  const double angle = acos(c);
  const double sa = sin(angle);
  const int nonzero = fabs(sa) > 1e-12;       /* :205 */
  const double f = 0.5 / (nonzero ? sa : 1.0) * angle; /* :207 */
  w[0] = f * (R[7] - R[5]);                   /* R21 - R12 */
  w[1] = f * (R[2] - R[6]);                   /* R02 - R20 */
  w[2] = f * (R[3] - R[1]);                   /* R10 - R01 */
  if (!nonzero) { w[0] = w[1] = w[2] = 0.0; } /* :213 */
}

/* SVGDICP.cpp:226-260 */
void orc_euler_to_R(double roll, double pitch, double yaw, double R[9]) {
  const double A = cos(yaw), Bs = sin(yaw), C = cos(pitch), D = sin(pitch), E = cos(roll), F = sin(roll);
  R[0] = C * A;  R[1] = F * D * A - E * Bs;  R[2] = F * Bs + E * D * A;
  R[3] = C * Bs; R[4] = E * A + F * D * Bs;  R[5] = E * D * Bs - F * A;
  R[6] = -D;     R[7] = F * C;               R[8] = E * C;
}

/* SVGDICP.cpp:335-396  partial derivatives of R w.r.t. roll/pitch/yaw, premultiplied by R0 */
static void euler_partials(const double R0[9], double roll, double pitch, double yaw, double dR[3][9]) {
  const double A = cos(yaw), Bs = sin(yaw), C = cos(pitch), D = sin(pitch), E = cos(roll), F = sin(roll);
  const double DE = D * E, DF = D * F, AC = A * C, AF = A * F, AE = A * E;
  const double ADE = A * DE, ADF = A * DF, BC = Bs * C, BE = Bs * E, BF = Bs * F, BDE = Bs * DE;
  const double pr[9] = {0, ADE + BF, BE - ADF, 0, -AF + BDE, Bs * (-DF) - AE, 0, C * E, C * (-F)};
  const double pp[9] = {A * -D, AC * F, AC * E, Bs * -D, BC * F, BC * E, -C, -DF, -DE};
  const double py[9] = {-BC, -Bs * DF - AE, AF - BDE, AC, -BE + ADF, ADE + BF, 0, 0, 0};
  mat3_mul(R0, pr, dR[0]);
  mat3_mul(R0, pp, dR[1]);
  mat3_mul(R0, py, dR[2]);
}

/* ------------------------------------------------------------------------------------------
 * solver object
 * ---------------------------------------------------------------------------------------- */
struct orc_solver {
  int mode;
  orc_params prm;
  int P, K;
  int64_t B, M;
  double *src, *tgt;
  double R0[9], t0[3];
  double *R;        /* [P][9]  R_  */
  double *t;        /* [P][3]  t_  */
  double *eul;      /* [P][6]  SVGD: x,y,z,rx,ry,rz optimizer parameters */
  double *pose;     /* [6][P]  pose_particles_ */
  float *history;   /* [I][6][P] particle_stack_ */
  int hist_I;
  int64_t *cand_idx;  /* [B][K] */
  double *cand_d2;    /* [B][K] */
  double *cand_xyz;   /* [B][K][3]  target_batch (one copy; the reference makes I identical ones) */
  double *opt_state;  /* SVGD optimizer state: [3][P][6] */
  int finish_iter;   /* finish_iter_: set in the constructor (SVGDICP.cpp:42) and by SVGDICP::stein_align's early stop only (:128) */
  int iters_run;     /* test tap: iterations the last align executed */
  int full_corr;     /* 1: get_correspondence (SVGDICP.cpp:274-298) instead of get_correspondence_fast */
  orc_trace tr;
  int has_trace;
};

static void set_particles(orc_solver *s, const double *init, int P) {
  if (P != s->P) {
    free(s->R); free(s->t); free(s->eul); free(s->pose); free(s->opt_state);
    s->R = (double *)calloc((size_t)P * 9, 8);
    s->t = (double *)calloc((size_t)P * 3, 8);
    s->eul = (double *)calloc((size_t)P * 6, 8);
    s->pose = (double *)calloc((size_t)P * 6, 8);
    s->opt_state = (double *)calloc((size_t)P * 18, 8);
    s->P = P;
  }
  for (int p = 0; p < P; ++p) {
    const double r[3] = {init[3 * P + p], init[4 * P + p], init[5 * P + p]};
    s->t[3 * p + 0] = init[0 * P + p];
    s->t[3 * p + 1] = init[1 * P + p];
    s->t[3 * p + 2] = init[2 * P + p];
    for (int d = 0; d < 6; ++d) s->eul[6 * p + d] = init[d * P + p];
    if (s->mode == ORC_MODE_SVN) orc_so3_exp(r, s->R + 9 * p, NULL);          /* SVNICP.cpp:34 */
    else orc_euler_to_R(r[0], r[1], r[2], s->R + 9 * p);                       /* SVGDICP.cpp:36 */
  }
}

static void refresh_pose_svn(orc_solver *s) { /* SVNICP.cpp:36-37,74-77,103-106,111-112 */
  const int P = s->P;
  for (int p = 0; p < P; ++p) {
    double w[3];
    orc_so3_log(s->R + 9 * p, w);
    for (int d = 0; d < 3; ++d) { s->pose[d * P + p] = s->t[3 * p + d]; s->pose[(3 + d) * P + p] = w[d]; }
  }
}
static void refresh_pose_svgd(orc_solver *s) { /* SVGDICP.cpp:33-35,118-121,136-138 */
  const int P = s->P;
  for (int p = 0; p < P; ++p)
    for (int d = 0; d < 6; ++d) s->pose[d * P + p] = s->eul[6 * p + d]; /* normalize_factor_ == 1 */
}

orc_solver *orc_create(int mode, const orc_params *prm, const double *init_pose6xP, int P) {
  orc_solver *s = (orc_solver *)calloc(1, sizeof *s);
  s->mode = mode;
  s->prm = *prm;
  s->K = prm->knn_count;                  /* SVGDICP.cpp:43 */
  s->finish_iter = prm->iterations;       /* SVGDICP.cpp:42 */
  s->iters_run = 0;
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(s->R0, I3, sizeof I3);           /* SVGDICP.cpp:38-39 */
  s->t0[0] = s->t0[1] = s->t0[2] = 0;
  set_particles(s, init_pose6xP, P);
  if (mode == ORC_MODE_SVN) refresh_pose_svn(s); else refresh_pose_svgd(s);
  return s;
}

void orc_destroy(orc_solver *s) {
  if (!s) return;
  free(s->src); free(s->tgt); free(s->R); free(s->t); free(s->eul); free(s->pose);
  free(s->history); free(s->cand_idx); free(s->cand_d2); free(s->cand_xyz); free(s->opt_state);
  free(s);
}

/* SVGDICP.cpp:46-62.  NB: pose_particles_ is NOT refreshed here (only in stein_align). */
void orc_add_cloud(orc_solver *s, const double *src, int64_t B, const double *tgt, int64_t M,
                   const double *init_pose6xP, int P) {
  free(s->src); free(s->tgt);
  s->src = (double *)malloc((size_t)B * 24);
  s->tgt = (double *)malloc((size_t)M * 24);
  memcpy(s->src, src, (size_t)B * 24);
  memcpy(s->tgt, tgt, (size_t)M * 24);
  s->B = B; s->M = M;
  if (P != s->P) { /* pose_particles_ keeps its old shape in the reference; here P must not change */
    fprintf(stderr, "orc_add_cloud: particle count change %d -> %d (reference statics forbid it)\n", s->P, P);
  }
  set_particles(s, init_pose6xP, P);
}

void orc_set_initial_mean(orc_solver *s, const double R0[9], const double t0[3]) {
  memcpy(s->R0, R0, 9 * sizeof(double));
  memcpy(s->t0, t0, 3 * sizeof(double));
}
void orc_set_k(orc_solver *s, int k) { s->K = k; }
void orc_set_threshold(orc_solver *s, double md) { s->prm.max_dist = md; }
void orc_set_trace(orc_solver *s, const orc_trace *t) {
  if (t) { s->tr = *t; s->has_trace = 1; } else { memset(&s->tr, 0, sizeof s->tr); s->has_trace = 0; }
}

/* SVGDICP.cpp:176-215  mini_batch_pair_generator + knn_source_cloud (use_minibatch is never
 * set => batch == whole source, the same candidates every epoch).  Split in "rows" + "table" so
 * that the test-side sharded backend can compute a row range per rank. */
static void candidate_alloc(orc_solver *s) {
  const int64_t B = s->B;
  const int K = s->K;
  free(s->cand_idx); free(s->cand_d2); free(s->cand_xyz);
  s->cand_idx = (int64_t *)calloc((size_t)B * K, 8);
  s->cand_d2 = (double *)calloc((size_t)B * K, 8);
  s->cand_xyz = (double *)malloc((size_t)B * K * 24);
}
static void candidate_rows(orc_solver *s, int64_t b_lo, int64_t b_hi) {
  const int K = s->K;
  if (b_hi <= b_lo) return;
  double *q = (double *)malloc((size_t)(b_hi - b_lo) * 24);
  orc_transform(s->src + 3 * b_lo, b_hi - b_lo, s->R0, s->t0, q);                           /* :204 */
  orc_knn_topk(q, b_hi - b_lo, s->tgt, s->M, K, s->cand_idx + b_lo * K, s->cand_d2 + b_lo * K); /* :205-214 */
  free(q);
}
static void candidate_table(orc_solver *s) {
  const int64_t n = s->B * (int64_t)s->K;
  for (int64_t e = 0; e < n; ++e) {                                  /* :191-193 index_select */
    const int64_t i = s->cand_idx[e];
    s->cand_xyz[3 * e + 0] = s->tgt[3 * i + 0];
    s->cand_xyz[3 * e + 1] = s->tgt[3 * i + 1];
    s->cand_xyz[3 * e + 2] = s->tgt[3 * i + 2];
  }
}
static void candidate_stage(orc_solver *s) {
  candidate_alloc(s);
  candidate_rows(s, 0, s->B);
  candidate_table(s);
}

/* one (particle, source point): transform (SVNICP.cpp:62-64), nearest-of-K
 * (SVGDICP.cpp:305-313 -> knn_cpu.cpp, K=1), point_filter (SVGDICP.cpp:331-333).
 * Returns mask; outputs masked source / transformed / target rows. */
static inline int correspond(const orc_solver *s, const double Rt[9], const double tt[3], int64_t b,
                             double sm[3], double Tm[3], double qm[3], int *kbest) {
  const double *sp = s->src + 3 * b;
  double Ts[3];
  for (int i = 0; i < 3; ++i)
    Ts[i] = (sp[0] * Rt[3 * i] + sp[1] * Rt[3 * i + 1] + sp[2] * Rt[3 * i + 2]) + tt[i];
  if (s->full_corr) {
    /* get_correspondence (SVGDICP.cpp:274-298): KNearestNeighborIdx(transformed_source, target, K = 1) over the WHOLE target
     * in index order (knn_cpu.cpp: size < K || dist < top, strict), then the same point_filter; kbest = the target index */
    int64_t bi = 0;
    double bdist = 0;
    for (int64_t j = 0; j < s->M; ++j) {
      double dist = 0;
      for (int d = 0; d < 3; ++d) { const double diff = Ts[d] - s->tgt[3 * j + d]; dist += diff * diff; }
      if (j == 0 || dist < bdist) { bdist = dist; bi = j; }
    }
    const int mm = bdist < s->prm.max_dist;
    const double mff = mm ? 1.0 : 0.0;
    for (int d = 0; d < 3; ++d) { sm[d] = mff * sp[d]; Tm[d] = mff * Ts[d]; qm[d] = mff * s->tgt[3 * bi + d]; }
    *kbest = (int)bi;
    return mm;
  }
  const double *c = s->cand_xyz + (size_t)b * s->K * 3;
  int best = 0;
  double bd = 0;
  for (int k = 0; k < s->K; ++k) {
    double dist = 0;
    for (int d = 0; d < 3; ++d) { const double diff = Ts[d] - c[3 * k + d]; dist += diff * diff; }
    if (k == 0 || dist < bd) { bd = dist; best = k; } /* size<K || dist < top (strict) */
  }
  const int m = bd < s->prm.max_dist; /* NB: squared distance vs un-squared max_dist (SVGDICP.cpp:332) */
  const double mf = m ? 1.0 : 0.0;
  for (int d = 0; d < 3; ++d) { sm[d] = mf * sp[d]; Tm[d] = mf * Ts[d]; qm[d] = mf * c[3 * best + d]; }
  *kbest = best;
  return m;
}

#define CHUNK 2048 /* fixed chunking => results independent of the thread count */

/* SVNICP.cpp:116-164 Newton_grad_right for particles [p_lo,p_hi): H [P][36], b [P][6] */
/* damping: add the 1e-6 diagonal of SVNICP.cpp:153 (0 when the caller sums several row-shard records first) */
static void newton_accumulate_range(orc_solver *s, int epoch, int p_lo, int p_hi, double *H, double *bv, int damping) {
  const int P = s->P;
  const int64_t B = s->B;
  const int64_t nchunk = (B + CHUNK - 1) / CHUNK;
  const double md = s->prm.max_dist;
  double *part = (double *)calloc((size_t)P * nchunk * 42, 8);
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(NTHREADS())
  for (int p = p_lo; p < p_hi; ++p)
    for (int64_t ch = 0; ch < nchunk; ++ch) {
      double Rt[9], tt[3], tmp[3];
      mat3_mul(s->R0, s->R + 9 * p, Rt);            /* SVNICP.cpp:58,145 */
      mat3_vec(s->R0, s->t + 3 * p, tmp);           /* :59 */
      for (int i = 0; i < 3; ++i) tt[i] = s->t0[i] + tmp[i];
      double *acc = part + ((size_t)p * nchunk + ch) * 42;
      const int64_t b1 = (ch + 1) * CHUNK < B ? (ch + 1) * CHUNK : B;
      for (int64_t b = ch * CHUNK; b < b1; ++b) {
        double sm[3], Tm[3], qm[3];
        int kb;
        const int m = correspond(s, Rt, tt, b, sm, Tm, qm, &kb);
        if (s->has_trace) {
          if (s->tr.corr) s->tr.corr[((size_t)epoch * P + p) * B + b] = kb;
          if (s->tr.mask) s->tr.mask[((size_t)epoch * P + p) * B + b] = (uint8_t)m;
        }
        double e[3] = {Tm[0] - qm[0], Tm[1] - qm[1], Tm[2] - qm[2]};        /* :119 */
        const double n = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);       /* :120 */
        const double wq = md / (md + 3 * n);
        const double w = wq * wq;                                            /* :122 */
        e[0] *= w; e[1] *= w; e[2] *= w;                                     /* :123 */
        const double sh[9] = {0, -sm[2], sm[1], sm[2], 0, -sm[0], -sm[1], sm[0], 0}; /* :126-142 */
        double Rs[9];
        mat3_mul(Rt, sh, Rs);
        double J[18];                                                        /* :146  [Rc | -Rc s^] */
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) { J[6 * i + j] = Rt[3 * i + j]; J[6 * i + 3 + j] = -Rs[3 * i + j]; }
        for (int k = 0; k < 6; ++k) {
          for (int l = 0; l < 6; ++l)                                        /* :149-152 */
            acc[6 * k + l] += J[k] * (w * J[l]) + J[6 + k] * (w * J[6 + l]) + J[12 + k] * (w * J[12 + l]);
          acc[36 + k] += J[k] * e[0] + J[6 + k] * e[1] + J[12 + k] * e[2];  /* :154-157 */
        }
      }
    }
  for (int p = p_lo; p < p_hi; ++p) {
    double *Hp = H + 36 * p, *bp = bv + 6 * p;
    for (int i = 0; i < 36; ++i) Hp[i] = 0;
    for (int i = 0; i < 6; ++i) bp[i] = 0;
    for (int64_t ch = 0; ch < nchunk; ++ch) {
      const double *acc = part + ((size_t)p * nchunk + ch) * 42;
      for (int i = 0; i < 36; ++i) Hp[i] += acc[i];
      for (int i = 0; i < 6; ++i) bp[i] += acc[36 + i];
    }
    if (damping) for (int i = 0; i < 6; ++i) Hp[7 * i] += 1e-6;              /* :153 */
  }
  free(part);
}

static void newton_accumulate(orc_solver *s, int epoch, double *H, double *bv) {
  newton_accumulate_range(s, epoch, 0, s->P, H, bv, 1);
}

static int cmp_double(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* SVNICP.cpp:254-266 / SVGDICP.cpp:464-474: K [P][P], h; x is [P][6] */
static double rbf_kernel(const double *x, int P, double *Kmat) {
  double *sq = (double *)malloc((size_t)P * P * 8);
  int has_nan = 0;
  for (int i = 0; i < P; ++i)
    for (int j = 0; j < P; ++j) {
      double a = 0;
      for (int d = 0; d < 6; ++d) { const double df = x[6 * i + d] - x[6 * j + d]; a += df * df; }
      sq[(size_t)i * P + j] = a;
      if (a != a) has_nan = 1;
    }
  double med;
  if (has_nan) med = NAN; /* torch::median propagates NaN */
  else {
    double *tmp = (double *)malloc((size_t)P * P * 8);
    memcpy(tmp, sq, (size_t)P * P * 8);
    qsort(tmp, (size_t)P * P, 8, cmp_double);
    med = tmp[((size_t)P * P - 1) / 2]; /* lower median */
    free(tmp);
  }
  const double h = med / log((double)(P + 1)); /* :262 */
  for (size_t i = 0; i < (size_t)P * P; ++i) Kmat[i] = exp(-sq[i] / h); /* :264 */
  free(sq);
  return h;
}

/* SVNICP.cpp:218-227  svgd_grad(x, newton_grad(= -N), H(= mean_p H)) */
static void svn_svgd_grad(const double *x, const double *negN, const double *Hmean, int P,
                          double *phi, double *h_out) {
  double *Km = (double *)malloc((size_t)P * P * 8);
  const double h = rbf_kernel(x, P, Km);
  double Hinv[36];
  orc_inv6(Hmean, Hinv);
  for (int i = 0; i < P; ++i) {
    double g[6] = {0}, kn[6] = {0}, ks = 0;
    for (int j = 0; j < P; ++j) {
      const double k = Km[(size_t)i * P + j];
      for (int d = 0; d < 6; ++d) { g[d] += (x[6 * i + d] - x[6 * j + d]) * k; kn[d] += k * negN[6 * j + d]; }
      ks += k;
    }
    for (int d = 0; d < 6; ++d) g[d] = 2 / h * g[d];                 /* :221-222 */
    for (int r = 0; r < 6; ++r) {
      double hg = 0;
      for (int c = 0; c < 6; ++c) hg += Hinv[6 * r + c] * g[c];
      phi[6 * i + r] = (kn[r] + hg) / ks;                            /* :224-226 */
    }
  }
  *h_out = h;
  free(Km);
}

/* SVNICP.cpp:229-252  svn_full_grad(x, H, b(= -b)) */
static void svn_full_grad(const double *x, const double *H, const double *negb, int P, double lr,
                          double *phi, double *h_out) {
  double *Km = (double *)malloc((size_t)P * P * 8);
  const double h = rbf_kernel(x, P, Km);
  for (int i = 0; i < P; ++i) {
    double Hm[36] = {0}, u[6] = {0};
    for (int j = 0; j < P; ++j) {
      const double k = Km[(size_t)i * P + j];
      double g[6];
      for (int d = 0; d < 6; ++d) g[d] = 2 / h * ((x[6 * i + d] - x[6 * j + d]) * k); /* :233-234 */
      const double k2 = k * k;                                                        /* :238 */
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) Hm[6 * r + c] += k2 * H[36 * j + 6 * r + c] + g[r] * g[c]; /* :236-242 */
        u[r] += k * negb[6 * j + r] + g[r];                                           /* :244 */
      }
    }
    for (int e = 0; e < 36; ++e) Hm[e] /= P;
    for (int r = 0; r < 6; ++r) u[r] /= P;
    double Hinv[36];
    orc_inv6(Hm, Hinv);
    for (int r = 0; r < 6; ++r) {
      double a = 0;
      for (int c = 0; c < 6; ++c) a += Hinv[6 * r + c] * u[c];
      phi[6 * i + r] = lr * a;                                                        /* :250 */
    }
  }
  *h_out = h;
  free(Km);
}

/* SVNICP.cpp:268-279 pose_update */
static void svn_pose_update(orc_solver *s, const double *phi) {
  for (int p = 0; p < s->P; ++p) {
    double dR[9], Jl[9], dt[3], Rn[9], Rdt[3];
    orc_so3_exp(phi + 6 * p + 3, dR, Jl);
    mat3_vec(Jl, phi + 6 * p, dt);            /* :275 */
    mat3_mul(s->R + 9 * p, dR, Rn);           /* :277 */
    memcpy(s->R + 9 * p, Rn, sizeof Rn);
    mat3_vec(Rn, dt, Rdt);                    /* :278  uses the UPDATED R */
    for (int d = 0; d < 3; ++d) s->t[3 * p + d] = Rdt[d] + s->t[3 * p + d];
  }
}

static void alloc_history(orc_solver *s) { /* SVGDICP.cpp:172-174 */
  free(s->history);
  s->hist_I = s->prm.iterations;
  s->history = (float *)calloc((size_t)s->hist_I * 6 * s->P + 1, sizeof(float));
}

/* everything of one epoch after Newton_grad_right's sums: solve, Stein direction, pose update,
 * early stop, history (SVNICP.cpp:71-107).  Returns 1 when the early stop fired. */
static int svn_update(orc_solver *s, int epoch, double *H, double *bv) {
  const int P = s->P;
  double *N = (double *)malloc((size_t)P * 6 * 8), *phi = (double *)malloc((size_t)P * 6 * 8);
  double *x = (double *)malloc((size_t)P * 6 * 8), *neg = (double *)malloc((size_t)P * 6 * 8);
  int stop = 0;
  for (int p = 0; p < P; ++p) orc_solve6(H + 36 * p, bv + 6 * p, N + 6 * p); /* :162 */
  refresh_pose_svn(s);                                                        /* :74-77 */
  for (int p = 0; p < P; ++p) for (int d = 0; d < 6; ++d) x[6 * p + d] = s->pose[d * P + p];
  double h = NAN;
  if (P > 1) {
    if (s->prm.svn_full_grad) {
      for (int i = 0; i < 6 * P; ++i) neg[i] = -bv[i];
      svn_full_grad(x, H, neg, P, s->prm.lr, phi, &h);                        /* :83 */
    } else {
      double Hm[36] = {0};
      for (int p = 0; p < P; ++p) for (int e = 0; e < 36; ++e) Hm[e] += H[36 * p + e];
      for (int e = 0; e < 36; ++e) Hm[e] /= P;                                /* :85 */
      for (int i = 0; i < 6 * P; ++i) neg[i] = -N[i];
      svn_svgd_grad(x, neg, Hm, P, phi, &h);                                  /* :86 */
    }
  } else {
    for (int i = 0; i < 6; ++i) phi[i] = -N[i];                               /* :89 */
  }
  if (s->has_trace) {
    if (s->tr.H) memcpy(s->tr.H + (size_t)epoch * P * 36, H, (size_t)P * 36 * 8);
    if (s->tr.b) memcpy(s->tr.b + (size_t)epoch * P * 6, bv, (size_t)P * 6 * 8);
    if (s->tr.newton) memcpy(s->tr.newton + (size_t)epoch * P * 6, N, (size_t)P * 6 * 8);
    if (s->tr.phi) memcpy(s->tr.phi + (size_t)epoch * P * 6, phi, (size_t)P * 6 * 8);
    if (s->tr.h) s->tr.h[epoch] = h;
  }
  svn_pose_update(s, phi);                                                    /* :92 */
  if (s->prm.check_early_stop) {                                              /* :95-101 */
    double m = 0;
    for (int p = 0; p < P; ++p) {
      double n2 = 0;
      for (int d = 0; d < 6; ++d) n2 += phi[6 * p + d] * phi[6 * p + d];
      m += sqrt(n2);
    }
    m /= P;
    /* torch::lt(f64 0-dim, f32 1-dim) computes in float32 (type promotion) */
    /* SVNICP::stein_align only breaks (SVNICP.cpp:95-101): finish_iter_ keeps its constructor value */
    if ((float)m < (float)s->prm.convergence_threshold) { s->iters_run = epoch + 1; stop = 1; }
  }
  if (!stop) {
    refresh_pose_svn(s);                                                      /* :103-106 */
    for (int i = 0; i < 6 * P; ++i) s->history[(size_t)epoch * 6 * P + i] = (float)s->pose[i]; /* :107 */
    if (s->has_trace && s->tr.pose) memcpy(s->tr.pose + (size_t)epoch * 6 * P, s->pose, (size_t)6 * P * 8);
  }
  free(N); free(phi); free(x); free(neg);
  return stop;
}

static int svn_align(orc_solver *s) { /* SVNICP.cpp:41-114 */
  const int P = s->P, I = s->prm.iterations;
  alloc_history(s);
  candidate_stage(s);
  double *H = (double *)malloc((size_t)P * 36 * 8), *bv = (double *)malloc((size_t)P * 6 * 8);
  s->iters_run = I;
  for (int epoch = 0; epoch < I; ++epoch) {
    newton_accumulate(s, epoch, H, bv);
    if (svn_update(s, epoch, H, bv)) break;
  }
  refresh_pose_svn(s);                                                        /* :111-112 */
  free(H); free(bv);
  return ORC_ALIGN_SUCCESS;
}

/* ---- split-phase form (SVN mode) used only by the CPU test of the sharded driver
 * (tests/oracle_backend.py): same arithmetic as svn_align, cut at the two exchange points. ---- */
void orc_sp_begin(orc_solver *s) {
  alloc_history(s);
  candidate_alloc(s);
  s->iters_run = s->prm.iterations;
}
void orc_sp_candidate_rows(orc_solver *s, int64_t b_lo, int64_t b_hi) { candidate_rows(s, b_lo, b_hi); }
int64_t *orc_sp_candidates(orc_solver *s) { return s->cand_idx; }
void orc_sp_build_table(orc_solver *s) { candidate_table(s); }
/* rec: [P][42] = H (36) | b (6); only rows [p_lo,p_hi) are written */
/* damping = 0: the record is one of several row-shard partials (this solver holds a slice of the source rows); the
 * caller adds the records and then the 1e-6 diagonal (tests/oracle_backend.py) */
void orc_sp_accumulate_rows(orc_solver *s, int epoch, int p_lo, int p_hi, double *rec, int damping) {
  const int P = s->P;
  double *H = (double *)malloc((size_t)P * 36 * 8), *bv = (double *)malloc((size_t)P * 6 * 8);
  newton_accumulate_range(s, epoch, p_lo, p_hi, H, bv, damping);
  for (int p = p_lo; p < p_hi; ++p) {
    memcpy(rec + (size_t)p * 42, H + 36 * p, 36 * 8);
    memcpy(rec + (size_t)p * 42 + 36, bv + 6 * p, 6 * 8);
  }
  free(H); free(bv);
}
void orc_sp_accumulate(orc_solver *s, int epoch, int p_lo, int p_hi, double *rec) {
  const int P = s->P;
  double *H = (double *)malloc((size_t)P * 36 * 8), *bv = (double *)malloc((size_t)P * 6 * 8);
  newton_accumulate_range(s, epoch, p_lo, p_hi, H, bv, 1);
  for (int p = p_lo; p < p_hi; ++p) {
    memcpy(rec + (size_t)p * 42, H + 36 * p, 36 * 8);
    memcpy(rec + (size_t)p * 42 + 36, bv + 6 * p, 6 * 8);
  }
  free(H); free(bv);
}
int orc_sp_update(orc_solver *s, int epoch, const double *rec) {
  const int P = s->P;
  double *H = (double *)malloc((size_t)P * 36 * 8), *bv = (double *)malloc((size_t)P * 6 * 8);
  for (int p = 0; p < P; ++p) {
    memcpy(H + 36 * p, rec + (size_t)p * 42, 36 * 8);
    memcpy(bv + 6 * p, rec + (size_t)p * 42 + 36, 6 * 8);
  }
  const int stop = svn_update(s, epoch, H, bv);
  free(H); free(bv);
  return stop;
}
void orc_sp_finish(orc_solver *s) { refresh_pose_svn(s); }

/* ------------------------------- SVGD mode ------------------------------------------------ */

/* SVGDICP.cpp:398-455 sgd_grad */
static void sgd_grad(orc_solver *s, int epoch, double *g /* [P][6] */) {
  const int P = s->P;
  const int64_t B = s->B;
  const int64_t nchunk = (B + CHUNK - 1) / CHUNK;
  const double md = s->prm.max_dist;
  double *part = (double *)calloc((size_t)P * nchunk * 7, 8);
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(NTHREADS())
  for (int p = 0; p < P; ++p)
    for (int64_t ch = 0; ch < nchunk; ++ch) {
      double Rt[9], tt[3], tmp[3], dR[3][9];
      const double *eu = s->eul + 6 * p;
      mat3_mul(s->R0, s->R + 9 * p, Rt);
      mat3_vec(s->R0, s->t + 3 * p, tmp);
      for (int i = 0; i < 3; ++i) tt[i] = s->t0[i] + tmp[i];
      euler_partials(s->R0, eu[3], eu[4], eu[5], dR);
      double *acc = part + ((size_t)p * nchunk + ch) * 7;
      const int64_t b1 = (ch + 1) * CHUNK < B ? (ch + 1) * CHUNK : B;
      for (int64_t b = ch * CHUNK; b < b1; ++b) {
        double sm[3], Tm[3], qm[3];
        int kb;
        const int m = correspond(s, Rt, tt, b, sm, Tm, qm, &kb);
        if (s->has_trace) {
          if (s->tr.corr) s->tr.corr[((size_t)epoch * P + p) * B + b] = kb;
          if (s->tr.mask) s->tr.mask[((size_t)epoch * P + p) * B + b] = (uint8_t)m;
        }
        if (((Tm[0] + Tm[1]) + Tm[2]) != 0.0) acc[6] += 1.0;                 /* :404 count_nonzero */
        double e[3] = {Tm[0] - qm[0], Tm[1] - qm[1], Tm[2] - qm[2]};         /* :407 */
        const double n = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);       /* :410 */
        const double wq = md / (md + 3 * n);
        const double w = wq * wq;                                            /* :411 */
        e[0] *= w; e[1] *= w; e[2] *= w;
        acc[0] += e[0]; acc[1] += e[1]; acc[2] += e[2];                      /* :414 error.sum(1) */
        for (int a = 0; a < 3; ++a) {                                        /* :418-452 */
          double ds[3];
          mat3_vec(dR[a], sm, ds);
          acc[3 + a] += e[0] * ds[0] + e[1] * ds[1] + e[2] * ds[2];
        }
      }
    }
  for (int p = 0; p < P; ++p) {
    double a[7] = {0};
    for (int64_t ch = 0; ch < nchunk; ++ch)
      for (int i = 0; i < 7; ++i) a[i] += part[((size_t)p * nchunk + ch) * 7 + i];
    const double cnt1 = a[6] + 1.0;
    for (int j = 0; j < 3; ++j) /* error.sum(1).matmul(R0) / (count+1) */
      g[6 * p + j] = (a[0] * s->R0[j] + a[1] * s->R0[3 + j] + a[2] * s->R0[6 + j]) / cnt1;
    for (int j = 0; j < 3; ++j) g[6 * p + 3 + j] = a[3 + j] / cnt1;
    for (int j = 0; j < 6; ++j) g[6 * p + j] *= (double)B;                   /* :454 gradient_scaling_factor_ */
  }
  free(part);
}

/* torch::optim step with param.grad = -stein_grad (SVGDICP.cpp:476-494; options :142-170) */
static void optimizer_step(orc_solver *s, const double *phi, int step /* 1-based */) {
  const int P = s->P;
  const double lr = s->prm.lr;
  double *m1 = s->opt_state, *m2 = s->opt_state + 6 * P, *m3 = s->opt_state + 12 * P;
  for (int i = 0; i < 6 * P; ++i) {
    double g = -phi[i];
    double *p = &s->eul[i];
    switch (s->prm.optimizer) {
      case 0: { /* Adam: betas (0.9,0.999), eps 1e-8, no weight decay, no amsgrad */
        const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
        m1[i] = b1 * m1[i] + (1 - b1) * g;
        m2[i] = b2 * m2[i] + (1 - b2) * g * g;
        const double bc1 = 1 - pow(b1, step), bc2 = 1 - pow(b2, step);
        const double step_size = lr / bc1;
        const double denom = sqrt(m2[i]) / sqrt(bc2) + eps;
        *p -= step_size * (m1[i] / denom);
      } break;
      case 1: { /* RMSprop: alpha .99, eps 1e-8, weight_decay 1e-8, momentum .9, not centered */
        const double alpha = 0.99, eps = 1e-8, wd = 1e-8, mom = 0.9;
        g = g + wd * (*p);
        m1[i] = alpha * m1[i] + (1 - alpha) * g * g;
        const double avg = sqrt(m1[i]) + eps;
        m2[i] = mom * m2[i] + g / avg;
        *p -= lr * m2[i];
      } break;
      case 2: /* SGD */
        *p -= lr * g;
        break;
      case 3: { /* Adagrad: lr_decay 0, eps 1e-10, initial accumulator 0 */
        m3[i] += g * g;
        *p -= lr * (g / (sqrt(m3[i]) + 1e-10));
      } break;
      default: break;
    }
  }
}

static int svgd_align(orc_solver *s) { /* SVGDICP.cpp:66-140 */
  const int P = s->P, I = s->prm.iterations;
  if (s->prm.optimizer < 0 || s->prm.optimizer > 3) return ORC_NO_OPTIMIZER; /* :73-75 */
  memset(s->opt_state, 0, (size_t)P * 18 * 8);                               /* fresh optimizer, :142-170 */
  alloc_history(s);
  candidate_stage(s);
  double *g = (double *)malloc((size_t)P * 6 * 8), *phi = (double *)malloc((size_t)P * 6 * 8);
  double *x = (double *)malloc((size_t)P * 6 * 8), *old = (double *)malloc((size_t)P * 6 * 8);
  double *Km = (double *)malloc((size_t)P * P * 8);
  s->iters_run = I;   /* finish_iter_ is NOT reset here: it keeps the last early stop's value (SVGDICP.cpp:42,128) */
  for (int epoch = 0; epoch < I; ++epoch) {
    for (int p = 0; p < P; ++p) {                                            /* :88-89 */
      const double *eu = s->eul + 6 * p;
      orc_euler_to_R(eu[3], eu[4], eu[5], s->R + 9 * p);
      s->t[3 * p] = eu[0]; s->t[3 * p + 1] = eu[1]; s->t[3 * p + 2] = eu[2];
    }
    sgd_grad(s, epoch, g);                                                   /* :106 */
    double h = NAN;
    if (P > 1) { /* :110, svgd_grad :457-462 ; NB pose_particles_ may be stale at epoch 0 */
      for (int p = 0; p < P; ++p) for (int d = 0; d < 6; ++d) x[6 * p + d] = s->pose[d * P + p];
      h = rbf_kernel(x, P, Km);
      for (int i = 0; i < P; ++i) {
        double gr[6] = {0}, kg[6] = {0};
        for (int j = 0; j < P; ++j) {
          const double k = Km[(size_t)i * P + j];
          for (int d = 0; d < 6; ++d) { gr[d] += (x[6 * i + d] - x[6 * j + d]) * k; kg[d] += k * (-g[6 * j + d]); }
        }
        for (int d = 0; d < 6; ++d) phi[6 * i + d] = (kg[d] + 2 / h * gr[d]) / P;
      }
    } else {
      for (int d = 0; d < 6; ++d) phi[d] = -g[d];                            /* :112 */
    }
    if (s->has_trace) {
      if (s->tr.phi) memcpy(s->tr.phi + (size_t)epoch * P * 6, phi, (size_t)P * 6 * 8);
      if (s->tr.newton) memcpy(s->tr.newton + (size_t)epoch * P * 6, g, (size_t)P * 6 * 8);
      if (s->tr.h) s->tr.h[epoch] = h;
    }
    memcpy(old, s->pose, (size_t)P * 6 * 8);                                 /* :114 */
    optimizer_step(s, phi, epoch + 1);                                       /* :115 */
    refresh_pose_svgd(s);                                                    /* :118-121 */
    if (s->prm.check_early_stop) {                                           /* :123-131 */
      double m = 0;
      for (int p = 0; p < P; ++p) {
        double n2 = 0;
        for (int d = 0; d < 6; ++d) { const double df = s->pose[d * P + p] - old[d * P + p]; n2 += df * df; }
        m += sqrt(n2);
      }
      m /= P;
      if ((float)m < (float)s->prm.convergence_threshold) { s->finish_iter = epoch + 1; s->iters_run = epoch + 1; break; }
    }
    for (int i = 0; i < 6 * P; ++i) s->history[(size_t)epoch * 6 * P + i] = (float)s->pose[i]; /* :133 */
    if (s->has_trace && s->tr.pose) memcpy(s->tr.pose + (size_t)epoch * 6 * P, s->pose, (size_t)6 * P * 8);
  }
  refresh_pose_svgd(s);                                                      /* :136-138 */
  free(g); free(phi); free(x); free(old); free(Km);
  return ORC_ALIGN_SUCCESS;
}

int orc_stein_align(orc_solver *s) {
  return s->mode == ORC_MODE_SVN ? svn_align(s) : svgd_align(s);
}

/* ------------------------------- outputs --------------------------------------------------- */

static double particle_weight(const orc_solver *s) {
  /* SVNICP.cpp:46: torch::ones({P,1}) / P is FLOAT32, promoted to f64 when multiplied */
  return (double)(1.0f / (float)s->P);
}

void orc_get_transformation(orc_solver *s, double out[6]) {
  const int P = s->P;
  if (s->mode == ORC_MODE_SVN) { /* SVNICP.cpp:286-290 */
    const double w = particle_weight(s);
    for (int d = 0; d < 6; ++d) { double a = 0; for (int p = 0; p < P; ++p) a += s->pose[d * P + p] * w; out[d] = a; }
  } else {                       /* SVGDICP.cpp:497-499 */
    for (int d = 0; d < 6; ++d) { double a = 0; for (int p = 0; p < P; ++p) a += s->pose[d * P + p]; out[d] = a / P; }
  }
}

void orc_get_distribution(orc_solver *s, double out[6]) {
  const int P = s->P;
  double mean[6];
  orc_get_transformation(s, mean);
  if (s->mode == ORC_MODE_SVN) { /* SVNICP.cpp:292-297 */
    const double w = particle_weight(s);
    for (int d = 0; d < 6; ++d) {
      double a = 0;
      for (int p = 0; p < P; ++p) { const double df = s->pose[d * P + p] - mean[d]; a += df * df * w; }
      out[d] = a;
    }
  } else {                       /* SVGDICP.cpp:501-503  torch::var (unbiased) */
    for (int d = 0; d < 6; ++d) {
      double a = 0;
      for (int p = 0; p < P; ++p) { const double df = s->pose[d * P + p] - mean[d]; a += df * df; }
      out[d] = a / (P - 1);
    }
  }
}

void orc_get_cov_matrix(orc_solver *s, double out[36]) {
  const int P = s->P;
  double mean[6];
  orc_get_transformation(s, mean);
  const double w = s->mode == ORC_MODE_SVN ? particle_weight(s) : 1.0;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) {
      double a = 0;
      for (int p = 0; p < P; ++p)
        a += w * ((s->pose[r * P + p] - mean[r]) * (s->pose[c * P + p] - mean[c]));
      out[6 * r + c] = s->mode == ORC_MODE_SVN ? a : a / P; /* SVNICP.cpp:299-308 / SVGDICP.cpp:505-513 */
    }
}

void orc_get_particles(orc_solver *s, double *out6P) { memcpy(out6P, s->pose, (size_t)6 * s->P * 8); }

void orc_get_particle_weight(orc_solver *s, double *outP) {
  const double w = s->mode == ORC_MODE_SVN ? particle_weight(s) : 1.0; /* SVNICP.cpp:281-284 / SVGDICP.cpp:522-524 */
  for (int p = 0; p < s->P; ++p) outP[p] = w;
}

void orc_get_particle_history(orc_solver *s, float *out) {
  if (s->history) memcpy(out, s->history, (size_t)s->hist_I * 6 * s->P * sizeof(float));
}
int orc_get_finish_iter(orc_solver *s) { return s->finish_iter; }
int orc_get_iterations_run(orc_solver *s) { return s->iters_run; }
void orc_set_correspondence_full(orc_solver *s, int on) { s->full_corr = on ? 1 : 0; }
const int64_t *orc_get_candidates(orc_solver *s) { return s->cand_idx; }
const double *orc_get_candidate_dist2(orc_solver *s) { return s->cand_d2; }
