// stein_split_device.hpp — device bodies of stage B's two kernels (k_stein_search_bf16, k_stein_accumulate_w) and their
// helpers.  Included by stein_split.hip (the kernels and their launchers) and by particle_update.hip (the persistent
// small-registration kernel runs the same bodies on virtual blocks).  See stein_split.hip for the overview.
#pragma once
#include <cstdlib>
#include "kernels.hpp"
#include "stein_common.hpp"

namespace svnicp {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
#define SVNICP_CONST_AS __attribute__((address_space(4)))   // read-only for the launch: wave-uniform addresses become s_load

__device__ __forceinline__ float pack_slot(float v, unsigned int mask, unsigned int bits) {
  return __uint_as_float((__float_as_uint(v) & ~mask) | bits);
}
// positive floats (and +inf) order like their bit patterns: integer min/max need no NaN canonicalisation
__device__ __forceinline__ float imin_f(float a, float b) {
  const int x = (int)__float_as_uint(a), y = (int)__float_as_uint(b);
  return __uint_as_float((unsigned int)(x < y ? x : y));
}
__device__ __forceinline__ float imax_f(float a, float b) {
  const int x = (int)__float_as_uint(a), y = (int)__float_as_uint(b);
  return __uint_as_float((unsigned int)(x > y ? x : y));
}

// ---------------------------------------------------------------------------------------------
// search on the bf16 matrix pipe
// ---------------------------------------------------------------------------------------------
// Measured on gfx950 (tests/microbench/search_loop.hip, valu_ops.hip): v_mfma_f32_16x16x4_f32 holds the SIMD's
// vector issue for all of its 32 cycles (tile = 32 + tracking, nothing overlaps), while v_mfma_f32_16x16x32_bf16
// takes 16 cycles and holds the vector issue for 8 — the tracking VALU work of other waves runs beside it.  So the
// float32 operands are split EXACTLY into three bf16 pieces each (a = a1 + a2 + a3, round-to-nearest-even pieces
// from v_cvt_pk_bf16_f32) and the six products a_i·b_j with i + j <= 4 of every component go into the K slots:
//   lane group g < 3 (component g):  A = [c1 c1 c1 c2 c2 c3 0 0]   B = [m1 m2 m3 m1 m2 m1 0 0]
//   lane group 3    (|c'|² row)   :  A = split of cc               B = split of 1.0 = [1 0 0 1 0 1 0 0]
// The dropped products (a2·b3, a3·b2, a3·b3) are below 2·2^-24·|c_d·m_d| per component.
// The per-particle term |x'|² is the same for every candidate and is left out: the scores S = cc + c'·m are compared as
// (signed) floats, the accumulator input is the constant 0.
// Error bound (u = 2^-24) — every term is a worst case, nothing in it is measured.  Notation: a = the point's first
// candidate (origin of the local frame), y_k = q_k − a and x = T_p(s) − a in exact arithmetic, s_k = |y_k|² − 2 y_k·x =
// |T − q_k|² − |x|² the exact score; c_k = fl32(y_k), cc_k = fl32(|c_k|²), m = −2·fl32(x) the float32 inputs;
// C2 = max_k |c_k|₂ (cmax[], rounded up), X2 = |fl32(x)|₂.
//   (i)   inputs: each coordinate is rounded twice (f64 subtraction, f32 conversion), relative 1.0001u; cc once more:
//         |cc_k − |y_k|²| <= 3.01u·C2², |c_k·m − (−2 y_k·x)| <= 4.01u·C2·X2
//   (ii)  dropped products a2·b3, a3·b2, a3·b3 (|a2| <= 2^-8·1.004|a|, |a3| <= 2^-16|a|): <= 2.02u·Σ_d|c_d m_d| <= 4.04u·C2·X2
//   (iii) the matrix pipe's sum of the 21 non-zero products (each exact in float32: 8-bit x 8-bit significands; the 11
//         zero products and the zero accumulator input add nothing).  Σ|products| <= 1.008·cc_k + 1.016·Σ_d|c_d m_d| <=
//         1.009·C2² + 2.032·C2·X2.  The hardware's summation is not documented, so the budget is the larger of the two
//         worst cases that exist for a float32 adder tree: (a) ANY order of 20 two-operand additions, each faithfully
//         rounded or truncated (relative error <= 2u per addition): (1+2u)^20 − 1 <= 40.01u; (b) a fused adder that
//         aligns all products to the largest exponent, truncates each to >= 24 bits and rounds once: 20·2u + 2u = 42u.
//         Budget 48u·Σ|products| <= 48.5u·C2² + 97.6u·C2·X2.  (What gfx950 does, from the probes of
//         tests/microbench/mfma_bf16x3_err.hip: model (b) with 25 bits kept and round-to-nearest-even at the end, worst
//         case 21u; largest error seen on adversarial operands 5.7u.)
//   (iv)  the reference's own float64 evaluation of d²_k: <= 5·2^-53·|T − q_k|² <= 2^-50·(C2 + X2)²
//   sum <= 51.5u·C2² + 105.7u·C2·X2 + 2^-50(C2+X2)²  <=  EPS := 54·u·Cq·(Cq + 2·X2),  Cq = C2 + u·X2
//   (the u·X2 in Cq keeps (iv) covered when the candidates are closer together than 2^-24 of their distance to the point).
// Round 2 used 64u(C∞+X∞)² with (iii) budgeted from a measurement; written in the ∞-norms this bound would be
// 160u(C∞+X∞)² — the 2-norms are what keeps the proven bound as tight as the measured one was.
// tests/test_gpu_parity.py::test_mfma_bf16x3_error_budget runs the microbenchmark (the kernel's operand construction on
// random, cancelling and extreme-ratio inputs; arbitrary bf16 operands in the 21 live slots: half-ulp ties, terms just
// below one ulp, graded magnitudes with alternating signs, cancelling pairs, 40 binades of exponents, every rotation over
// the slots) and fails when any result is off by more than HALF of (iii)'s budget — a matrix pipe that behaved worse than
// every model above would be noticed, not trusted.
// How EPS is used — the decision between 4-candidate tiles and inside the winning tile, conditions (a) and (b) — is
// written above search_body.  Ties, padded duplicates, NaN/Inf never pass and go to the exact f64 pass.  Tracking: tags
// are v_bitop3_b32 (full-rate VALU class; v_and_or_b32, v_min*, v_med3_* issue at 0.6x).
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int cvt_pk_bf16(float lo, float hi) {  // v_cvt_pk_bf16_f32: RNE, lo in bits 15:0
  const f2v v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ bf8 split_a3(float x) {   // [a1 a1 | a1 a2 | a2 a3 | a2 a3], a1 + a2 + a3 == x exactly; slots 6,7 meet zeros
  const unsigned int r0 = cvt_pk_bf16(x, x);
  const float e1 = x - __uint_as_float(r0 & 0xffff0000u);
  const unsigned int r1 = cvt_pk_bf16(x, e1);
  const float e2 = e1 - __uint_as_float(r1 & 0xffff0000u);
  const unsigned int r2 = cvt_pk_bf16(e1, e2);
  const u4v t = {r0, r1, r2, r2};
  return __builtin_bit_cast(bf8, t);
}
__device__ __forceinline__ bf8 split_b3(float x) {   // [b1 b2 | b3 b1 | b2 b1 | 0 0]
  const unsigned int r0 = cvt_pk_bf16(x, x);
  const float e1 = x - __uint_as_float(r0 & 0xffff0000u);
  const unsigned int q0 = cvt_pk_bf16(x, e1);
  const float e2 = e1 - __uint_as_float(q0 & 0xffff0000u);
  const unsigned int q1 = cvt_pk_bf16(e2, x);
  const unsigned int q2 = cvt_pk_bf16(e1, x);
  const u4v t = {q0, q1, q2, 0u};
  return __builtin_bit_cast(bf8, t);
}
__device__ __forceinline__ float pack_slot3(float v, unsigned int bits) {  // (v & ~31) | bits in one full-rate op
  return __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(v), 31u, bits, 0xBA));
}
// min / max / median of scores of either sign.  Compiler-visible operations, not inline assembly: the result registers of a
// matrix instruction need software wait states before a vector instruction may read them, and the compiler's hazard
// recognizer does not look inside an asm statement (a v_min3_f32 written in asm read stale registers: 44 % of the pairs came
// out "undecided").  IEEE-2019 minimum / maximum (v_minimum3_f32 / v_maximum3_f32 on gfx950) need no canonicalising copies
// of their inputs and propagate a NaN — which only ever comes with a NaN error bound and sends the pair to the exact pass.
__device__ __forceinline__ float fmin_raw(float a, float b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ float fmax_raw(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float fmed3_raw(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ float fmin3_raw(float a, float b, float c) { return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c); }
constexpr int kQueueCap = 1024;  // undecided pairs a workgroup can defer to its exact pass (4 KB of LDS)

// exact float64 nearest-of-K of one (source point, particle) pair, candidate-parallel over G lanes (lane id `sub` in
// 0..G-1, all G lanes call with the same pair): the reference's arithmetic and tie rule (knn_cpu.cpp:43-50 with K = 1,
// SVGDICP.cpp:300-329) — strict '<' from candidate 0, so a NaN first distance is never replaced
template <int G>
__device__ __forceinline__ int exact_nearest_of_k(const AccumArgs& a, const double (*pose)[64], int64_t b, int pl, int K, int sub) {
  constexpr int NIT = 128 / G;   // K <= 128 in the matrix-pipe kernels: every lane's candidates are requested back to back
  const double* sp = a.src + 3 * b;
  const int32_t* ci = a.cand + (size_t)b * K;
  int64_t ti[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int k = sub + i * G;
    int64_t t = ci[k < K ? k : 0];
    ti[i] = t < 0 ? 0 : (t >= a.M ? a.M - 1 : t);
  }
  double rx[NIT], ry[NIT], rz[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const double* r = a.tgt + 3 * ti[i];
    rx[i] = r[0]; ry[i] = r[1]; rz[i] = r[2];
  }
  const double u0 = sp[0], u1 = sp[1], u2 = sp[2];
  const double t0 = (u0 * pose[0][pl] + u1 * pose[1][pl] + u2 * pose[2][pl]) + pose[9][pl];   // SVNICP.cpp:62-64, as in the step
  const double t1 = (u0 * pose[3][pl] + u1 * pose[4][pl] + u2 * pose[5][pl]) + pose[10][pl];
  const double t2 = (u0 * pose[6][pl] + u1 * pose[7][pl] + u2 * pose[8][pl]) + pose[11][pl];
  double bd = __builtin_huge_val(), d_first = 0.0;
  int bk = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int k = sub + i * G;
    const double dx = t0 - rx[i], dy = t1 - ry[i], dz = t2 - rz[i];
    const double d = (dx * dx + dy * dy) + dz * dz;   // knn_cpu.cpp:43-50 order, unfused
    if (k == 0) d_first = d;
    if (k < K && (d < bd || (d == bd && k < bk))) { bd = d; bk = k; }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {
    const double od = __shfl_xor(bd, off, kWave);
    const int ok = __shfl_xor(bk, off, kWave);
    if (od < bd || (od == bd && ok < bk)) { bd = od; bk = ok; }
  }
  const double d0 = __shfl(d_first, (threadIdx.x & (kWave - 1)) & ~(G - 1), kWave);   // lane sub == 0 of this group
  // the serial reference loop starts from candidate 0 and only replaces on '<': a NaN first
  // distance is never replaced, and an all-NaN row keeps index 0
  return (d0 != d0 || bk == 0x7fffffff) ? 0 : bk;
}

#ifndef SVNICP_SEARCH_EPS_U
#define SVNICP_SEARCH_EPS_U 54   // derivation: header of this section (another value is only ever built to MEASURE what the bound costs)
#endif
constexpr float kEpsBf16 = (float)SVNICP_SEARCH_EPS_U * 5.9604644775390625e-08f;

// Work split: all four waves of a workgroup walk source points (one point per wave step when PW = 64); a wave handles
// ALL WP groups of PW particles of its points one after the other, so a point's table rows are fetched and split into
// bf16 pieces once for the whole workgroup's particles.
//
// Tracking by TILES (round 3, second half).  A result register quad of one MFMA holds four CONSECUTIVE candidates
// (16·rb + 4·mk + 0..3) of one particle: a tile.  Keeping the smallest and second smallest of all 24 scores of a lane cost
// 32 slow-class + 24 tag instructions per column block — measured (timing-only builds): the second minimum alone was 31 % of
// the kernel, the per-score tags 11 %.  Now a lane keeps the minimum of each tile (v_min3 + v_min), tags the six TILE minima
// (3 bits row block, 2 bits lane group) and tracks the smallest and second smallest tile minimum: 19 slow + 6 tags per column
// block.  That decides between tiles; inside the winning tile the lane that owns the particle scores the four candidates
// itself, one step later, from an array-of-rows copy of the table (tablef: 64 contiguous bytes per lane, requested at the end
// of the step, consumed after the next step's transform), in float32 FMAs on the same inputs:
//   (a) inside the tile: packed VALU scores v1 < v2 (2 tag bits, 3 ulp) with v2 − v1 > 2·EPS + 2^-21(|v1| + |v2|): each
//       VALU score is within EPS of the exact score (inputs (i), three roundings <= 3.01u·Σ|products|, (iv)), so the tile's
//       other three candidates are strictly farther in exact arithmetic than the VALU argmin t*;
//   (b) other tiles: with b1 < b2 the smallest and second smallest TAGGED tile minima (5 tag bits: < 2^-18 relative),
//       b2 − b1 > 2·EPS + 2^-18(|b1| + |b2|): every candidate j outside the winning tile has exact score
//       s_j >= b2 − 2^-18|b2| − EPS, and the winning tile's matrix-pipe argmin j* has s_j* <= b1 + 2^-18|b1| + EPS; by (a)
//       s_t* <= s_j*, so s_t* < s_j.
// (a) and (b) and t* < K: t* is the float64 argmin with no tie; anything else goes to the exact pass as before.
#ifndef SVNICP_SEARCH_WAVES
#define SVNICP_SEARCH_WAVES 4
#endif
// (bx, by): the workgroup's place in the launch geometry — blockIdx for k_stein_search_bf16, a virtual block for the
// persistent small-registration kernel (particle_update.hip)
template <int PW, int WP, int NRB, bool TAIL>
__device__ __forceinline__ void search_body(const AccumArgs& a, int bx, int by) {
  if (a.ctl[0]) return;
  if (a.fin_iteration >= 0 && bx == 0 && by == 0) {
    // The previous iteration's early-stop decision (SVNICP.cpp:95-101 / SVGDICP.cpp:123-131) and, if the run goes on, its
    // history row (SVNICP.cpp:103-107): k_upd_finish's work, done here by one workgroup instead of a launch of its own.  The
    // other workgroups of THIS launch may already be searching (a search only writes scratch); every later launch sees the
    // flag.  Fixed tree: every replica decides alike.
    __shared__ double sh_fin[NT];
    __shared__ int sh_fin_stop;
    const int ftid = threadIdx.x, FP = a.fin_P;
    double fs = 0.0;
    for (int p = ftid; p < FP; p += NT) fs += a.fin_norms[p];
    sh_fin[ftid] = fs;
    __syncthreads();
    for (int off = NT / 2; off > 0; off >>= 1) {
      if (ftid < off) sh_fin[ftid] += sh_fin[ftid + off];
      __syncthreads();
    }
    if (ftid == 0) {
      const double m = sh_fin[0] / FP;
      const int stop = (float)m < (float)a.fin_thr;
      if (stop) { a.fin_ctl[0] = 1; a.fin_ctl[1] = a.fin_iteration + 1; }
      sh_fin_stop = stop;
    }
    __syncthreads();
    if (sh_fin_stop) return;
    for (int e = ftid; e < 6 * FP; e += NT) a.fin_history[(size_t)a.fin_iteration * 6 * FP + e] = (float)a.fin_pose[e];
  }
  constexpr int BW = kWave / PW;   // source points per wave step (1, 2, 4)
  constexpr int CBP = PW / 16;     // 16-particle column blocks per source point (4, 2, 1)
  constexpr int NPT = 4 / CBP;     // distinct source points per wave step
  constexpr int ST = 4 * BW;       // source points between two steps of a wave (four waves along the points)
  constexpr bool PIPE = NPT == 1;  // one point per step: its table rows are fetched a step ahead
  constexpr int NTILE = 4 * NRB + (TAIL ? 1 : 0);   // tiles of four consecutive candidates per source point
  __shared__ float4 s_rows[4][NPT][4][NTILE];   // per wave and point: row k = (c'x, c'y, c'z, |c'|²) at [k & 3][k >> 2]
  __shared__ float4 s_scr4[4][64];       // per wave: (−2x', 1) of each particle lane of the group in flight
  __shared__ double s_pose[WP][12][64];  // the lanes' total poses, re-read every step: 24 VGPRs less than keeping them
  __shared__ unsigned int s_queue[kQueueCap];   // undecided pairs: (point − blk_lo) << 8 | particle lane of the workgroup
  __shared__ unsigned int s_qn, s_qsteps;
  const int tid = threadIdx.x;
  if (tid == 0) { s_qn = 0u; s_qsteps = 0u; }
#pragma unroll
  for (int i = 0; i < kQueueCap / NT; ++i) s_queue[tid + i * NT] = 0xffffffffu;   // a step that straddles the end leaves holes
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = lane % PW, bs = lane / PW;
  const int mj = lane & 15, mk = lane >> 4;
  const int pbase = by * (WP * PW);   // first particle lane of this workgroup
  for (int g = wave; g < WP; g += 4) {        // wave g fills group g's poses (WP <= 4)
    const int p = a.p_lo + pbase + g * PW + pl;
    const double* rp = a.Rtot + 12 * (size_t)(p < a.p_hi ? p : a.p_lo);
#pragma unroll
    for (int i = 0; i < 12; ++i) s_pose[g][i][lane] = rp[i];
  }
  const int K = a.K;
  const SVNICP_CONST_AS v4f* ctab = (const SVNICP_CONST_AS v4f*)a.tablea;
  // LDS copy of a point's rows, written from the A-operand registers: lane (mj, mk) holds component mk of candidates
  // 16·rb + mj — one 4-byte store per row block
  float* const rows_w = reinterpret_cast<float*>(&s_rows[wave][0][mj & 3][mj >> 2]) + mk;
  const int64_t blk_lo = (int64_t)bx * a.spts_per_block;
  const int64_t blk_hi = (blk_lo + a.spts_per_block < a.B) ? blk_lo + a.spts_per_block : a.B;
  __syncthreads();
  const int64_t nfirst = blk_lo + wave * BW;

  int pend_idx = 0;       // winner's target index of the previous step, stored one step late (see the end of the step)
  size_t pend_off = 0;
  bool pend_have = false;
  v4f alo_n, ahi_n;   // PIPE: table rows of the NEXT step, in flight while this step's tiles run
  // … and the next point's source row, local origin and C2 — as VECTOR loads (lanes 0-2, 3-5: one double each; every lane:
  // C2).  As scalar loads at the top of the step they were waited for at once, and the tail rows' scalar loads of every
  // group step held up the next LDS wait (scalar loads and LDS share one counter): three exposed L2 round trips per point.
  double sa_n = 0.0;
  float c_n = 0.0f;
  auto fetch_point = [&](int64_t nn) {
    const double* base = lane < 3 ? a.src + 3 * nn : a.anchor + 3 * nn - 3;
    sa_n = base[lane < 6 ? lane : 3];
    c_n = a.cmax[nn];
  };
  // several points per step (PW < 64): the rows of ALL the next step's points, requested at the top of this step (requested
  // where they are used, each point's two row loads were an exposed round trip: C2's search ran at 2.4x C3's time per pair)
  // (one buffer per point: a point's registers are free once its rows are split, and are refilled at once)
  v4f ralo_n[PIPE ? 1 : NPT], rahi_n[PIPE ? 1 : NPT];
  auto fetch_rows = [&](int64_t n0, int pt) {
    int64_t bq = n0 + pt;
    bq = bq < blk_hi ? bq : (n0 < blk_hi ? n0 : blk_lo);
    const SVNICP_CONST_AS v4f* rowp = ctab + (size_t)bq * 128 + lane;
    ralo_n[pt] = rowp[0];
    rahi_n[pt] = ralo_n[pt];
    if constexpr (NRB > 4) rahi_n[pt] = rowp[64];
  };
  if constexpr (!PIPE) {
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) fetch_rows(nfirst, pt);
  }
  if constexpr (PIPE) {
    fetch_point(nfirst < blk_hi ? nfirst : blk_lo);
    const SVNICP_CONST_AS v4f* rowp = ctab + (size_t)(nfirst < blk_hi ? nfirst : blk_lo) * 128 + lane;
    alo_n = rowp[0];
    ahi_n = alo_n;
    if constexpr (NRB > 4) ahi_n = rowp[64];
  }

  for (int64_t n = nfirst; n < blk_hi; n += ST) {  // wave-uniform
    const int64_t b = n + bs;
    const bool inb = b < blk_hi;
    double s0, s1, s2, a0, a1, a2;   // source row; first candidate = origin of the local frame
    float C;
    if constexpr (PIPE) {
      s0 = rdlane_f64(sa_n, 0); s1 = rdlane_f64(sa_n, 1); s2 = rdlane_f64(sa_n, 2);
      a0 = rdlane_f64(sa_n, 3); a1 = rdlane_f64(sa_n, 4); a2 = rdlane_f64(sa_n, 5);
      C = __uint_as_float((unsigned int)__builtin_amdgcn_readfirstlane((int)__float_as_uint(c_n)));
    } else {   // several points per step: each lane's own point
      const int64_t bl = inb ? b : n;
      const double* sp = a.src + 3 * bl;
      const double* an = a.anchor + 3 * bl;
      s0 = sp[0]; s1 = sp[1]; s2 = sp[2];
      a0 = an[0]; a1 = an[1]; a2 = an[2];
      C = a.cmax[bl];
    }

    // the A operands of a point: its table rows split into bf16 pieces.  One point per step (PW = 64): split once here and
    // used by every particle group; several points per step (PW < 64, one particle group): split when the column blocks reach
    // the point
    bf8 afr[NRB];
    auto split_rows = [&](v4f alo, v4f ahi, int pt) {   // pt: which of the step's NPT points
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const float v = rb == 0 ? alo.x : rb == 1 ? alo.y : rb == 2 ? alo.z : rb == 3 ? alo.w
                      : rb == 4 ? ahi.x : rb == 5 ? ahi.y : rb == 6 ? ahi.z : ahi.w;
        afr[rb] = split_a3(v);
        rows_w[(pt * 4 * NTILE + 4 * rb) * 4] = v;   // row 16·rb + mj lives at [(mj & 3)][4·rb + (mj >> 2)]
      }
      if constexpr (TAIL) {   // candidates 16·NRB … +3 sit in row block NRB of the table (lanes mj < 4): the tile the owner lanes score
        const float v = NRB == 4 ? ahi.x : NRB == 5 ? ahi.y : NRB == 6 ? ahi.z : ahi.w;
        if (mj < 4) rows_w[(pt * 4 * NTILE + 4 * NRB) * 4] = v;
      }
    };
    if constexpr (PIPE) {
      const v4f alo = alo_n, ahi = ahi_n;
      int64_t nn = n + ST;
      nn = nn < blk_hi ? nn : n;
      const SVNICP_CONST_AS v4f* rowp = ctab + (size_t)nn * 128 + lane;
      alo_n = rowp[0];
      if constexpr (NRB > 4) ahi_n = rowp[64];
      fetch_point(nn);
      __builtin_amdgcn_wave_barrier();   // the previous step's tile reads are done
      split_rows(alo, ahi, 0);
    }

#pragma nounroll
    for (int g = 0; g < WP; ++g) {   // not unrolled: the groups would only compete for registers
      float E, mm0, mm1, mm2;
      {
        const double (*pose)[64] = s_pose[g];
        const double T0 = (s0 * pose[0][lane] + s1 * pose[1][lane] + s2 * pose[2][lane]) + pose[9][lane];   // SVNICP.cpp:62-64
        const double T1 = (s0 * pose[3][lane] + s1 * pose[4][lane] + s2 * pose[5][lane]) + pose[10][lane];
        const double T2 = (s0 * pose[6][lane] + s1 * pose[7][lane] + s2 * pose[8][lane]) + pose[11][lane];
        const float xf0 = (float)(T0 - a0), xf1 = (float)(T1 - a1), xf2 = (float)(T2 - a2);
        // EPS = 54u·Cq·(Cq + 2·X2); v_sqrt_f32 is good to 1 ulp, the factor covers it and the roundings of this line
        const float X2 = __builtin_amdgcn_sqrtf(__builtin_fmaf(xf0, xf0, __builtin_fmaf(xf1, xf1, xf2 * xf2))) * 1.000002f;
        const float Cq = __builtin_fmaf(5.9604644775390625e-08f, X2, C);   // NaN (a sentinel row, a non-finite point) stays NaN
        E = kEpsBf16 * Cq * __builtin_fmaf(2.0f, X2, Cq);
        mm0 = -2.0f * xf0; mm1 = -2.0f * xf1; mm2 = -2.0f * xf2;
        if (g > 0) __builtin_amdgcn_wave_barrier();      // the previous group's readers are done with the scratch
        s_scr4[wave][lane] = make_float4(mm0, mm1, mm2, 1.0f);
        __builtin_amdgcn_wave_barrier();
      }
      const float* scr4f = reinterpret_cast<const float*>(s_scr4[wave]);
      float b1[4], b2[4];
      float braw[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) braw[cb] = scr4f[(16 * cb + mj) * 4 + mk];

#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        if constexpr (!PIPE) {
          if (cb % CBP == 0) {
            split_rows(ralo_n[cb / CBP], rahi_n[cb / CBP], cb / CBP);
            fetch_rows(n + ST, cb / CBP);
          }
        }
        const bf8 bfr = split_b3(braw[cb]);
        const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
        // smallest and second smallest of the lane's NRB tagged TILE minima.  Three values give a (smallest, second) pair
        // in two instructions (v_min3, v_med3); two such pairs are folded into the running pair (the second smallest of
        // three pairs is min(med3 of the three smallest, the three seconds)) — these are the slow-issue instruction class.
        float m1 = 0.0f, m2 = 0.0f, wp = 0.0f, rp = 0.0f, pend0 = 0.0f, pend1 = 0.0f;
        int np = 0;            // tile minima waiting for a triple      (all three: compile-time after unrolling)
        bool have_m = false;   // (m1, m2) hold a pair
        bool have_p = false;   // (wp, rp) hold a pair waiting for its partner
        v4f dcur = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr, zero, 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {  // tile rb+1 goes to the matrix pipe before the VALU consumes tile rb
          v4f dnext = dcur;
          if (rb + 1 < NRB) dnext = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[rb + 1], bfr, zero, 0, 0, 0);
          const float pk = pack_slot3(fmin_raw(fmin3_raw(dcur[0], dcur[1], dcur[2]), dcur[3]), (unsigned int)rb);
          if (np == 0) { pend0 = pk; np = 1; }
          else if (np == 1) { pend1 = pk; np = 2; }
          else {
            np = 0;
            const float w = fmin3_raw(pend0, pend1, pk), r = fmed3_raw(pend0, pend1, pk);
            if (!have_p) { wp = w; rp = r; have_p = true; }
            else {
              have_p = false;
              if (!have_m) {      // first two triples: a plain merge of two pairs
                m1 = fmin_raw(wp, w);
                m2 = fmin3_raw(fmax_raw(wp, w), rp, r);
                have_m = true;
              } else {
                const float md = fmed3_raw(m1, wp, w);
                m2 = fmin_raw(fmin3_raw(m2, rp, r), md);
                m1 = fmin3_raw(m1, wp, w);
              }
            }
          }
          dcur = dnext;
        }
        // what is left over when NRB is not a multiple of six
        if (!have_m) { m1 = __builtin_huge_valf(); m2 = __builtin_huge_valf(); }
        if (have_p) { const float t = fmax_raw(m1, wp); m1 = fmin_raw(m1, wp); m2 = fmin3_raw(t, m2, rp); }
        if (np >= 1) { m2 = fmed3_raw(m1, m2, pend0); m1 = fmin_raw(m1, pend0); }
        if (np >= 2) { m2 = fmed3_raw(m1, m2, pend1); m1 = fmin_raw(m1, pend1); }
        b1[cb] = m1; b2[cb] = m2;
      }
      // the four lanes that share a particle (lane groups mk = 0..3): a 4 x 4 transpose-reduce over the lane groups — after
      // v_permlane16_swap on the registers of column blocks (0,1) and (2,3) a lane holds, for the column block of its parity,
      // its own entry and its row partner's; after v_permlane32_swap on those two results lane group mk holds both halves of
      // column block mk.  Six swaps, no copies, no selects; the result lands in the lane that owns the particle.
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) b1[cb] = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(b1[cb]), 0x18u, (unsigned int)mk << 3, 0xBA));
      auto merge2 = [&](float p1, float q1, float p2, float q2, float& o1, float& o2) {
        o1 = fmin_raw(p1, q1);
        o2 = fmin_raw(fmax_raw(p1, q1), fmin_raw(p2, q2));
      };
      float b1own, b2own;
      {
        float h1[2], h2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(b1[2 * j]), __float_as_uint(b1[2 * j + 1]), false, false);
          const auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(b2[2 * j]), __float_as_uint(b2[2 * j + 1]), false, false);
          merge2(__uint_as_float(r1[0]), __uint_as_float(r1[1]), __uint_as_float(r2[0]), __uint_as_float(r2[1]), h1[j], h2[j]);
        }
        const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(h1[0]), __float_as_uint(h1[1]), false, false);
        const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(h2[0]), __float_as_uint(h2[1]), false, false);
        merge2(__uint_as_float(r1[0]), __uint_as_float(r1[1]), __uint_as_float(r2[0]), __uint_as_float(r2[1]), b1own, b2own);
      }
      if constexpr (TAIL) {   // candidates 16·NRB … +3: one more tile (row block NRB of lane group 0), scored by the owner lane
        const float4* tl = &s_rows[wave][bs][0][4 * NRB];   // rows 16·NRB + t at [t][4·NRB]: the same address in every lane (broadcast)
        float sc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 c = tl[t * NTILE];  // (c'x, c'y, c'z, |c'|²), a finite sentinel past K
          sc[t] = __builtin_fmaf(c.x, mm0, __builtin_fmaf(c.y, mm1, __builtin_fmaf(c.z, mm2, c.w)));
        }
        const float pk = pack_slot(fmin_raw(fmin3_raw(sc[0], sc[1], sc[2]), sc[3]), 0x1fu, (unsigned int)NRB);
        b2own = fmed3_raw(b1own, b2own, pk);
        b1own = fmin_raw(b1own, pk);
      }

      const int pin = g * PW + pl;                        // particle lane inside the workgroup
      const bool valid = inb && (a.p_lo + pbase + pin) < a.p_hi;
      const unsigned int wbits = __float_as_uint(b1own);
      int tile = (int)(((wbits & 7u) << 2) | ((wbits >> 3) & 3u));    // 4·rb + mk: candidates 4·tile … 4·tile + 3
      tile = tile < NTILE ? tile : NTILE - 1;                         // (a NaN's tag bits are anything: stay inside the rows)
      const float thr = 2.0f * E + 3.83e-06f * (__builtin_fabsf(b1own) + __builtin_fabsf(b2own)) + 1.0e-30f;   // 2^-18 and a little
      const bool tiles_ok = b2own - b1own > thr;
      // inside the winning tile: the four candidates scored by this lane from the LDS rows, tagged with two bits
      float pk4[4];
      {
        const float4* rt = &s_rows[wave][bs][0][tile];
        if constexpr (!PIPE) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 r = rt[t * NTILE];
          const float sc = __builtin_fmaf(r.x, mm0, __builtin_fmaf(r.y, mm1, __builtin_fmaf(r.z, mm2, r.w)));
          pk4[t] = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(sc), 3u, (unsigned int)t, 0xBA));
        }
      }
      const float w3 = fmin3_raw(pk4[0], pk4[1], pk4[2]), r3 = fmed3_raw(pk4[0], pk4[1], pk4[2]);
      const float v1 = fmin_raw(w3, pk4[3]), v2 = fmed3_raw(w3, r3, pk4[3]);
      const float thrv = 2.0f * E + 4.76837158203125e-07f * (__builtin_fabsf(v1) + __builtin_fabsf(v2)) + 1.0e-30f;   // 2^-21
      int kb = 4 * tile + (int)(__float_as_uint(v1) & 3u);
      const bool ambiguous = valid && (!tiles_ok || !(v2 - v1 > thrv) || kb >= K);
      kb = kb < K ? kb : 0;
      unsigned long long am = __ballot(ambiguous);
      if (am) {  // rare (about one wave step in ten, a lane or two each): queue the undecided pairs for the exact pass below
        const int cnt = __builtin_popcountll(am);
        int base = 0;
        if (lane == 0) { base = (int)atomicAdd(&s_qn, (unsigned int)cnt); atomicAdd(&s_qsteps, 1u); }
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + cnt <= kQueueCap) {
          if (ambiguous) {
            const int rank = __builtin_popcountll(am & ((1ull << lane) - 1ull));
            s_queue[base + rank] = ((unsigned int)(b - blk_lo) << 8) | (unsigned int)pin;
          }
        } else {  // queue full (degenerate clouds: every pair tied): settle this step's pairs here, one lane at a time
          do {
            const int L = (int)__builtin_ctzll(am);
            am &= am - 1;
            const int ke = exact_nearest_of_k<kWave>(a, s_pose[g], n + L / PW, L % PW, K, lane);
            if (lane == L) kb = ke;
          } while (am);
        }
      }
      // the winner's slot byte, and its target index for the accumulate kernel (one dependent load less over there).  The
      // index is a scattered load: it is STORED one step later, so that its latency hides behind the next step's tiles
      // (an undecided pair's two entries are rewritten by the exact pass)
      if (pend_have) a.kidx[pend_off] = pend_idx;
      pend_have = inb;
      if (inb) {
        pend_off = (size_t)b * a.Ppad + (pbase + pin);
        a.kbest[pend_off] = (uint8_t)kb;
        pend_idx = a.cand[(size_t)b * K + kb];
      }
    }
    __builtin_amdgcn_wave_barrier();  // scratch is rewritten by the next step
  }

  if (pend_have) a.kidx[pend_off] = pend_idx;
  // exact pass over the queued pairs: 32 lanes per pair, all waves of the workgroup, no lane waits for another pair
  __syncthreads();
  {
    const unsigned int total = s_qn < (unsigned int)kQueueCap ? s_qn : (unsigned int)kQueueCap;
    // steps that did not fit (base + cnt > kQueueCap) were settled in the loop and left their slots at the sentinel
    constexpr int kExactLanes = 32;
    const int sub = lane & (kExactLanes - 1);
    for (unsigned int e = (unsigned int)(tid / kExactLanes); e < total; e += NT / kExactLanes) {
      const unsigned int ent = s_queue[e];
      if (ent == 0xffffffffu) continue;
      const int64_t be = blk_lo + (int64_t)(ent >> 8);
      const int pin = (int)(ent & 0xffu);
      const int ke = exact_nearest_of_k<kExactLanes>(a, s_pose[pin / PW], be, pin % PW, K, sub);
      if (sub == 0) {
        a.kbest[(size_t)be * a.Ppad + (pbase + pin)] = (uint8_t)ke;
        a.kidx[(size_t)be * a.Ppad + (pbase + pin)] = a.cand[(size_t)be * K + ke];
      }
    }
    if (tid == 0 && a.ambig_count && s_qsteps) { atomicAdd(a.ambig_count, (int)s_qsteps); atomicAdd(a.ambig_count + 1, (int)s_qn); }
  }
}

// (16-particle groups hold four points' rows in LDS, 256-particle workgroups four groups' poses: three workgroups per CU)
template <int PW, int WP, int NRB, bool TAIL>
__global__ __launch_bounds__(NT, (PW == 16 || WP == 4) ? 3 : SVNICP_SEARCH_WAVES) void k_stein_search_bf16(AccumArgs a) {
  search_body<PW, WP, NRB, TAIL>(a, xcd_block((int)blockIdx.x, (int)gridDim.x), (int)blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// accumulation from the winner bytes
// ---------------------------------------------------------------------------------------------
// PLAIN: no correspondence trace and no full-correspondence indices (the timed configurations): no per-pair branches at
// all; SVGD: the first-order mode's count in place of one sum (compile-time with PLAIN, run-time flag otherwise)
template <int PW, int WP, bool PLAIN, bool SVGD = false>
__device__ __forceinline__ void accumulate_body(const AccumArgs& a, int bx, int by, double* lds) {   // lds: the launch's dynamic LDS
  if (a.ctl[0]) return;
  constexpr int BW = kWave / PW;
  constexpr int WB = 4 / WP;
#ifndef SVNICP_ACCUM_U
#define SVNICP_ACCUM_U 4
#endif
  constexpr int U = SVNICP_ACCUM_U;   // points per wave and loop trip: their dependent loads go out as batches (2, 6, 8 measured: no better)
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WP, wb = wave / WP;
  const int pl = lane % PW, bs = lane / PW;
  const int pidx = by * (WP * PW) + wp * PW + pl;
  const int p = a.p_lo + pidx;
  const bool pvalid = p < a.p_hi;

  double Rt[9], tt[3];
  {
    const double* rp = a.Rtot + 12 * (size_t)(pvalid ? p : a.p_lo);
#pragma unroll
    for (int i = 0; i < 9; ++i) Rt[i] = rp[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tt[i] = rp[9 + i];
  }
  double acc[kNSums];
#pragma unroll
  for (int i = 0; i < kNSums; ++i) acc[i] = 0.0;
  const int K = a.K;
  const int64_t blk_lo = (int64_t)bx * a.pts_per_block;
  const int64_t blk_hi = (blk_lo + a.pts_per_block < a.B) ? blk_lo + a.pts_per_block : a.B;
  constexpr int STEP = WB * BW;
  const SVNICP_CONST_AS double* csrc = (const SVNICP_CONST_AS double*)a.src;       // wave-uniform rows become s_load
  const SVNICP_CONST_AS int32_t* ccand = (const SVNICP_CONST_AS int32_t*)a.cand;
  const SVNICP_CONST_AS double* ctgt = (const SVNICP_CONST_AS double*)a.tgt;
  const uint8_t* kbp = a.kbest + pidx;
  const int32_t* kip = a.kidx + pidx;   // PLAIN: the winner's target index, written by the search kernel

  // per trip: winner bytes of the NEXT trip's U points | target indices | winner coordinates | sums — each batch of loads
  // is requested back to back, so a trip pays the byte -> index -> coordinates chain once for U points, and nothing is
  // carried between trips except the U prefetched bytes (no rotating copies).  Rows are clamped, never predicated: a
  // point past the block only changes `on`.
  const int32_t* fullp = (!PLAIN && a.full_idx) ? a.full_idx + (size_t)(pvalid ? p : a.p_lo) * a.B : nullptr;
  auto load_kb = [&](int64_t n) -> int {   // PLAIN: the winner's target index; else the winner byte (correspondence = full: the target index itself)
    const int64_t b = n + bs;
    const int64_t bc = b < blk_hi ? b : blk_lo;
    if (PLAIN) return kip[(size_t)bc * a.Ppad];
    if (fullp) return fullp[bc];
    return (int)kbp[(size_t)bc * a.Ppad];
  };
  // The reference zeroes a rejected row by multiplying with the mask (SVGDICP.cpp:331-333): e = 0, |e| = 0, so w = 1 and
  // J = [R | 0] — the same products are formed here (mf = 0 or 1), which also makes the pair branch-free.
  // `on`: 1.0 for a pair that exists, 0.0 for a point past the block (padding particle lanes are never read back).
  //
  // Instruction diet of round 3 (this kernel issues ~one f64 instruction per 3.3 cycles per SIMD; 78 -> 73 per pair).
  // Ts and d² keep the search kernel's and the oracle's unfused expressions: the winner was certified for exactly that Ts,
  // the mask compares exactly that d², and at map-frame coordinates of kilometres one rounding of Ts is 1e-12 m — a fused
  // transform (nine instructions fewer) was built and moved H by 1e-11 relative there
  // (test_map_frame_coordinates_far_from_the_origin).  What changed:
  //  * sqrt and the division are v_rsq_f64 / v_rcp_f64 seeds (2^-24 relative, measured: tests/microbench/
  //    f64_seed_accuracy.hip) refined by hand: root = one Goldschmidt step on s = x·r with h = r/2 and one residual step,
  //    quotient = two Newton steps; both come out within 0.5 ulp of the exact value on 4 M inputs (same microbenchmark).
  //    The compiler's IEEE sequences spend twelve more instructions on scaling for denormals and on special cases that
  //    cannot occur here.  The weight is formed as 1 / (1 + (3/d)·|e|) with 3/d rounded once per launch: w agrees with
  //    the reference's (d / (d + 3|e|))² to a few 2^-52 relative (tests hold the raw sums to 1e-12 against the f64 kernel);
  //  * mask·s is not formed: w·(mask·s) = (mask·w)·s exactly, because the mask is 0 or 1.
  const double c3d = 3.0 / a.max_dist;
  auto accumulate = [&](double on, double s0, double s1, double s2, double q0, double q1, double q2) {
    const double T0 = (s0 * Rt[0] + s1 * Rt[1] + s2 * Rt[2]) + tt[0];   // SVNICP.cpp:62-64, the search kernel's expression
    const double T1 = (s0 * Rt[3] + s1 * Rt[4] + s2 * Rt[5]) + tt[1];
    const double T2 = (s0 * Rt[6] + s1 * Rt[7] + s2 * Rt[8]) + tt[2];
    const double dx = T0 - q0, dy = T1 - q1, dz = T2 - q2;
    const double best = (dx * dx + dy * dy) + dz * dz;   // exact d² of the winner (knn_cpu.cpp:43-50 order)
    const double mf = best < a.max_dist ? on : 0.0;       // point_filter, SVGDICP.cpp:331-333 (squared distance against max_dist)
    const double x = mf * best;                           // 0 for a rejected row; NaN stays NaN (a non-finite point)
    // |e| = sqrt(x), SVNICP.cpp:120 on the masked rows: rsq seed (x + 2^-1000 keeps x = 0 finite: 0·2^500 = 0), s = x·r,
    // one coupled Newton step on (s, h = r/2), one residual step on s
    const double r0 = __builtin_amdgcn_rsq(x + 0x1p-1000);
    const double sa = x * r0, h0 = 0.5 * r0;
    const double ea = fma(-h0, sa, 0.5);
    const double sb = fma(sa, ea, sa);                    // 1.5·2^-48 relative
    const double nn = fma(fma(-sb, sb, x), h0, sb);       // residual step: h0's 2^-24 is enough here
    // wq = d / (d + 3|e|) = 1 / (1 + (3/d)|e|), SVNICP.cpp:121-122: rcp seed + two Newton steps; exactly 1 for a rejected row
    const double den = fma(c3d, nn, 1.0);
    const double y0 = __builtin_amdgcn_rcp(den);
    const double y1 = fma(y0, fma(-den, y0, 1.0), y0);
    const double wq = fma(y1, fma(-den, y1, 1.0), y1);
    const double w = on * (wq * wq);                      // SVNICP.cpp:122
    const double we = mf * w;
    const double e0 = we * dx, e1 = we * dy, e2 = we * dz;  // SVNICP.cpp:119,123
    const double w0 = we * s0, w1 = we * s1, w2 = we * s2;
    acc[0] += w;
    acc[1] += w0; acc[2] += w1; acc[3] += w2;
    // SVGD mode needs count_nonzero(mask·Ts summed over xyz) (SVGDICP.cpp:404) instead of Σw·s_x²
    if (PLAIN ? SVGD : (a.svgd != 0)) acc[4] += (((T0 + T1) + T2) != 0.0) ? mf : 0.0;
    else acc[4] = fma(w0, s0, acc[4]);
    acc[5] = fma(w0, s1, acc[5]); acc[6] = fma(w0, s2, acc[6]);
    acc[7] = fma(w1, s1, acc[7]); acc[8] = fma(w1, s2, acc[8]); acc[9] = fma(w2, s2, acc[9]);
    acc[10] += e0; acc[11] += e1; acc[12] += e2;
    acc[13] = fma(e0, s0, acc[13]); acc[14] = fma(e0, s1, acc[14]); acc[15] = fma(e0, s2, acc[15]);
    acc[16] = fma(e1, s0, acc[16]); acc[17] = fma(e1, s1, acc[17]); acc[18] = fma(e1, s2, acc[18]);
    acc[19] = fma(e2, s0, acc[19]); acc[20] = fma(e2, s1, acc[20]); acc[21] = fma(e2, s2, acc[21]);
  };

  const int64_t n0 = blk_lo + wb * BW;
  int kbn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) kbn[u] = load_kb(n0 + u * STEP);
  for (int64_t n = n0; n < blk_hi; n += U * STEP) {  // wave-uniform
    int kb[U];
    int64_t ti[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {   // target index of the winner; clamped like k_build_table3
      kb[u] = kbn[u];
      const int64_t b = n + u * STEP + bs;
      const int64_t bl = b < blk_hi ? b : blk_lo;
      const int64_t t = (PLAIN || fullp) ? (int64_t)kb[u] : (int64_t)ccand[(size_t)bl * K + kb[u]];
      ti[u] = t < 0 ? 0 : (t >= a.M ? a.M - 1 : t);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) kbn[u] = load_kb(n + (U + u) * STEP);
    double q[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const SVNICP_CONST_AS double* r = ctgt + 3 * ti[u];
      q[u][0] = r[0]; q[u][1] = r[1]; q[u][2] = r[2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t b = n + u * STEP + bs;
      const bool valid = pvalid && b < blk_hi;
      const int64_t bl = b < blk_hi ? b : blk_lo;
      const SVNICP_CONST_AS double* sp = csrc + 3 * bl;
      const double s0 = sp[0], s1 = sp[1], s2 = sp[2];
      if (!PLAIN && a.corr && valid) a.corr[(size_t)p * a.B + b] = kb[u];
      accumulate(b < blk_hi ? 1.0 : 0.0, s0, s1, s2, q[u][0], q[u][1], q[u][2]);
    }
  }

#pragma unroll
  for (int off = PW; off < kWave; off <<= 1) {
#pragma unroll
    for (int i = 0; i < kNSums; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
  }
  if constexpr (WB > 1) {
    double* red = lds;
    if (wb > 0 && bs == 0) {
      double* r = red + ((size_t)(wb - 1) * (WP * PW) + wp * PW + pl) * kNSums;
#pragma unroll
      for (int i = 0; i < kNSums; ++i) r[i] = acc[i];
    }
    __syncthreads();
    if (wb == 0 && bs == 0) {
      for (int o = 0; o < WB - 1; ++o) {
        const double* r = red + ((size_t)o * (WP * PW) + wp * PW + pl) * kNSums;
#pragma unroll
        for (int i = 0; i < kNSums; ++i) acc[i] += r[i];
      }
    }
  }
  if (wb == 0 && bs == 0) {
    double* out = a.partial + ((size_t)bx * a.Ppad + pidx) * kNSums;
#pragma unroll
    for (int i = 0; i < kNSums; ++i) out[i] = acc[i];
  }
}

// launch bounds: the plain SVN variant reaches 127 VGPRs (four waves per SIMD) on its own and schedules worse when forced
// (80 -> 91 us); the plain SVGD variant needs the bound to come down from 135 (92 -> 82 us)
template <int PW, int WP, bool PLAIN, bool SVGD = false>
__global__ __launch_bounds__(NT, (PLAIN && SVGD) ? 4 : 3) void k_stein_accumulate_w(AccumArgs a) {
  extern __shared__ __align__(16) double lds[];
  accumulate_body<PW, WP, PLAIN, SVGD>(a, xcd_block((int)blockIdx.x, (int)gridDim.x), (int)blockIdx.y, lds);
}

}  // namespace
}  // namespace svnicp
