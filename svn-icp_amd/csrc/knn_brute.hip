// knn_brute.hip — Stage A for SMALL registrations: exact top-K by brute force, one launch, no target layout.
//
// The scan-to-map loop hands the solver ~1 000 source points against a ~50 000-point local map
// (OdometryPipeline.cpp:559-560; shipped config/*.yaml), BASELINE C1 is 4 096 x 8 192.  At these sizes the Morton-tile
// path is all fixed cost: bounding box, two radix sorts, the re-ordered target copies, three kernels whose grids do not
// fill the chip (k_knn_seed: 18 workgroups for 1 113 queries, 140 us) and three fallback launches — 0.26–0.44 ms for
// 3e7–6e7 point pairs.  k_knn_brute visits every pair: a float32 score (relative to an origin near the queries, with a
// proven error bound: sweep_pass) decides which pairs get the float64 evaluation with the reference's arithmetic
// (SVGDICP.cpp:201-215, knn_cpu.cpp:35-67: q = (s·R0ᵀ) + t0, d² = ((dx·dx)+dy·dy)+dz·dz unfused); every target that can be
// among the K nearest gets it, and the selection runs on those float64 values (non-negative doubles order like unsigned
// integers) — the rows are the oracle's bit for bit:
//   a workgroup = 1..6 queries (chosen per launch, see QB below) x all targets (1024 threads, thread <-> every 1024th target, AoS rows as given: no copy of the cloud);
//   pass A   every thread keeps the minimum float32 score of ITS targets: 1024 values from 1024 distinct targets per query,
//            so their K-th smallest (bisection on the bit patterns, one wave per query), widened by the error bound, is a
//            float64 bound L that at least K targets meet — and a tight one: the K nearest targets mostly fall to different
//            threads (the trick of k_knn_seed);
//   pass B   the targets whose float32 score can belong to a d² <= L are evaluated in float64 and those within L (typically
//            1.2–2 K of them) collected into an LDS pool of 256 entries, ranked by counting under (d², original index) and
//            the first K written in ascending order — the contract of every stage-A kernel here (ties by lowest index,
//            zero padding when fewer than K targets exist, NaN distances never selected).
// A query whose bound admits more than 256 targets (dense duplicates, exact ties) takes the general path instead: a
// histogram of a 10-bit key of d² (exponent + 3 mantissa bits, window 2^-64 … 2^64) under the bound, and if the bins up to
// the K-th still hold more than 256 targets, radix select over the full 64 bits of d², 10 bits per pass, then over the 32
// bits of the index for exact ties — always terminates with at most 256 collected targets, so this kernel has no
// fallback launches.
#include "kernels.hpp"

namespace svnicp {

namespace {

// QB (template parameter): queries per workgroup, 1..6 (wave q analyses query q's histogram).  The targets a thread loads are
// shared by the workgroup's queries, so more queries per workgroup cost less per query — but the kernel's 97+ registers at
// 1024 threads mean ONE workgroup per CU, and a launch of a few more workgroups than CUs runs two rounds, the second nearly
// empty: launch_knn_brute picks the QB with the least rounds x time per round (1 113 queries, the scan-to-map loop's size, on
// 256 CUs: 279 workgroups of four = two rounds; 223 of five = one).
constexpr int kQBMax = 6;
constexpr int kCap = 256;       // pool entries per query (four threads per entry in the ranking)
constexpr int kNTB = 1024;      // threads per workgroup: 16 waves, four per SIMD — the sweeps are f64 dependency chains, occupancy hides them
constexpr int kBins = 1024;
constexpr int kWinBase = (1023 - 64) << 3;   // key 0 <-> d² < 2^-64 (incl. 0), key 1023 <-> d² >= 2^64 (incl. +inf)

// per-query selection state (LDS)
enum : int { M_WINDOW = 0, M_DBITS = 1, M_INDEX = 2, R_WINDOW = 3, R_DBITS = 4, R_INDEX = 5, R_ALL = 6, R_DONE = 7 };   // R_DONE: settled by passes A and B
struct QState {
  int mode;
  int m, mi;                 // matched high bits of d² / of the index
  int below;                 // targets known to lie below the current prefix
  int b0;                    // R_WINDOW: collect keys <= b0
  int filter;                // M_WINDOW: count keys <= filter only
  unsigned int ipfx;
  unsigned long long pfx;
};

__device__ __forceinline__ int win_key(unsigned long long bits) {
  int d = (int)(bits >> 49) - kWinBase;
  d = d < 0 ? 0 : d;
  return d > kBins - 1 ? kBins - 1 : d;
}

template <bool COLLECT, int kQB>
__device__ __forceinline__ void brute_pass(const KnnBruteArgs& a, const double (&qx)[kQB], const double (&qy)[kQB], const double (&qz)[kQB],
                                           int nq, const QState* st, unsigned int (*hist)[kBins], double (*pd)[kCap], int (*pi)[kCap],
                                           unsigned int* pn, int64_t j0, int64_t stride, int64_t n_steps, bool only_window) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  QState s[kQB];
#pragma unroll
  for (int q = 0; q < kQB; ++q) s[q] = st[q];   // wave-uniform
  for (int64_t it = 0; it < n_steps; ++it) {   // workgroup-uniform trip count
    const int64_t j = (j0 + it * kNTB + tid) * stride;
    const bool in = j < a.M;
    const double* tp = a.tgt + 3 * (in ? j : 0);
    const double tx = tp[0], ty = tp[1], tz = tp[2];
    const unsigned int idx = (unsigned int)j;
#pragma unroll
    for (int q = 0; q < kQB; ++q) {
      if (q >= nq) break;
      const int mode = s[q].mode;
      if (!COLLECT && (mode >= R_WINDOW || (only_window && mode != M_WINDOW))) continue;
      if (mode == R_DONE) continue;
      const double dx = qx[q] - tx, dy = qy[q] - ty, dz = qz[q] - tz;
      const double d = (dx * dx + dy * dy) + dz * dz;   // knn_cpu.cpp:43-50 order, unfused
      const unsigned long long bits = (unsigned long long)__double_as_longlong(d);
      const bool valid = in && d == d;                  // a NaN distance is never a neighbour (strict '<' insertion, knn_cpu.cpp:46)
      if (COLLECT) {
        bool take;
        if (mode == R_WINDOW) take = valid && win_key(bits) <= s[q].b0;
        else if (mode == R_DBITS) take = valid && (s[q].m >= 64 ? bits : (bits >> (64 - s[q].m))) <= s[q].pfx;
        else if (mode == R_INDEX) take = valid && (bits < s[q].pfx || (bits == s[q].pfx && (s[q].mi >= 32 ? idx : (idx >> (32 - s[q].mi))) <= s[q].ipfx));
        else take = valid;
        if (take) {
          const unsigned int pos = atomicAdd(&pn[q], 1u);
          if (pos < (unsigned int)kCap) { pd[q][pos] = d; pi[q][pos] = (int)idx; }
        }
      } else {
        bool ok;
        int dg;
        if (mode == M_WINDOW) { dg = win_key(bits); ok = valid && dg <= s[q].filter; }
        else if (mode == M_DBITS) {
          const int m = s[q].m, w = 64 - m < 10 ? 64 - m : 10;
          ok = valid && (m == 0 || (bits >> (64 - m)) == s[q].pfx);
          dg = (int)((bits >> (64 - m - w)) & ((1ull << w) - 1ull));
        } else {
          const int mi = s[q].mi, w = 32 - mi < 10 ? 32 - mi : 10;
          ok = valid && bits == s[q].pfx && (mi == 0 || (idx >> (32 - mi)) == s[q].ipfx);
          dg = (int)((idx >> (32 - mi - w)) & ((1u << w) - 1u));
        }
        // one LDS update per distinct key of the wave: neighbouring targets lie at similar distances, and 64 lanes adding
        // to one address are served one after the other
        unsigned long long act = __ballot(ok);
        while (act) {
          const int L = (int)__builtin_ctzll(act);
          const int dL = __builtin_amdgcn_readlane(dg, L);
          const unsigned long long same = __ballot(ok && dg == dL);
          if (lane == L) atomicAdd(&hist[q][dL], (unsigned int)__builtin_popcountll(same));
          act &= ~same;
        }
      }
    }
  }
}

// pass A / pass B: four targets per thread and trip, the next four already in flight.  Both sweeps score every pair in
// FLOAT32, relative to an origin o near the workgroup's queries (x' = fl32(fl64(x − o)), δ = fl32(q' − t'),
// s = fma(δz, δz, fma(δy, δy, δx·δx))): a float64 evaluation is eight half-rate instructions per pair, this one six full-rate
// ones — and only decides who gets the exact evaluation, never the result.  With u = 2^-24, E = the largest |x'| of all
// targets and the workgroup's queries, c = 3.46411·u·E + 1e-37 and Δ = q − t (exact):
//     |x' − (x − o)| <= 1.000001·u·E per coordinate (two roundings of a value of magnitude <= E(1 + 1.1u)),
//     |δ_i − (q'_i − t'_i)| <= 1.0000001·u·|δ_i|,                hence  ‖δ − Δ‖ <= 1.0000001·u·‖δ‖ + c,
//     s ∈ ‖δ‖²·[(1 − u)³, (1 + u)³] (+- 3.6e-38 should products be flushed), hence ‖δ‖² <= s(1 + 3.1u) + 1e-37 and
//                                                                             s <= ‖δ‖²(1 + 3.1u) + 1e-37.
//   BOUND:   per-thread minimum of s per query (v_min_f32 skips NaN: a NaN never replaces a number; rows past the end are
//            NaN) and per-thread maximum of |t'|.  The K-th smallest v of a query's 1024 minima belongs to K distinct
//            targets with s <= v, so their float64 d² is at most  L = ((1 + 1.0000001u)·sqrt(v(1 + 3.1u) + 1e-37) + c)²
//            (·(1 + 1e-12) for the roundings of float64's own d² and of this expression): at least K targets have d² <= L.
//   !BOUND:  a target with float64 d² <= L has  s <= τ = (((sqrt(L)(1 + 1e-15) + c) / (1 − 1.0000001u))²(1 + 3.1u) + 1e-37)
//            (·(1 + 1e-12), rounded UP to float32): only pairs with s <= τ are evaluated in float64 (the reference's
//            expression) and collected when bits(d²) <= bits(L) — every target within L is, so the pool holds the K nearest.
// Neither E nor the origin can make a result wrong, only the pool large (a far outlier, a non-finite query: E = inf, L = inf,
// everything passes, the pool overflows and the query goes through the float64 histogram path below).
constexpr double kU32 = 5.9604644775390625e-08;   // 2^-24
struct SweepQ { float x, y, z; };
template <bool BOUND, int kQB>
__device__ __forceinline__ void sweep_pass(const KnnBruteArgs& a, const double (&org)[3], const SweepQ (&qf)[kQB], const double (*sq)[3],
                                           float (&mn)[kQB], float& emax, const float (&tau)[kQB], const unsigned long long (&lim)[kQB],
                                           double (*pd)[kCap], int (*pi)[kCap], unsigned int* pn, int64_t n_steps) {
  constexpr int U = 4;
  const int tid = threadIdx.x;
  double tx[U], ty[U], tz[U], nx[U], ny[U], nz[U];
  // Which thread sees which target decides how good the bound is: scan order is (azimuth, beam), so with thread = j mod 1024
  // the neighbours of a query — a few beams x a few dozen azimuths — all fall to the same ~30 threads and the K-th smallest
  // of the minima lies far out.  Each block of 1024 targets is therefore rotated by a pseudo-random amount (a wave
  // still reads 64 consecutive rows, wrap-around aside): the blocks' neighbours land on different threads.
  // (32-bit row arithmetic: M < 2^31, knn_brute_applicable; the byte offset is one 64-bit multiply-add)
  const unsigned int Mu = (unsigned int)a.M;
  auto target_of = [&](unsigned int step) -> unsigned int {
    const unsigned int rot = (step * 0x9E3779B1u) >> 22;
    return step * (unsigned int)kNTB + (((unsigned int)tid + rot) & (unsigned int)(kNTB - 1));
  };
  auto fetch = [&](unsigned int it, double (&x)[U], double (&y)[U], double (&z)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned int j = target_of(it + u);
      j = j < Mu ? j : 0u;
      const double* tp = reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.tgt) + (unsigned long long)j * 24ull);
      x[u] = tp[0]; y[u] = tp[1]; z[u] = tp[2];
    }
  };
  fetch(0u, nx, ny, nz);
  const unsigned int steps = (unsigned int)n_steps;
  for (unsigned int it = 0; it < steps; it += U) {   // workgroup-uniform
#pragma unroll
    for (int u = 0; u < U; ++u) { tx[u] = nx[u]; ty[u] = ny[u]; tz[u] = nz[u]; }
    if (it + U < steps) fetch(it + U, nx, ny, nz);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned int j = target_of(it + u);
      const bool in = j < Mu;
      const float fx = in ? (float)(tx[u] - org[0]) : __builtin_nanf("");
      const float fy = in ? (float)(ty[u] - org[1]) : __builtin_nanf("");
      const float fz = in ? (float)(tz[u] - org[2]) : __builtin_nanf("");
      if (BOUND) emax = __builtin_fmaxf(emax, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fx), __builtin_fabsf(fy)), __builtin_fabsf(fz)));
#pragma unroll
      for (int q = 0; q < kQB; ++q) {
        const float dx = qf[q].x - fx, dy = qf[q].y - fy, dz = qf[q].z - fz;
        const float sc = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        if (BOUND) {
          mn[q] = __builtin_fminf(mn[q], sc);
        } else if (sc <= tau[q]) {   // (false for the NaN of a row past the end and for tau = -1 of a query that takes no part)
          const double ex = sq[q][0] - tx[u], ey = sq[q][1] - ty[u], ez = sq[q][2] - tz[u];
          const double d = (ex * ex + ey * ey) + ez * ez;   // knn_cpu.cpp:43-50 order, unfused
          const unsigned long long bits = (unsigned long long)__double_as_longlong(d);
          if (bits <= lim[q]) {      // (a NaN's bits lie above +inf's)
            const unsigned int pos = atomicAdd(&pn[q], 1u);
            if (pos < (unsigned int)kCap) { pd[q][pos] = d; pi[q][pos] = (int)j; }
          }
        }
      }
    }
  }
}

template <int kQB>
__global__ __launch_bounds__(kNTB) void k_knn_brute(KnnBruteArgs a) {
  __shared__ unsigned int s_hist[kQB][kBins];
  __shared__ double s_pd[kQB][kCap];
  __shared__ int s_pi[kQB][kCap];
  __shared__ unsigned int s_pn[kQB];
  __shared__ QState s_st[kQB];
  __shared__ int s_open;   // queries still selecting
  __shared__ unsigned long long s_bound[kQB];
  __shared__ float s_tau[kQB];
  __shared__ float s_min[kQB][kNTB];    // pass A: the threads' float32 minima
  __shared__ float s_emax[kNTB];        // … and their largest |t'|
  __shared__ double s_q[kQB][3];        // the queries in float64 (exact evaluations read them from here: broadcast reads, few)
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int64_t qb0 = a.b_lo + (int64_t)blockIdx.x * kQB;
  const int nq = (int)((a.b_hi - qb0) < kQB ? (a.b_hi - qb0) : kQB);
  const int K = a.K;

  // the queries: float64 in LDS, float32 relative to the origin (the workgroup's first query, 0 where that is not finite) in registers
  double org[3];
  SweepQ qf[kQB];
#pragma unroll
  for (int q = 0; q < kQB; ++q) {
    const int64_t b = qb0 + (q < nq ? q : 0);
    const double sx = a.src[3 * b], sy = a.src[3 * b + 1], sz = a.src[3 * b + 2];
    const double* R = a.pose.R0;
    const double x = (sx * R[0] + sy * R[1] + sz * R[2]) + a.pose.t0[0];   // SVGDICP.cpp:204, as in the other stage-A kernels
    const double y = (sx * R[3] + sy * R[4] + sz * R[5]) + a.pose.t0[1];
    const double z = (sx * R[6] + sy * R[7] + sz * R[8]) + a.pose.t0[2];
    if (q == 0) {
      org[0] = __builtin_fabs(x) < __builtin_huge_val() ? x : 0.0;   // (false for inf and NaN)
      org[1] = __builtin_fabs(y) < __builtin_huge_val() ? y : 0.0;
      org[2] = __builtin_fabs(z) < __builtin_huge_val() ? z : 0.0;
    }
    qf[q].x = (float)(x - org[0]); qf[q].y = (float)(y - org[1]); qf[q].z = (float)(z - org[2]);
    if (tid == 0) { s_q[q][0] = x; s_q[q][1] = y; s_q[q][2] = z; }
  }
  for (int i = tid; i < kQB * kBins; i += kNTB) (&s_hist[0][0])[i] = 0u;
  if (tid < kQB) s_pn[tid] = 0u;
  __syncthreads();

  // analysis of query `wave`'s histogram after a counting pass (one wave per query): the bin that holds the K'-th
  // smallest among the counted targets, K' = K − below
  auto analyse = [&](bool sample_pass) {
    const int q = wave;
    if (q < nq) {
      QState st = s_st[q];
      if (st.mode < R_WINDOW) {
        constexpr int PL = kBins / kWave;   // bins per lane
        unsigned int h[PL];
        unsigned int mine = 0u;
#pragma unroll
        for (int i = 0; i < PL; ++i) { h[i] = s_hist[q][lane * PL + i]; mine += h[i]; }
        unsigned int incl = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const unsigned int o = __shfl_up(incl, off, kWave); if (lane >= off) incl += o; }
        const unsigned int total = __shfl(incl, kWave - 1, kWave);
        const unsigned int want = (unsigned int)(K - st.below);   // >= 1
        if (total < want) {
          // fewer counted targets than asked for: only without a filter, i.e. fewer than K valid targets in all
          if (sample_pass) st.filter = kBins - 1;
          else { st.mode = R_ALL; }
        } else {
          const unsigned int excl = incl - mine;
          const bool here = excl < want && want <= incl;   // exactly one lane
          int bin = 0;
          unsigned int before = 0u, inbin = 0u;
          if (here) {
            unsigned int c = excl;
#pragma unroll
            for (int i = 0; i < PL; ++i) {
              if (c < want && want <= c + h[i]) { bin = lane * PL + i; before = c; inbin = h[i]; }
              c += h[i];
            }
          }
          const int src_lane = (int)__builtin_ctzll(__ballot(here));
          bin = __shfl(bin, src_lane, kWave); before = __shfl(before, src_lane, kWave); inbin = __shfl(inbin, src_lane, kWave);
          if (sample_pass) st.filter = bin;
          else if (st.mode == M_WINDOW) {
            if (before + inbin <= (unsigned int)kCap) { st.mode = R_WINDOW; st.b0 = bin; }
            else { st.mode = M_DBITS; st.m = 0; st.pfx = 0ull; st.below = 0; }   // general path, from the top bits
          } else if (st.mode == M_DBITS) {
            const int w = 64 - st.m < 10 ? 64 - st.m : 10;
            st.below += (int)before; st.pfx = (st.pfx << w) | (unsigned long long)bin; st.m += w;
            if ((unsigned int)st.below + inbin <= (unsigned int)kCap) st.mode = R_DBITS;
            else if (st.m >= 64) { st.mode = M_INDEX; st.mi = 0; st.ipfx = 0u; }
          } else {
            const int w = 32 - st.mi < 10 ? 32 - st.mi : 10;
            st.below += (int)before; st.ipfx = (st.ipfx << w) | (unsigned int)bin; st.mi += w;
            if ((unsigned int)st.below + inbin <= (unsigned int)kCap || st.mi >= 32) st.mode = R_INDEX;
          }
        }
        if (lane == 0) { s_st[q] = st; if (st.mode < R_WINDOW) atomicAdd(&s_open, 1); }
        for (int i = lane; i < kBins; i += kWave) s_hist[q][i] = 0u;
      }
    }
  };

  const int64_t steps_all = (a.M + kNTB - 1) / kNTB;
  long long t_prev = a.phase_cycles ? (long long)__builtin_amdgcn_s_memtime() : 0;
  auto stamp = [&](int ph) {   // debug = 1: cycles of thread 0 per phase, summed over the workgroups
    if (a.phase_cycles && tid == 0) { const long long t = (long long)__builtin_amdgcn_s_memtime(); atomicAdd(&a.phase_cycles[ph], (unsigned long long)(t - t_prev)); t_prev = t; }
  };
  unsigned long long lim[kQB];
  float tau[kQB];
  // pass A: per-thread float32 minima -> a bound at least K targets meet
  {
    float mn[kQB];
    float emax = 0.0f;
#pragma unroll
    for (int q = 0; q < kQB; ++q) { mn[q] = __builtin_huge_valf(); lim[q] = 0ull; tau[q] = -1.0f; }
    sweep_pass<true, kQB>(a, org, qf, s_q, mn, emax, tau, lim, s_pd, s_pi, s_pn, steps_all);
#pragma unroll
    for (int q = 0; q < kQB; ++q) s_min[q][tid] = mn[q];
    s_emax[tid] = emax;
  }
  __syncthreads();
  stamp(0);
  if (wave < nq) {   // wave q: the K-th smallest of query q's 1024 minima, bit by bit (non-negative floats order like their bits)
    const int q = wave;
    constexpr int NK = kNTB / kWave;
    unsigned int key[NK];
    float e = 0.0f;
#pragma unroll
    for (int i = 0; i < NK; ++i) { key[i] = __float_as_uint(s_min[q][lane + kWave * i]); e = __builtin_fmaxf(e, s_emax[lane + kWave * i]); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e = __builtin_fmaxf(e, __shfl_xor(e, off, kWave));
#pragma unroll
    for (int r = 0; r < kQB; ++r)   // E covers the workgroup's queries as well (wave-uniform registers)
      e = __builtin_fmaxf(e, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(qf[r].x), __builtin_fabsf(qf[r].y)), __builtin_fabsf(qf[r].z)));
    // on the 16 bits below the sign (exponent, 8 mantissa bits): the K-th smallest lies in [vh, vh | 0x7fff], and the upper
    // end is the bound — 2^-8 looser in s (0.2 % in radius, a handful of pool entries), half the iterations (9 us -> 5 us)
    unsigned int vh = 0u;
    for (int bit = 30; bit >= 15; --bit) {   // +inf (a thread without a target) is 0x7f800000: bit 31 is never needed
      const unsigned int cand = vh | (1u << bit);
      int c = 0;
#pragma unroll
      for (int i = 0; i < NK; ++i) c += __builtin_popcountll(__ballot(key[i] < cand));
      if (c < K) vh = cand;   // fewer than K minima below: the K-th is at or above the candidate
    }
    vh |= 0x7fffu;
    // the bound in float64 and the float32 filter that goes with it (sweep_pass's comment)
    const double v = vh >= 0x7f800000u ? __builtin_huge_val() : (double)__uint_as_float(vh), E = (double)e;   // (fewer than K targets in all: +inf — every number passes)
    const double c = 3.46411 * kU32 * E + 1e-37;
    const double r = (1.0 + 1.0000001 * kU32) * sqrt(v * (1.0 + 3.1 * kU32) + 1e-37) + c;
    const double L = (r * r) * (1.0 + 1e-12);
    const bool fin = L < __builtin_huge_val();                       // (false for inf and NaN)
    const double rt = (sqrt(L) * (1.0 + 1e-15) + c) / (1.0 - 1.0000001 * kU32);
    const double tq = ((rt * rt) * (1.0 + 3.1 * kU32) + 1e-37) * (1.0 + 1e-12);
    float tf = (float)tq;
    if ((double)tf < tq) tf = __uint_as_float(__float_as_uint(tf) + 1u);   // round up (tq > 0; the next float of the largest finite one is +inf)
    if (!fin || !(tq < __builtin_huge_val())) tf = __builtin_huge_valf();
    if (lane == 0) { s_bound[q] = fin ? (unsigned long long)__double_as_longlong(L) : 0x7ff0000000000000ull; s_tau[q] = tf; }
  }
  __syncthreads();
  stamp(1);
#pragma unroll
  for (int q = 0; q < kQB; ++q) {
    const unsigned long long l = s_bound[q < nq ? q : 0];
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)l), hi = __builtin_amdgcn_readfirstlane((unsigned int)(l >> 32));
    lim[q] = ((unsigned long long)hi << 32) | lo;
    tau[q] = q < nq ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_tau[q]))) : -1.0f;
  }
  // pass B: collect within the bound
  {
    float mn[kQB];
    float emax = 0.0f;
#pragma unroll
    for (int q = 0; q < kQB; ++q) mn[q] = 0.0f;
    sweep_pass<false, kQB>(a, org, qf, s_q, mn, emax, tau, lim, s_pd, s_pi, s_pn, steps_all);
  }
  __syncthreads();
  stamp(2);
  // overflow (more than 256 targets within the bound): those queries go through the histogram / general path, the
  // window key of the bound as the filter
  bool general = false;
#pragma unroll
  for (int q = 0; q < kQB; ++q) general = general || (q < nq && s_pn[q] > (unsigned int)kCap);
  if (general) {   // workgroup-uniform
    double qx[kQB], qy[kQB], qz[kQB];
#pragma unroll
    for (int q = 0; q < kQB; ++q) { qx[q] = s_q[q][0]; qy[q] = s_q[q][1]; qz[q] = s_q[q][2]; }
    __syncthreads();
    if (tid < kQB) {
      QState z{};
      if (tid < nq && s_pn[tid] > (unsigned int)kCap) { z.mode = M_WINDOW; z.filter = win_key(s_bound[tid]); s_pn[tid] = 0u; }
      else z.mode = R_DONE;
      s_st[tid] = z;
    }
    __syncthreads();
    for (int round = 0; round < 13; ++round) {   // window histogram, then 7 + 4 rounds at most
      brute_pass<false, kQB>(a, qx, qy, qz, nq, s_st, s_hist, s_pd, s_pi, s_pn, 0, 1, steps_all, false);
      __syncthreads();
      if (tid == 0) s_open = 0;
      __syncthreads();
      analyse(false);
      __syncthreads();
      if (s_open == 0) break;
      __syncthreads();
    }
    brute_pass<true, kQB>(a, qx, qy, qz, nq, s_st, s_hist, s_pd, s_pi, s_pn, 0, 1, steps_all, false);
    __syncthreads();
  }
  stamp(3);

  // rank by counting under (d², original index), ascending output (pad like torch::full(..., 0), knn_cpu.cpp:25-26)
  for (int q = 0; q < nq; ++q) {   // four threads per pool entry, a quarter of the pool each; all 1024 threads on one query at a time
    const int e = tid >> 2, part = tid & 3;
    const int64_t b = qb0 + q;
    const int n = (int)(s_pn[q] < (unsigned int)kCap ? s_pn[q] : (unsigned int)kCap);
    const int n4 = (n + 3) >> 2;
    const bool has = e < n;
    const double de = has ? s_pd[q][e] : 0.0;
    const int ie = has ? s_pi[q][e] : 0;
    int rank = 0;
    if (has) {
      int f = part * n4;
      const int fe = f + n4 < n ? f + n4 : n;
      for (; f + 8 <= fe; f += 8) {   // eight broadcast reads in flight per trip
        double df[8];
        int jf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { df[u] = s_pd[q][f + u]; jf[u] = s_pi[q][f + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (df[u] < de || (df[u] == de && jf[u] < ie)) ? 1 : 0;
      }
      for (; f < fe; ++f) {
        const double df = s_pd[q][f];
        const int jf = s_pi[q][f];
        rank += (df < de || (df == de && jf < ie)) ? 1 : 0;
      }
    }
    rank += __shfl_xor(rank, 1, kWave);   // (the four threads of an entry are neighbours in a wave; threads without an entry add 0)
    rank += __shfl_xor(rank, 2, kWave);
    if (has && part == 0 && rank < K) { a.out_idx[b * K + rank] = ie; a.out_d2[b * K + rank] = de; }
    for (int k = n + tid; k < K; k += kNTB) { a.out_idx[b * K + k] = 0; a.out_d2[b * K + k] = 0.0; }
  }
  stamp(4);
}

}  // namespace

// the sizes this kernel is for: every pair is scored twice (float32)
bool knn_brute_applicable(int64_t B, int64_t M, int K) {
  return K >= 1 && K <= 128 && B >= 1 && M >= 1 && M < (1ll << 31) && (double)B * (double)M <= 268435456.0;   // 2^28 pairs (2.4e8: 0.20 ms against 0.37 ms for the tile chain with its sorts; 5.4e8: 0.41 against 0.42)
}

template <int QB>
static hipError_t launch_qb(const KnnBruteArgs& a, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_knn_brute<QB>, dim3((unsigned)((n + QB - 1) / QB)), dim3(kNTB), 0, st, a);
  return hipGetLastError();
}

// queries per workgroup for n queries on num_cus CUs (one workgroup per CU at a time): fewest rounds x (time of one round).
// One round, measured at 50 000 targets, K = 100 (tests/gpu_time_brute.py): 39, 47, 54, 59, 67, 75 us for 1..6 queries
// (43, 58, 66, 77, 93, 104 us before the float32 pre-filter, the 16-step bound and the four-thread ranking).
// 1 113 queries, four per workgroup as at first: 152 us; five: 98 us; with the pre-filter etc.: 70 us.
int knn_brute_queries_per_block(int64_t n, int num_cus) {
  static const int round_us[kQBMax + 1] = {0, 39, 47, 54, 59, 67, 75};
  const int64_t cus = num_cus > 0 ? num_cus : 256;
  int best = 4;
  int64_t best_cost = -1;
  for (int qb = 1; qb <= kQBMax; ++qb) {
    const int64_t blocks = (n + qb - 1) / qb, rounds = (blocks + cus - 1) / cus, cost = rounds * round_us[qb];
    if (best_cost < 0 || cost < best_cost) { best = qb; best_cost = cost; }
  }
  return best;
}

hipError_t launch_knn_brute(const KnnBruteArgs& a, int num_cus, int queries_per_block, hipStream_t st) {
  const int64_t n = a.b_hi - a.b_lo;
  if (n <= 0) return hipSuccess;
  switch (queries_per_block > 0 ? queries_per_block : knn_brute_queries_per_block(n, num_cus)) {
    case 1: return launch_qb<1>(a, n, st);
    case 2: return launch_qb<2>(a, n, st);
    case 3: return launch_qb<3>(a, n, st);
    case 4: return launch_qb<4>(a, n, st);
    case 5: return launch_qb<5>(a, n, st);
    case 6: return launch_qb<6>(a, n, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace svnicp
