// stein_split.hip — Stage B as two kernels per iteration: winner search, then float64 accumulation.
//
// Same results as the fused kernels (stein_iter.hip, stein_mfma.hip), bit for bit; the split exists
// for occupancy.  The fused MFMA kernel carries 22 f64 accumulators, the pose and a pending winner
// next to the MFMA tiles (168 VGPRs, ~2 resident waves per SIMD, VALU busy ~50 %, rocprofv3 PMC in
// profiles/).  Here
//   k_stein_search_mfma   finds, for every (source point, particle), the index of the nearest of the
//                         K candidates (SVGDICP.cpp:300-329 with knn.cu:204-251, K = 1) — float32 MFMA
//                         scores + rigorous ambiguity test + exact f64 fallback exactly as in
//                         stein_mfma.hip — and writes one byte per pair; no workgroup barrier, operands
//                         straight from global memory, ~100 VGPRs;
//   k_stein_accumulate_w  re-derives Ts in f64, gathers the winner, applies point_filter / weight and
//                         accumulates the 22 sums (SVGDICP.cpp:331-333, SVNICP.cpp:116-157) with the
//                         loads of the next two points in flight.
// Lane ↔ particle in both; the byte array is [B][Ppad] so a wave writes/reads 64 consecutive bytes.
#include "kernels.hpp"
#include "stein_common.hpp"

namespace svnicp {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

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
// search: error bound, packing and ambiguity test are those of stein_mfma.hip (see its header)
// ---------------------------------------------------------------------------------------------
// NRB row blocks of 16 candidates go through the matrix cores; with TAIL the (at most four) candidates 16·NRB …
// 16·NRB+3 are scored by the VALU in the owner lane instead of spending four mostly empty tiles on them (K = 100)
template <int PW, int WP, int NRB, bool TAIL>
__global__ __launch_bounds__(NT, 4) void k_stein_search_mfma(AccumArgs a) {
  if (a.ctl[0]) return;
  constexpr int BW = kWave / PW;   // source points per wave step (1, 2, 4)
  constexpr int WB = 4 / WP;       // waves along the source-point axis
  constexpr int CBP = PW / 16;     // 16-particle column blocks per source point (4, 2, 1)
  constexpr int NPT = 4 / CBP;     // distinct source points per wave step
  __shared__ float4 s_scr4[4][64];
  __shared__ float s_scrb[4][64];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WP, wb = wave / WP;
  const int pl = lane % PW, bs = lane / PW;
  const int mj = lane & 15, mk = lane >> 4;
  const int pidx = blockIdx.y * (WP * PW) + wp * PW + pl;
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
  const int K = a.K;
  float4* scr4 = s_scr4[wave];
  float* scrb = s_scrb[wave];
  const float* scr4f = reinterpret_cast<const float*>(scr4);
  const float kEps = 48.0f * 5.9604644775390625e-08f;
  const int64_t blk_lo = (int64_t)blockIdx.x * a.spts_per_block;
  const int64_t blk_hi = (blk_lo + a.spts_per_block < a.B) ? blk_lo + a.spts_per_block : a.B;

  for (int64_t n = blk_lo + wb * BW; n < blk_hi; n += WB * BW) {  // wave-uniform
    const int64_t b = n + bs;
    const bool inb = b < blk_hi;
    const bool valid = pvalid && inb;
    const int64_t bl = inb ? b : n;
    const double* sp = a.src + 3 * bl;
    const double* an = a.anchor + 3 * bl;               // first candidate = origin of the local frame
    const double s0 = sp[0], s1 = sp[1], s2 = sp[2];
    const double T0 = (s0 * Rt[0] + s1 * Rt[1] + s2 * Rt[2]) + tt[0];   // SVNICP.cpp:62-64
    const double T1 = (s0 * Rt[3] + s1 * Rt[4] + s2 * Rt[5]) + tt[1];
    const double T2 = (s0 * Rt[6] + s1 * Rt[7] + s2 * Rt[8]) + tt[2];
    const float xf0 = (float)(T0 - an[0]), xf1 = (float)(T1 - an[1]), xf2 = (float)(T2 - an[2]);
    const float X = __builtin_fmaxf(__builtin_fabsf(xf0), __builtin_fmaxf(__builtin_fabsf(xf1), __builtin_fabsf(xf2)));
    const float C = a.cmax[bl];
    const float E = kEps * (C + X) * (C + X);
    const float beta = __builtin_fmaf(xf0, xf0, __builtin_fmaf(xf1, xf1, xf2 * xf2)) + 4.0f * E;
    scr4[lane] = make_float4(-2.0f * xf0, -2.0f * xf1, -2.0f * xf2, 1.0f);
    scrb[lane] = beta;

    float av[NPT][8];
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      int64_t bq = n + q;
      bq = bq < blk_hi ? bq : n;
      const float4* rowp = a.tablea + (size_t)bq * 128 + lane;
      const float4 alo = rowp[0], ahi = rowp[64];
      av[q][0] = alo.x; av[q][1] = alo.y; av[q][2] = alo.z; av[q][3] = alo.w;
      av[q][4] = ahi.x; av[q][5] = ahi.y; av[q][6] = ahi.z; av[q][7] = ahi.w;
    }
    __builtin_amdgcn_wave_barrier();
    float bvv[4], bee[4], b1[4], b2[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      bvv[cb] = scr4f[(16 * cb + mj) * 4 + mk];
      bee[cb] = scrb[16 * cb + mj];
      b1[cb] = __builtin_huge_valf(); b2[cb] = __builtin_huge_valf();
    }
    auto tile = [&](int i) -> v4f {
      const int cb = i / NRB, rb = i % NRB;
      const v4f cin = {bee[cb], bee[cb], bee[cb], bee[cb]};
      return __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb / CBP][rb], bvv[cb], cin, 0, 0, 0);
    };
    v4f dcur = tile(0);
#pragma unroll
    for (int i = 0; i < 4 * NRB; ++i) {  // tile i+1 goes to the matrix pipe before the VALU consumes tile i
      v4f dnext = dcur;
      if (i + 1 < 4 * NRB) dnext = tile(i + 1);
      const int cb = i / NRB, rb = i % NRB;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float pk = pack_slot(dcur[v], 0x1fu, (unsigned int)(rb * 4 + v));
        b2[cb] = __builtin_amdgcn_fmed3f(b1[cb], b2[cb], pk);
        b1[cb] = imin_f(b1[cb], pk);
      }
      dcur = dnext;
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) b1[cb] = pack_slot(b1[cb], 0x60u, (unsigned int)mk << 5);
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {  // the four lanes that share a particle
      float o1[4], o2[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) { o1[cb] = __shfl_xor(b1[cb], off, kWave); o2[cb] = __shfl_xor(b2[cb], off, kWave); }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const float hi = imax_f(b1[cb], o1[cb]);
        b1[cb] = imin_f(b1[cb], o1[cb]);
        b2[cb] = imin_f(hi, imin_f(b2[cb], o2[cb]));
      }
    }
    float b1own = b1[0], b2own = b2[0];
#pragma unroll
    for (int cb = 1; cb < 4; ++cb) {
      if (mk == cb) { b1own = b1[cb]; b2own = b2[cb]; }
    }
    __builtin_amdgcn_wave_barrier();  // scratch is rewritten by the next step
    if constexpr (TAIL) {
      const float4* tl = a.tail + (size_t)bl * 4;
      const float m0 = -2.0f * xf0, m1 = -2.0f * xf1, m2 = -2.0f * xf2;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 c = tl[t];  // (c'x, c'y, c'z, |c'|²), a finite sentinel past K
        const float sc = __builtin_fmaf(c.x, m0, __builtin_fmaf(c.y, m1, __builtin_fmaf(c.z, m2, c.w + beta)));
        const float pk = pack_slot(sc, 0x7fu, (unsigned int)(NRB * 4 + t));  // slot of candidate 16·NRB + t, lane group 0
        b2own = __builtin_amdgcn_fmed3f(b1own, b2own, pk);
        b1own = imin_f(b1own, pk);
      }
    }

    const unsigned int wbits = __float_as_uint(b1own);
    int kb = (int)(((wbits & 0x1cu) << 2) | ((wbits >> 3) & 0xcu) | (wbits & 3u));  // 16·rb + 4·mk + v
    const float thr = 2.0f * E + 6.103515625e-05f * b2own + 1.0e-30f;
    const bool ambiguous = valid && (!(b2own - b1own > thr) || kb >= K);
    kb = kb < K ? kb : 0;
    unsigned long long am = __ballot(ambiguous);
    if (am) {  // rare: exact f64 nearest-of-K for the undecided lanes, candidate-parallel across the wave
      if (a.ambig_count && lane == 0) atomicAdd(a.ambig_count, 1);
      do {
        const int L = (int)__builtin_ctzll(am);
        am &= am - 1;
        const double t0 = rdlane_f64(T0, L), t1 = rdlane_f64(T1, L), t2 = rdlane_f64(T2, L);
        const int bsL = L / PW;
        const int32_t* ci = a.cand + (size_t)(n + bsL) * K;
        double bd = __builtin_huge_val(), d_first = 0.0;
        int bk = 0x7fffffff;
        for (int k = lane; k < K; k += kWave) {
          int64_t ti = ci[k];
          ti = ti < 0 ? 0 : (ti >= a.M ? a.M - 1 : ti);
          const double* r = a.tgt + 3 * ti;
          const double dx = t0 - r[0], dy = t1 - r[1], dz = t2 - r[2];
          const double d = (dx * dx + dy * dy) + dz * dz;   // knn_cpu.cpp:43-50 order, unfused
          if (k == 0) d_first = d;
          if (d < bd || (d == bd && k < bk)) { bd = d; bk = k; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const double od = __shfl_xor(bd, off, kWave);
          const int ok = __shfl_xor(bk, off, kWave);
          if (od < bd || (od == bd && ok < bk)) { bd = od; bk = ok; }
        }
        // the serial reference loop starts from candidate 0 and only replaces on '<': a NaN first
        // distance is never replaced, and an all-NaN row keeps index 0
        const double d0 = rdlane_f64(d_first, 0);
        const int ke = (d0 != d0 || bk == 0x7fffffff) ? 0 : bk;
        if (lane == L) kb = ke;
      } while (am);
    }
    if (inb) a.kbest[(size_t)b * a.Ppad + pidx] = (uint8_t)kb;
  }
}

// ---------------------------------------------------------------------------------------------
// accumulation from the winner bytes
// ---------------------------------------------------------------------------------------------
template <int PW, int WP>
__global__ __launch_bounds__(NT, 3) void k_stein_accumulate_w(AccumArgs a) {
  if (a.ctl[0]) return;
  constexpr int BW = kWave / PW;
  constexpr int WB = 4 / WP;
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WP, wb = wave / WP;
  const int pl = lane % PW, bs = lane / PW;
  const int pidx = blockIdx.y * (WP * PW) + wp * PW + pl;
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
  const int64_t blk_lo = (int64_t)blockIdx.x * a.pts_per_block;
  const int64_t blk_hi = (blk_lo + a.pts_per_block < a.B) ? blk_lo + a.pts_per_block : a.B;
  constexpr int STEP = WB * BW;

  // four-stage pipeline: winner byte of point n+3·STEP | its target index (n+2·STEP) | winner coordinates and Ts
  // (n+STEP) | sums of n — every load has a full step of other work between issue and use
  struct Stage { double s0, s1, s2, q0, q1, q2; bool valid; };  // raw loads only: nothing in fetch waits for memory
  auto load_kb = [&](int64_t n) -> int {
    const int64_t b = n + bs;
    return (pvalid && b < blk_hi) ? (int)a.kbest[(size_t)b * a.Ppad + pidx] : 0;
  };
  auto load_ti = [&](int64_t n, int kb) -> int64_t {  // target index of the winner; clamped like k_build_table3
    const int64_t b = n + bs;
    const int64_t bl = b < blk_hi ? b : blk_lo;
    int64_t ti = a.cand[(size_t)bl * K + kb];
    return ti < 0 ? 0 : (ti >= a.M ? a.M - 1 : ti);
  };
  auto fetch = [&](int64_t n, int64_t ti, int kb, Stage& st) {
    const int64_t b = n + bs;
    st.valid = pvalid && b < blk_hi;
    const int64_t bl = b < blk_hi ? b : blk_lo;
    const double* sp = a.src + 3 * bl;
    st.s0 = sp[0]; st.s1 = sp[1]; st.s2 = sp[2];
    const double* q = a.tgt + 3 * ti;
    st.q0 = q[0]; st.q1 = q[1]; st.q2 = q[2];
    if (a.corr && st.valid) a.corr[(size_t)p * a.B + b] = kb;
  };
  auto finish = [&](const Stage& st) {
    if (!st.valid) return;
    const double T0 = (st.s0 * Rt[0] + st.s1 * Rt[1] + st.s2 * Rt[2]) + tt[0];   // SVNICP.cpp:62-64
    const double T1 = (st.s0 * Rt[3] + st.s1 * Rt[4] + st.s2 * Rt[5]) + tt[1];
    const double T2 = (st.s0 * Rt[6] + st.s1 * Rt[7] + st.s2 * Rt[8]) + tt[2];
    const double dx = T0 - st.q0, dy = T1 - st.q1, dz = T2 - st.q2;
    const double best = (dx * dx + dy * dy) + dz * dz;   // exact d² of the winner (knn_cpu.cpp:43-50 order)
    double w = 1.0, e0 = 0.0, e1 = 0.0, e2 = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0;
    if (best < a.max_dist) {  // point_filter, SVGDICP.cpp:331-333
      const double nn = sqrt(best);                       // SVNICP.cpp:120
      const double wq = a.max_dist / (a.max_dist + 3 * nn);
      w = wq * wq;                                        // SVNICP.cpp:122
      e0 = w * dx; e1 = w * dy; e2 = w * dz;              // SVNICP.cpp:119,123
      n0 = st.s0; n1 = st.s1; n2 = st.s2;
    }
    const double w0 = w * n0, w1 = w * n1, w2 = w * n2;
    acc[0] += w;
    acc[1] += w0; acc[2] += w1; acc[3] += w2;
    // SVGD mode needs count_nonzero(mask·Ts summed over xyz) (SVGDICP.cpp:404) instead of Σw·s_x²
    acc[4] = a.svgd ? acc[4] + ((best < a.max_dist && ((T0 + T1) + T2) != 0.0) ? 1.0 : 0.0) : fma(w0, n0, acc[4]);
    acc[5] = fma(w0, n1, acc[5]); acc[6] = fma(w0, n2, acc[6]);
    acc[7] = fma(w1, n1, acc[7]); acc[8] = fma(w1, n2, acc[8]); acc[9] = fma(w2, n2, acc[9]);
    acc[10] += e0; acc[11] += e1; acc[12] += e2;
    acc[13] = fma(e0, n0, acc[13]); acc[14] = fma(e0, n1, acc[14]); acc[15] = fma(e0, n2, acc[15]);
    acc[16] = fma(e1, n0, acc[16]); acc[17] = fma(e1, n1, acc[17]); acc[18] = fma(e1, n2, acc[18]);
    acc[19] = fma(e2, n0, acc[19]); acc[20] = fma(e2, n1, acc[20]); acc[21] = fma(e2, n2, acc[21]);
  };

  const int64_t n0 = blk_lo + wb * BW;
  if (n0 < blk_hi) {
    // prologue: fill the pipeline (points past the range load harmless clamped addresses)
    int kb1 = load_kb(n0);
    int64_t ti1 = load_ti(n0, kb1);
    Stage cur;
    fetch(n0, ti1, kb1, cur);
    kb1 = load_kb(n0 + STEP);                 // point n+STEP: byte, then index
    ti1 = load_ti(n0 + STEP, kb1);
    int kb2 = load_kb(n0 + 2 * STEP);         // point n+2·STEP: byte
    for (int64_t n = n0; n < blk_hi; n += STEP) {  // wave-uniform
      Stage nxt;
      nxt.valid = false;
      if (n + STEP < blk_hi) fetch(n + STEP, ti1, kb1, nxt);
      kb1 = kb2;
      ti1 = load_ti(n + 2 * STEP, kb1);
      kb2 = load_kb(n + 3 * STEP);
      finish(cur);
      cur = nxt;
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
    double* out = a.partial + ((size_t)blockIdx.x * a.Ppad + pidx) * kNSums;
#pragma unroll
    for (int i = 0; i < kNSums; ++i) out[i] = acc[i];
  }
}

// (row blocks, tail) with an instantiation: K <= 16, 32, 64, 96, 100 (96 + four VALU candidates), 112, 128
inline int row_code_for(int K) { return K <= 16 ? 1 : K <= 32 ? 2 : K <= 64 ? 4 : K <= 96 ? 6 : K <= 100 ? 60 : K <= 112 ? 7 : 8; }

template <int PW, int WP, int NRB, bool TAIL>
hipError_t launch_s(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((k_stein_search_mfma<PW, WP, NRB, TAIL>), dim3(plan.sgrid_x, plan.grid_y), dim3(NT), 0, st, a);
  return hipGetLastError();
}
template <int PW, int WP>
hipError_t launch_srb(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  switch (row_code_for(a.K)) {
    case 1: return launch_s<PW, WP, 1, false>(plan, a, st);
    case 2: return launch_s<PW, WP, 2, false>(plan, a, st);
    case 4: return launch_s<PW, WP, 4, false>(plan, a, st);
    case 6: return launch_s<PW, WP, 6, false>(plan, a, st);
    case 60: return launch_s<PW, WP, 6, true>(plan, a, st);
    case 7: return launch_s<PW, WP, 7, false>(plan, a, st);
    default: return launch_s<PW, WP, 8, false>(plan, a, st);
  }
}
template <int PW, int WP>
hipError_t launch_w(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((k_stein_accumulate_w<PW, WP>), dim3(plan.grid_x, plan.grid_y), dim3(NT), plan.smem, st, a);
  return hipGetLastError();
}

template <int PW, int WP>
void occ_split(int K, size_t smem, int* search, int* accum) {
  int n = 0;
  hipError_t e;
  switch (row_code_for(K)) {
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_search_mfma<PW, WP, 1, false>, NT, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_search_mfma<PW, WP, 2, false>, NT, 0); break;
    case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_search_mfma<PW, WP, 4, false>, NT, 0); break;
    case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_search_mfma<PW, WP, 6, false>, NT, 0); break;
    case 60: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_search_mfma<PW, WP, 6, true>, NT, 0); break;
    case 7: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_search_mfma<PW, WP, 7, false>, NT, 0); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_search_mfma<PW, WP, 8, false>, NT, 0); break;
  }
  *search = (e != hipSuccess || n < 1) ? 4 : (n > 8 ? 8 : n);
  n = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_accumulate_w<PW, WP>, NT, smem);
  *accum = (e != hipSuccess || n < 1) ? 3 : (n > 8 ? 8 : n);
}

}  // namespace

// resident workgroups per CU of the two kernels (grids are sized to one resident round)
void split_occupancy_blocks(int PW, int WP, int K, size_t smem, int* search, int* accum) {
  switch (PW) {
    case 16: return occ_split<16, 1>(K, smem, search, accum);
    case 32: return occ_split<32, 1>(K, smem, search, accum);
    default:
      if (WP == 1) return occ_split<64, 1>(K, smem, search, accum);
      if (WP == 2) return occ_split<64, 2>(K, smem, search, accum);
      return occ_split<64, 4>(K, smem, search, accum);
  }
}

// search kernel, then (launch_accumulate_split) the accumulation kernel; api.hip brackets them separately
hipError_t launch_search_split(const AccumPlan& plan, AccumArgs a, hipStream_t st) {
  a.Ppad = plan.Ppad; a.pts_per_block = plan.pts_per_block; a.spts_per_block = plan.spts_per_block;
  switch (plan.PW) {
    case 16: return launch_srb<16, 1>(plan, a, st);
    case 32: return launch_srb<32, 1>(plan, a, st);
    default:
      if (plan.WP == 1) return launch_srb<64, 1>(plan, a, st);
      if (plan.WP == 2) return launch_srb<64, 2>(plan, a, st);
      return launch_srb<64, 4>(plan, a, st);
  }
}

hipError_t launch_accumulate_split(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  switch (plan.PW) {
    case 16: return launch_w<16, 1>(plan, a, st);
    case 32: return launch_w<32, 1>(plan, a, st);
    default:
      if (plan.WP == 1) return launch_w<64, 1>(plan, a, st);
      if (plan.WP == 2) return launch_w<64, 2>(plan, a, st);
      return launch_w<64, 4>(plan, a, st);
  }
}

}  // namespace svnicp
