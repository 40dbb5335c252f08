// stein_mfma.hip — Stage B with the nearest-of-K search on the float32 matrix cores.
//
// Same contract and same exact finish as k_stein_accumulate_f32 (stein_iter.hip): a float32 search
// picks the winner, a rigorous error bound decides whether that pick is provably the float64 argmin,
// undecided lanes re-run the reference's float64 loop, and the winner's d², mask, weight and the 22
// sums are computed in float64 — correspondences and sums are bit-identical to the f64 kernel.
// What changes is where the K·P scores per source point come from (SVGDICP.cpp:300-329 /
// knn.cu:204-251 with K = 1 in the reference):
//
//   S'[k][p] = β_p + cc_k + c'_k · m_p ,   m_p = −2x'_p ,  β_p = |x'_p|² + 4·EPS
//
// is a [K × 4]·[4 × P] product (rows (c'x, c'y, c'z, cc), columns (mx, my, mz, 1)) with β as the
// accumulator input, i.e. v_mfma_f32_16x16x4_f32 tiles of 16 candidates × 16 particles: 1024
// multiply-adds per 32 cycles instead of 3 VALU FMAs and one broadcast LDS read per (candidate, wave).
// An MFMA result register holds 4 candidates of ONE particle per lane, so the running minimum and
// second minimum stay lane-local: the candidate slot (5 bits) is packed into the low mantissa bits of
// the positive score (v_and_or), then v_med3 / v_min — 3 VALU ops per score.  Four lanes share a
// particle; they are merged once per source point (2 xor-shuffle steps).
//
// Error bound (u = 2^-24, C = max|c'|∞ of the point, X = |x'|∞ of the particle):
//   inputs: |cc − |y|²| <= 9.1uC², cross term 12.1uXC (as in stein_iter.hip);
//   MFMA: at most 4 additions on partial sums <= 3(C+X)²(1+…) and 3 product roundings <= 2uXC each,
//   whatever the internal order  =>  |S'_k − s'_k| <= 22u(C+X)²; EPS := 48·u·(C+X)² leaves a factor 2
//   for a truncating accumulator.  β >= |x'|² + 4·EPS keeps every score positive, so the float order
//   equals the integer order of the packed words.  Packing perturbs a score by < 128 ulp <= 2^-16·S'.
//   If  b2 − b1 > 2·EPS + 2^-14·b2 + 1e-30  then every other candidate is strictly farther in exact
//   arithmetic and the packed argmin is the f64 argmin (ties, padded duplicates, NaN/Inf never pass).
#include "kernels.hpp"
#include "stein_common.hpp"

namespace svnicp {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kRowBlocks = 8;          // 8 x 16 = 128 candidate rows at most
constexpr float kSentinelCC = 1.0e30f; // padded rows: finite, so packed words never become NaN patterns

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

template <int PW, int WP, int NRB>  // NRB row blocks of 16 candidates cover K (rows past K are sentinels)
__global__ __launch_bounds__(NT, 2) void k_stein_accumulate_mfma(AccumArgs a) {
  if (a.ctl[0]) return;
  constexpr int BW = kWave / PW;   // source points per wave step (1, 2, 4)
  constexpr int WB = 4 / WP;       // waves along the source-point axis
  constexpr int CBP = PW / 16;     // 16-particle column blocks per source point (4, 2, 1)
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const int wp = wave % WP, wb = wave / WP;
  const int pl = lane % PW, bs = lane / PW;
  const int mj = lane & 15, mk = lane >> 4;  // MFMA operand coordinates of this lane
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

  const int K = a.K, TP = a.TP;
  // LDS: rowsA [TP][2][64] float4 | spts [TP][3] f64 | anch [TP][3] f64 | cmx [TP'] f32 | per wave: scr4 [64] float4, scrb [64] f32
  float4* rowsA = reinterpret_cast<float4*>(lds);
  double* spts = reinterpret_cast<double*>(rowsA + (size_t)TP * 128);
  double* anch = spts + 3 * TP;
  float* cmx = reinterpret_cast<float*>(anch + 3 * TP);
  float4* scr4 = reinterpret_cast<float4*>(cmx + ((TP + 3) & ~3)) + wave * 64;
  float* scrb = reinterpret_cast<float*>(reinterpret_cast<float4*>(cmx + ((TP + 3) & ~3)) + 4 * 64) + wave * 64;
  const float* scr4f = reinterpret_cast<const float*>(scr4);
  const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_block;
  const int64_t tile1 = (tile0 + a.tiles_per_block < a.n_tiles) ? tile0 + a.tiles_per_block : a.n_tiles;
  const float kEps = 48.0f * 5.9604644775390625e-08f;

  for (int64_t tile = tile0; tile < tile1; ++tile) {
    const int64_t b0 = tile * TP;
    const int npts = (a.B - b0) < TP ? (int)(a.B - b0) : TP;
    __syncthreads();
    {
      const float4* g = a.tablea + (size_t)b0 * 128;
      const int total = npts * 128;
      for (int e = tid; e < total; e += NT) rowsA[e] = g[e];
      const float4 pad = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int e = total + tid; e < TP * 128; e += NT) rowsA[e] = pad;  // points past the end of the cloud (never valid)
      const double* gs = a.src + 3 * (size_t)b0;
      for (int e = tid; e < npts * 3; e += NT) spts[e] = gs[e];
      for (int e = tid; e < npts; e += NT) {
        const double* c0 = a.table + (size_t)(b0 + e) * K * 3;
        anch[3 * e] = c0[0]; anch[3 * e + 1] = c0[1]; anch[3 * e + 2] = c0[2];
        cmx[e] = a.cmax[b0 + e];
      }
    }
    __syncthreads();

    Pending pend;
    bool have = false;
    for (int pt = wb * BW + bs; pt < TP; pt += WB * BW) {
      const bool valid = pvalid && (pt < npts);
      const int ptl = valid ? pt : 0;
      const double s0 = spts[3 * ptl], s1 = spts[3 * ptl + 1], s2 = spts[3 * ptl + 2];
      const double T0 = (s0 * Rt[0] + s1 * Rt[1] + s2 * Rt[2]) + tt[0];   // SVNICP.cpp:62-64
      const double T1 = (s0 * Rt[3] + s1 * Rt[4] + s2 * Rt[5]) + tt[1];
      const double T2 = (s0 * Rt[6] + s1 * Rt[7] + s2 * Rt[8]) + tt[2];
      const float xf0 = (float)(T0 - anch[3 * ptl]), xf1 = (float)(T1 - anch[3 * ptl + 1]), xf2 = (float)(T2 - anch[3 * ptl + 2]);
      const float X = __builtin_fmaxf(__builtin_fabsf(xf0), __builtin_fmaxf(__builtin_fabsf(xf1), __builtin_fabsf(xf2)));
      const float C = cmx[ptl];
      const float E = kEps * (C + X) * (C + X);
      const float beta = __builtin_fmaf(xf0, xf0, __builtin_fmaf(xf1, xf1, xf2 * xf2)) + 4.0f * E;
      scr4[lane] = make_float4(-2.0f * xf0, -2.0f * xf1, -2.0f * xf2, 1.0f);
      scrb[lane] = beta;
      __builtin_amdgcn_wave_barrier();

      // operands: A rows of each distinct source point of this step, B column + accumulator input per column block
      constexpr int NPT = 4 / CBP;
      const int ptbase = pt - bs;  // wave-uniform
      float av[NPT][kRowBlocks];
#pragma unroll
      for (int q = 0; q < NPT; ++q) {
        int ptc = ptbase + q;
        ptc = ptc < TP ? ptc : TP - 1;
        const float4 alo = rowsA[(size_t)(ptc * 2) * 64 + lane];
        const float4 ahi = rowsA[(size_t)(ptc * 2 + 1) * 64 + lane];
        av[q][0] = alo.x; av[q][1] = alo.y; av[q][2] = alo.z; av[q][3] = alo.w;
        av[q][4] = ahi.x; av[q][5] = ahi.y; av[q][6] = ahi.z; av[q][7] = ahi.w;
      }
      float bvv[4], bee[4], b1[4], b2[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        bvv[cb] = scr4f[(16 * cb + mj) * 4 + mk];
        bee[cb] = scrb[16 * cb + mj];
        b1[cb] = __builtin_huge_valf(); b2[cb] = __builtin_huge_valf();
      }
      // software pipeline over the 4·NRB tiles: tile i+1 is issued to the matrix pipe before the VALU
      // consumes tile i, so the 12 tracking ops of one tile run under the 32 cycles of the next
      auto tile = [&](int i) -> v4f {
        const int cb = i / NRB, rb = i % NRB;
        const v4f cin = {bee[cb], bee[cb], bee[cb], bee[cb]};
        return __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb / CBP][rb], bvv[cb], cin, 0, 0, 0);
      };
      v4f dcur = tile(0);
#pragma unroll
      for (int i = 0; i < 4 * NRB; ++i) {
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

      const unsigned int wbits = __float_as_uint(b1own);
      int kb = (int)(((wbits & 0x1cu) << 2) | ((wbits >> 3) & 0xcu) | (wbits & 3u));  // 16·rb + 4·mk + v
      const float thr = 2.0f * E + 6.103515625e-05f * b2own + 1.0e-30f;
      const bool ambiguous = valid && (!(b2own - b1own > thr) || kb >= K);
      kb = kb < K ? kb : 0;
      const int64_t b = b0 + pt;
      const double* drow = a.table + (size_t)(valid ? b : b0) * K * 3;
      unsigned long long am = __ballot(ambiguous);
      if (am) {  // rare: exact f64 nearest-of-K for the undecided lanes, candidate-parallel across the wave
        if (a.ambig_count && lane == 0) atomicAdd(a.ambig_count, 1);
        do {
          const int L = (int)__builtin_ctzll(am);
          am &= am - 1;
          const double t0 = rdlane_f64(T0, L), t1 = rdlane_f64(T1, L), t2 = rdlane_f64(T2, L);
          const int ptL = __builtin_amdgcn_readlane(pt, L);
          const double* r = a.table + (size_t)(b0 + ptL) * K * 3;
          double bd = __builtin_huge_val(), d_first = 0.0;
          int bk = 0x7fffffff;
          for (int k = lane; k < K; k += kWave) {
            const double dx = t0 - r[3 * k], dy = t1 - r[3 * k + 1], dz = t2 - r[3 * k + 2];
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
      if (a.corr && valid) a.corr[(size_t)p * a.B + b] = kb;
      Pending cur;
      cur.T0 = T0; cur.T1 = T1; cur.T2 = T2; cur.pt = valid ? pt : -1;
      cur.q0 = drow[3 * kb]; cur.q1 = drow[3 * kb + 1]; cur.q2 = drow[3 * kb + 2];
      if (have && pend.pt >= 0) accumulate_point(pend, spts, a.max_dist, a.svgd, acc);
      pend = cur;
      have = true;
    }
    if (have && pend.pt >= 0) accumulate_point(pend, spts, a.max_dist, a.svgd, acc);  // drain before the tile is replaced
  }

#pragma unroll
  for (int off = PW; off < kWave; off <<= 1) {
#pragma unroll
    for (int i = 0; i < kNSums; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
  }
  if constexpr (WB > 1) {
    __syncthreads();
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

// candidate table for the MFMA kernel: f64 absolute coordinates [B][K][3] (target_batch of
// SVGDICP.cpp:191-193), C_b, and the float32 local rows in MFMA A-operand order:
// tablea[b][h][lane] = float4 over row blocks rb = 4h..4h+3 of component (lane/16) of candidate 16·rb + lane%16.
__global__ __launch_bounds__(256) void k_build_table3(const int32_t* __restrict__ idx, int64_t B, int K,
                                                      const double* __restrict__ tgt, int64_t M, double* __restrict__ table,
                                                      double* __restrict__ anchor, float4* __restrict__ tablea,
                                                      float4* __restrict__ tail, float* __restrict__ cmax) {
  __shared__ float rowbuf[4][128 * 4];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= B) return;  // whole wave
  int64_t i0 = idx[b * K];
  i0 = i0 < 0 ? 0 : (i0 >= M ? M - 1 : i0);  // clamped: a corrupted candidate list must never become a wild gather
  const double a0 = tgt[3 * i0], a1 = tgt[3 * i0 + 1], a2 = tgt[3 * i0 + 2];
  float cm = 0.0f;
  float* rb = rowbuf[wave];
  for (int k = lane; k < 128; k += kWave) {
    float cx = 0.f, cy = 0.f, cz = 0.f, cc = kSentinelCC;
    if (k < K) {
      int64_t i = idx[b * K + k];
      i = i < 0 ? 0 : (i >= M ? M - 1 : i);
      const double x = tgt[3 * i], y = tgt[3 * i + 1], z = tgt[3 * i + 2];
      if (table) {
        double* o = table + ((size_t)b * K + k) * 3;
        o[0] = x; o[1] = y; o[2] = z;
      }
      cx = (float)(x - a0); cy = (float)(y - a1); cz = (float)(z - a2);
      cc = (float)(((double)cx * cx + (double)cy * cy) + (double)cz * cz);
      // C_b = max |c'|_2 over the point's candidates, rounded up (>= max |c'|_inf, which is all the f32 kernels' bounds need;
      // the bf16 search kernel's bound is written in the 2-norm, stein_split.hip)
      cm = __builtin_fmaxf(cm, (float)sqrt(((double)cx * cx + (double)cy * cy) + (double)cz * cz) * 1.0000002f);
      if (!(cc < kSentinelCC)) cm = __builtin_nanf("");  // huge or NaN rows: every step of this point takes the exact path
    }
    rb[4 * k] = cx; rb[4 * k + 1] = cy; rb[4 * k + 2] = cz; rb[4 * k + 3] = cc;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(cm, off, kWave);
    cm = (cm != cm || o != o) ? __builtin_nanf("") : __builtin_fmaxf(cm, o);
  }
  if (lane == 0) {
    cmax[b] = cm;
    anchor[3 * b] = a0; anchor[3 * b + 1] = a1; anchor[3 * b + 2] = a2;
  }
  __builtin_amdgcn_wave_barrier();
  if (tail && lane < 4) tail[(size_t)b * 4 + lane] = make_float4(rb[4 * (96 + lane)], rb[4 * (96 + lane) + 1], rb[4 * (96 + lane) + 2], rb[4 * (96 + lane) + 3]);
  const int mi = lane & 15, mk = lane >> 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float4 v;
    v.x = rb[4 * (16 * (4 * h + 0) + mi) + mk];
    v.y = rb[4 * (16 * (4 * h + 1) + mi) + mk];
    v.z = rb[4 * (16 * (4 * h + 2) + mi) + mk];
    v.w = rb[4 * (16 * (4 * h + 3) + mi) + mk];
    tablea[((size_t)b * 2 + h) * 64 + lane] = v;
  }
}

template <int PW, int WP, int NRB>
hipError_t launch_m(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  auto kern = k_stein_accumulate_mfma<PW, WP, NRB>;
  if (plan.smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(plan.grid_x, plan.grid_y), dim3(NT), plan.smem, st, a);
  return hipGetLastError();
}

template <int PW, int WP, int NRB>
int occ_m(size_t smem) {
  int n = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_accumulate_mfma<PW, WP, NRB>, NT, smem);
  return (e != hipSuccess || n < 1) ? 1 : n;
}

// row-block counts with an instantiation: K <= 16, 32, 64, 112, 128
inline int row_blocks_for(int K) { return K <= 16 ? 1 : K <= 32 ? 2 : K <= 64 ? 4 : K <= 112 ? 7 : 8; }

template <int PW, int WP>
hipError_t launch_rb(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  switch (row_blocks_for(a.K)) {
    case 1: return launch_m<PW, WP, 1>(plan, a, st);
    case 2: return launch_m<PW, WP, 2>(plan, a, st);
    case 4: return launch_m<PW, WP, 4>(plan, a, st);
    case 7: return launch_m<PW, WP, 7>(plan, a, st);
    default: return launch_m<PW, WP, 8>(plan, a, st);
  }
}
template <int PW, int WP>
int occ_rb(int K, size_t smem) {
  switch (row_blocks_for(K)) {
    case 1: return occ_m<PW, WP, 1>(smem);
    case 2: return occ_m<PW, WP, 2>(smem);
    case 4: return occ_m<PW, WP, 4>(smem);
    case 7: return occ_m<PW, WP, 7>(smem);
    default: return occ_m<PW, WP, 8>(smem);
  }
}

}  // namespace

int mfma_occupancy_blocks(int PW, int WP, int K, size_t smem) {
  switch (PW) {
    case 16: return occ_rb<16, 1>(K, smem);
    case 32: return occ_rb<32, 1>(K, smem);
    default: return WP == 1 ? occ_rb<64, 1>(K, smem) : WP == 2 ? occ_rb<64, 2>(K, smem) : occ_rb<64, 4>(K, smem);
  }
}

hipError_t launch_accumulate_mfma(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  switch (plan.PW) {
    case 16: return launch_rb<16, 1>(plan, a, st);
    case 32: return launch_rb<32, 1>(plan, a, st);
    default:
      if (plan.WP == 1) return launch_rb<64, 1>(plan, a, st);
      if (plan.WP == 2) return launch_rb<64, 2>(plan, a, st);
      return launch_rb<64, 4>(plan, a, st);
  }
}

hipError_t launch_build_table3(const int32_t* idx, int64_t B, int K, const double* tgt, int64_t M, double* table,
                               double* anchor, float4* tablea, float4* tail, float* cmax, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_build_table3, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, idx, B, K, tgt, M, table, anchor, tablea,
                     tail, cmax);
  return hipGetLastError();
}

}  // namespace svnicp
