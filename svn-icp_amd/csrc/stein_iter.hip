// stein_iter.hip — Stage B: one fused pass per iteration over (particle, source point):
//   transform → nearest-of-K candidates → point_filter mask → robust weight → Gauss–Newton sums.
//
// Replaces, per iteration, the reference's ~60 ATen launches in SVNICP::stein_align
// (src/core/SVNICP.cpp:58-71): pose compose + bmm transform (:58-64), get_correspondence_fast
// (src/core/SVGDICP.cpp:300-329 → KNearestNeighborKernelV3, src/core/knn/knn.cu:204-251, K=1),
// point_filter ×3 (SVGDICP.cpp:331-333) and the accumulation half of Newton_grad_right
// (SVNICP.cpp:116-157).  Nothing of size [P,B,…] is ever materialised.
//
// Result contract (oracle/svnicp_oracle.c correspond()/newton_accumulate()):
//  * Ts_i = ((s0·Rt[i][0] + s1·Rt[i][1]) + s2·Rt[i][2]) + tt_i, unfused; d² = ((dx·dx)+dy·dy)+dz·dz
//    unfused; argmin over k with strict '<' (first k wins); mask = d² < max_dist (the reference
//    compares the SQUARED distance with the un-squared max_dist, SVGDICP.cpp:332).
//  * masked-out rows are zeroed, not dropped (SVGDICP.cpp:331-333): they still add w = 1 with
//    s = 0, e = 0, i.e. +1 to Σw only.
//  * H, b are not accumulated entry by entry.  With J = [Rc | −Rc·ŝ] (SVNICP.cpp:145-146),
//    JᵀwJ = [[wI, −wŝ],[wŝ, w(‖s‖²I − ssᵀ)]] does not depend on Rc, and
//    Jᵀ(we) = [Rcᵀ(we) ; s × Rcᵀ(we)], so 22 raw sums per particle suffice:
//      [0] Σw  [1..3] Σw·s  [4..9] Σw·ssᵀ (xx,xy,xz,yy,yz,zz)  [10..12] Σwe  [13..21] Σ(we)_i s_j
//    (checked against the oracle's literal J accumulation in tests/).  These sums are also the
//    record that is all-gathered between GPUs.
//
// MI355X mapping: lane ↔ particle (PW = 8…64 lanes), the remaining 64/PW lane groups and the
// workgroup's waves take different source points; the K candidate rows of a point are read from
// LDS with one address per lane group (broadcast), rows padded to an odd stride so that distinct
// rows never share a bank.  Each lane keeps its 22 accumulators in VGPRs for the whole launch;
// cross-lane/wave/block reduction happens once at the end, in a fixed order (deterministic).
#include "kernels.hpp"

namespace svnicp {

namespace {

constexpr int NT = 256;

template <int PW, int WP>
__global__ __launch_bounds__(NT) void k_stein_accumulate(AccumArgs a) {
  if (a.ctl[0]) return;  // early stop already signalled (SVNICP.cpp:95-101)
  constexpr int BW = kWave / PW;  // source points per wave pass
  constexpr int WB = 4 / WP;      // waves along the source-point axis
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const int wp = wave % WP, wb = wave / WP;
  const int pl = lane % PW, bs = lane / PW;
  const int pidx = blockIdx.y * (WP * PW) + wp * PW + pl;  // index inside the shard (padded)
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

  const int K = a.K, RS = a.RS, TP = a.TP;
  const int K3 = 3 * K;
  double* rows = lds;                        // [TP][RS]
  double* spts = lds + (size_t)TP * RS;      // [TP][3]
  const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_block;
  const int64_t tile1 = (tile0 + a.tiles_per_block < a.n_tiles) ? tile0 + a.tiles_per_block : a.n_tiles;

  for (int64_t tile = tile0; tile < tile1; ++tile) {
    const int64_t b0 = tile * TP;
    const int npts = (a.B - b0) < TP ? (int)(a.B - b0) : TP;
    __syncthreads();  // previous tile fully consumed
    {                 // stage candidate rows: contiguous npts*3K doubles → padded rows
      const double* g = a.table + (size_t)b0 * K3;
      const int total = npts * K3;
      int row = tid / K3, col = tid - row * K3;
      for (int e = tid; e < total; e += NT) {
        rows[row * RS + col] = g[e];
        col += NT;
        while (col >= K3) { col -= K3; ++row; }
      }
      const double* gs = a.src + 3 * (size_t)b0;
      for (int e = tid; e < npts * 3; e += NT) spts[e] = gs[e];
    }
    __syncthreads();

    for (int pt = wb * BW + bs; pt < TP; pt += WB * BW) {
      const bool valid = pvalid && (pt < npts);
      const double s0 = spts[3 * pt], s1 = spts[3 * pt + 1], s2 = spts[3 * pt + 2];
      // SVNICP.cpp:62-64
      const double T0 = (s0 * Rt[0] + s1 * Rt[1] + s2 * Rt[2]) + tt[0];
      const double T1 = (s0 * Rt[3] + s1 * Rt[4] + s2 * Rt[5]) + tt[1];
      const double T2 = (s0 * Rt[6] + s1 * Rt[7] + s2 * Rt[8]) + tt[2];
      const double* row = rows + pt * RS;
      double best;
      int kb = 0;
      {
        const double dx = T0 - row[0], dy = T1 - row[1], dz = T2 - row[2];
        best = (dx * dx + dy * dy) + dz * dz;
      }
#pragma unroll 4
      for (int k = 1; k < K; ++k) {
        const double dx = T0 - row[3 * k], dy = T1 - row[3 * k + 1], dz = T2 - row[3 * k + 2];
        const double d = (dx * dx + dy * dy) + dz * dz;
        const bool lt = d < best;  // strict: first k wins ties (knn_cpu.cpp:52 with K = 1)
        best = lt ? d : best;
        kb = lt ? k : kb;
      }
      if (valid) {
        if (a.corr) a.corr[(size_t)p * a.B + (b0 + pt)] = kb;
        double w = 1.0, e0 = 0.0, e1 = 0.0, e2 = 0.0, m0 = 0.0, m1 = 0.0, m2 = 0.0;
        if (best < a.max_dist) {  // point_filter, SVGDICP.cpp:331-333
          const double n = sqrt(best);                      // ‖Ts − q‖, SVNICP.cpp:120
          const double wq = a.max_dist / (a.max_dist + 3 * n);
          w = wq * wq;                                      // SVNICP.cpp:122
          e0 = w * (T0 - row[3 * kb]);                      // SVNICP.cpp:119,123
          e1 = w * (T1 - row[3 * kb + 1]);
          e2 = w * (T2 - row[3 * kb + 2]);
          m0 = s0; m1 = s1; m2 = s2;
        }
        const double w0 = w * m0, w1 = w * m1, w2 = w * m2;
        acc[0] += w;
        acc[1] += w0; acc[2] += w1; acc[3] += w2;
        acc[4] = fma(w0, m0, acc[4]); acc[5] = fma(w0, m1, acc[5]); acc[6] = fma(w0, m2, acc[6]);
        acc[7] = fma(w1, m1, acc[7]); acc[8] = fma(w1, m2, acc[8]); acc[9] = fma(w2, m2, acc[9]);
        acc[10] += e0; acc[11] += e1; acc[12] += e2;
        acc[13] = fma(e0, m0, acc[13]); acc[14] = fma(e0, m1, acc[14]); acc[15] = fma(e0, m2, acc[15]);
        acc[16] = fma(e1, m0, acc[16]); acc[17] = fma(e1, m1, acc[17]); acc[18] = fma(e1, m2, acc[18]);
        acc[19] = fma(e2, m0, acc[19]); acc[20] = fma(e2, m1, acc[20]); acc[21] = fma(e2, m2, acc[21]);
      }
    }
  }

  // ---- fixed-order reduction: lane groups → waves → one partial row per particle ----
#pragma unroll
  for (int off = PW; off < kWave; off <<= 1) {
#pragma unroll
    for (int i = 0; i < kNSums; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
  }
  if constexpr (WB > 1) {
    __syncthreads();  // tile memory is free now
    double* red = lds;  // [(WB-1)][WP*PW][kNSums]
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

// sums[p_lo + i][s] = Σ_blk partial[blk][i][s], block order fixed.  Workgroup = 16 entries × 16
// block lanes; each block lane walks blk = bl, bl+16, … and the 16 lanes are folded in order.
__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partial, int nblk, int Ppad,
                                                          int p_lo, int n_particles, double* __restrict__ sums,
                                                          const int* __restrict__ ctl) {
  if (ctl[0]) return;
  __shared__ double red[16][17];
  const int el = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int entry = blockIdx.x * 16 + el;  // index into [n_particles][kNSums]
  const int n_entries = n_particles * kNSums;
  double a = 0.0;
  if (entry < n_entries) {
    const size_t stride = (size_t)Ppad * kNSums;
    for (int blk = bl; blk < nblk; blk += 16) a += partial[(size_t)blk * stride + entry];
  }
  red[bl][el] = a;
  __syncthreads();
  if (bl == 0 && entry < n_entries) {
    double s = red[0][el];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += red[i][el];
    sums[(size_t)p_lo * kNSums + entry] = s;
  }
}

template <int PW, int WP>
hipError_t launch_t(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  auto kern = k_stein_accumulate<PW, WP>;
  if (plan.smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(plan.grid_x, plan.grid_y), dim3(NT), plan.smem, st, a);
  return hipGetLastError();
}

}  // namespace

AccumPlan plan_accumulate(int n_particles, int64_t B, int K, int num_cus) {
  AccumPlan pl{};
  int PW = 8;
  while (PW < 64 && PW < n_particles) PW <<= 1;
  int WP = 1;
  if (PW == 64) { WP = (n_particles + 63) / 64; if (WP >= 3) WP = 4; }
  pl.PW = PW; pl.WP = WP;
  const int per_wg = PW * WP;
  pl.grid_y = (n_particles + per_wg - 1) / per_wg;
  pl.Ppad = pl.grid_y * per_wg;
  const int BW = 64 / PW, WB = 4 / WP;
  const int pass = BW * WB;                  // points per workgroup pass
  pl.RS = (3 * K) | 1;
  int TP = pass;
  while (TP < 16) TP += pass;                // at least 16 points per tile …
  while (TP > pass && (size_t)TP * (pl.RS + 3) * 8 > 60 * 1024) TP -= pass;  // … within ~60 KB
  pl.TP = TP;
  const size_t tile_bytes = (size_t)TP * (pl.RS + 3) * sizeof(double);
  const size_t red_bytes = (size_t)(WB - 1) * per_wg * kNSums * sizeof(double);
  pl.smem = tile_bytes > red_bytes ? tile_bytes : red_bytes;
  pl.n_tiles = (B + TP - 1) / TP;
  int64_t want = (int64_t)num_cus * 4 / (pl.grid_y > 0 ? pl.grid_y : 1);  // ≈4 workgroups per CU in total
  if (want < 1) want = 1;
  int64_t gx = pl.n_tiles < want ? pl.n_tiles : want;
  if (gx < 1) gx = 1;
  pl.tiles_per_block = (int)((pl.n_tiles + gx - 1) / gx);
  if (pl.tiles_per_block < 1) pl.tiles_per_block = 1;
  pl.grid_x = (int)((pl.n_tiles + pl.tiles_per_block - 1) / pl.tiles_per_block);
  if (pl.grid_x < 1) pl.grid_x = 1;
  return pl;
}

hipError_t launch_accumulate(const AccumPlan& plan, AccumArgs a, hipStream_t st) {
  a.TP = plan.TP; a.RS = plan.RS; a.tiles_per_block = plan.tiles_per_block; a.n_tiles = plan.n_tiles;
  a.Ppad = plan.Ppad;
  switch (plan.PW) {
    case 8: return launch_t<8, 1>(plan, a, st);
    case 16: return launch_t<16, 1>(plan, a, st);
    case 32: return launch_t<32, 1>(plan, a, st);
    default:
      if (plan.WP == 1) return launch_t<64, 1>(plan, a, st);
      if (plan.WP == 2) return launch_t<64, 2>(plan, a, st);
      return launch_t<64, 4>(plan, a, st);
  }
}

hipError_t launch_reduce_partials(const double* partial, int nblk, int Ppad, int p_lo, int n_particles,
                                  double* sums, const int* ctl, hipStream_t st) {
  const int n_entries = n_particles * kNSums;
  if (n_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_reduce_partials, dim3((n_entries + 15) / 16), dim3(256), 0, st, partial, nblk, Ppad, p_lo,
                     n_particles, sums, ctl);
  return hipGetLastError();
}

}  // namespace svnicp
