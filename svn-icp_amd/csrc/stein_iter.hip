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
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"
#include "stein_common.hpp"
#include "update_single.hpp"

namespace svnicp {

namespace {

// ONE particle and nothing between the sums and the update (no exchange between ranks): the workgroup that finishes last —
// one __threadfence and one atomic ticket per workgroup, nobody waits — adds the workgroups' partial sums in block order
// and runs the whole Stein step (update_single.hpp: for P = 1 it is the Newton step and the pose update, a few
// microseconds on one thread).  An iteration of a plain-ICP registration is then ONE launch instead of three
// (accumulate, k_reduce_partials, k_particle_update).  Block-wide call; `lds` = the kernel's dynamic LDS (free by now).
__device__ inline void single_particle_tail(const AccumArgs& a, const UpdateArgs& u, double* lds) {
  __shared__ int sh_last;
  const int tid = threadIdx.x;
  __threadfence();                       // this workgroup's partial sums are device-visible before its ticket
  __syncthreads();
  if (tid == 0) {
    const unsigned int t = atomicAdd(a.ticket, 1u);
    sh_last = t == gridDim.x * gridDim.y - 1;
    if (sh_last) *a.ticket = 0u;         // ready for the next iteration (stream order)
  }
  __syncthreads();
  if (!sh_last) return;
  __threadfence();                       // the other workgroups' partial sums
  constexpr int RL = 11;                 // 11 block lanes x 22 entries = 242 of the 256 threads
  double (*red)[kNSums + 1] = reinterpret_cast<double (*)[kNSums + 1]>(lds);
  const int es = tid % kNSums, bl = tid / kNSums;
  const int nblk = (int)gridDim.x;
  if (bl < RL) {
    const size_t stride = (size_t)a.Ppad * kNSums;
    const double* src = a.partial + es;  // particle lane 0 of the shard
    double v = 0.0;
    for (int blk = bl; blk < nblk; blk += RL) v += src[(size_t)blk * stride];
    red[bl][es] = v;
  }
  __syncthreads();
  double* sums = lds + RL * (kNSums + 1);
  if (tid < kNSums) {
    double v = red[0][tid];
#pragma unroll
    for (int k = 1; k < RL; ++k) v += red[k][tid];
    sums[tid] = v;
    const_cast<double*>(u.sums)[tid] = v;   // svnicp_sums_devptr stays meaningful
  }
  __syncthreads();
  if (tid == 0) update_single_particle(u, sums);
}

// ONE particle: the whole iteration in one launch, lanes along the SOURCE POINTS.  The stage-B kernels put particles on
// the lanes (lane <-> particle): with one particle 63 of 64 lanes idle and an iteration of BASELINE C1 (4 096 points, K = 100)
// took 34 us.  Here four lanes share a source point and a quarter of its K candidates each (float64 rows of `table`, the
// reference's arithmetic and first-index tie rule: strict '<' in ascending k inside a lane, (d², k) order across the four),
// the first of the four accumulates the 22 sums of its point, the wave and the workgroup fold them in a fixed order, and the
// last workgroup to finish reduces the workgroups' partial sums and runs the Stein step (single_particle_tail).
__global__ __launch_bounds__(NT) void k_icp_single(AccumArgs a, UpdateArgs u) {
  if (a.ctl[0]) return;  // early stop already signalled (SVNICP.cpp:95-101)
  __shared__ double s_red[4][kNSums];
  __shared__ double s_tail[12 * (kNSums + 1) + kNSums];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int sub = lane & 3, ptl = lane >> 2;
  const int64_t b = (int64_t)blockIdx.x * 64 + wave * 16 + ptl;
  const bool inb = b < a.B;
  const int64_t bl = inb ? b : a.B - 1;
  const int K = a.K;
  const double* rp = a.Rtot + 12 * (size_t)a.p_lo;
  const double* sp = a.src + 3 * bl;
  const double s0 = sp[0], s1 = sp[1], s2 = sp[2];
  const double T0 = (s0 * rp[0] + s1 * rp[1] + s2 * rp[2]) + rp[9];    // SVNICP.cpp:62-64
  const double T1 = (s0 * rp[3] + s1 * rp[4] + s2 * rp[5]) + rp[10];
  const double T2 = (s0 * rp[6] + s1 * rp[7] + s2 * rp[8]) + rp[11];
  const double* row = a.table + (size_t)bl * K * 3;
  double bd = __builtin_huge_val(), d_first = 0.0;
  int bk = 0x7fffffff;
  constexpr int U = 5;   // candidates per lane and trip: their loads go out together
  for (int k0 = sub; k0 < K; k0 += 4 * U) {
    double x[U], y[U], z[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int k = k0 + 4 * i;
      const double* r = row + 3 * (k < K ? k : 0);
      x[i] = r[0]; y[i] = r[1]; z[i] = r[2];
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int k = k0 + 4 * i;
      const double dx = T0 - x[i], dy = T1 - y[i], dz = T2 - z[i];
      const double d = (dx * dx + dy * dy) + dz * dz;   // knn_cpu.cpp:43-50 order, unfused
      if (k == 0) d_first = d;
      if (k < K && d < bd) { bd = d; bk = k; }          // strict: the first k wins ties (knn_cpu.cpp:52 with K = 1)
    }
  }
#pragma unroll
  for (int off = 1; off < 4; off <<= 1) {
    const double od = __shfl_xor(bd, off, kWave);
    const int ok = __shfl_xor(bk, off, kWave);
    if (od < bd || (od == bd && ok < bk)) { bd = od; bk = ok; }
  }
  const double d0 = __shfl(d_first, lane & ~3, kWave);
  // the serial reference loop starts from candidate 0 and only replaces on '<': a NaN first distance is never replaced
  const int kb = (d0 != d0 || bk == 0x7fffffff) ? 0 : bk;

  double acc[kNSums];
#pragma unroll
  for (int i = 0; i < kNSums; ++i) acc[i] = 0.0;
  if (sub == 0 && inb) {
    if (a.corr) a.corr[(size_t)a.p_lo * a.B + b] = kb;
    const double* r = row + 3 * kb;
    const double dx = T0 - r[0], dy = T1 - r[1], dz = T2 - r[2];
    const double best = (dx * dx + dy * dy) + dz * dz;
    double w = 1.0, e0 = 0.0, e1 = 0.0, e2 = 0.0, m0 = 0.0, m1 = 0.0, m2 = 0.0;
    if (best < a.max_dist) {  // point_filter, SVGDICP.cpp:331-333
      const double n = sqrt(best);                      // ‖Ts − q‖, SVNICP.cpp:120
      const double wq = a.max_dist / (a.max_dist + 3 * n);
      w = wq * wq;                                      // SVNICP.cpp:122
      e0 = w * dx; e1 = w * dy; e2 = w * dz;            // SVNICP.cpp:119,123
      m0 = s0; m1 = s1; m2 = s2;
    } else if (best != best) {
      // the reference masks by MULTIPLYING the rows with 0 / 1 (SVGDICP.cpp:331-333): a non-finite row (a particle whose pose
      // went NaN) stays NaN, and so does every sum it enters — k_stein_accumulate_w forms the same products
      w = best; e0 = best; e1 = best; e2 = best;
    }
    const double w0 = w * m0, w1 = w * m1, w2 = w * m2;
    acc[0] = w;
    acc[1] = w0; acc[2] = w1; acc[3] = w2;
    // SVGD mode needs count_nonzero(mask·Ts summed over xyz) (SVGDICP.cpp:404) instead of Σw·s_x²
    acc[4] = a.svgd ? ((best < a.max_dist && ((T0 + T1) + T2) != 0.0) ? 1.0 : 0.0) : w0 * m0;
    acc[5] = w0 * m1; acc[6] = w0 * m2;
    acc[7] = w1 * m1; acc[8] = w1 * m2; acc[9] = w2 * m2;
    acc[10] = e0; acc[11] = e1; acc[12] = e2;
    acc[13] = e0 * m0; acc[14] = e0 * m1; acc[15] = e0 * m2;
    acc[16] = e1 * m0; acc[17] = e1 * m1; acc[18] = e1 * m2;
    acc[19] = e2 * m0; acc[20] = e2 * m1; acc[21] = e2 * m2;
  }
  // fixed-order reduction: the wave's 16 points, the workgroup's four waves, then (tail) the workgroups in block order
#pragma unroll
  for (int off = 4; off < kWave; off <<= 1) {
#pragma unroll
    for (int i = 0; i < kNSums; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < kNSums; ++i) s_red[wave][i] = acc[i];
  }
  __syncthreads();
  if (tid < kNSums) a.partial[((size_t)blockIdx.x * a.Ppad) * kNSums + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
  single_particle_tail(a, u, s_tail);
}

template <int PW, int WP>
__global__ __launch_bounds__(NT) void k_stein_accumulate(AccumArgs a, UpdateArgs u, int fuse_single) {
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
        } else if (best != best) {   // masking is a multiplication in the reference: a NaN row stays NaN (see k_icp_single)
          w = best; e0 = best; e1 = best; e2 = best;
        }
        const double w0 = w * m0, w1 = w * m1, w2 = w * m2;
        acc[0] += w;
        acc[1] += w0; acc[2] += w1; acc[3] += w2;
        // SVGD mode needs count_nonzero(mask·Ts summed over xyz) (SVGDICP.cpp:404) instead of Σw·s_x²
        acc[4] = a.svgd ? acc[4] + ((best < a.max_dist && ((T0 + T1) + T2) != 0.0) ? 1.0 : 0.0) : fma(w0, m0, acc[4]);
        acc[5] = fma(w0, m1, acc[5]); acc[6] = fma(w0, m2, acc[6]);
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

// ---------------------------------------------------------------------------------------------
// Fast variant: float32 search in local coordinates, exact float64 finish.
//
// For source point b let a_b = its first candidate.  k_build_table2 stores every candidate as
// c' = fl32(c − a_b) and cc = fl32(|c'|²) (16 B), plus C_b = max |c'|∞.  For a particle with
// x' = fl32(T_p(s_b) − a_b) the score  S_k = fma(c'z,mz, fma(c'y,my, fma(c'x,mx, cc))),  m = −2x',
// approximates d²_k − |x'|², so argmin_k S_k is the nearest candidate unless two scores are closer
// than the rounding error.  Bound (u = 2^-24, C = C_b, X = |x'|∞):
//   |cc − |c'|²_real| <= 10uC²,  cross-term inputs 12.6uXC,  three fma roundings 3u(3C²+6XC)
//   => |S_k − s_k| <= 19.1uC² + 30.8uXC  <  EPS := 40·u·C·(C+X).
// A lane keeps (min, argmin, second-min) of S; if second-min − min > 2·EPS its argmin is the
// exact f64 argmin (every other candidate is strictly farther in exact arithmetic, so the
// reference's first-index tie rule cannot matter).  Otherwise — exact ties, padded duplicates,
// overflow/NaN — the wave step re-runs the exact f64 loop of the baseline kernel on the f64 table.
// The winner's d², mask, weight and the 22 sums are computed in f64 exactly as in the baseline
// kernel, so correspondences and sums are bit-identical to it.
// ---------------------------------------------------------------------------------------------
template <int PW, int WP>
__global__ __launch_bounds__(NT, 4) void k_stein_accumulate_f32(AccumArgs a, UpdateArgs u, int fuse_single) {
  if (a.ctl[0]) return;
  constexpr int BW = kWave / PW;
  constexpr int WB = 4 / WP;
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
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

  const int K = a.K, TP = a.TP;
  const int Kp = (K + 3) & ~3;  // LDS row length: padded to a multiple of 4 with never-selected sentinels
  // LDS: rows [TP][Kp] float4 (+4 slack for the prefetch) | spts [TP][3] f64 | anch [TP][3] f64 | cmax [TP] f32
  float4* rows = reinterpret_cast<float4*>(lds);
  double* spts = reinterpret_cast<double*>(rows + (size_t)TP * Kp + 4);
  double* anch = spts + 3 * TP;
  float* cmx = reinterpret_cast<float*>(anch + 3 * TP);
  const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_block;
  const int64_t tile1 = (tile0 + a.tiles_per_block < a.n_tiles) ? tile0 + a.tiles_per_block : a.n_tiles;
  const float kEpsScale = 40.0f * 5.9604644775390625e-08f;

  for (int64_t tile = tile0; tile < tile1; ++tile) {
    const int64_t b0 = tile * TP;
    const int npts = (a.B - b0) < TP ? (int)(a.B - b0) : TP;
    __syncthreads();
    {
      const float4* g = a.tablef + (size_t)b0 * K;
      const int total = TP * Kp + 4;
      const float4 sentinel = make_float4(0.f, 0.f, 0.f, __builtin_huge_valf());
      int r = tid / Kp, c = tid - r * Kp;
      for (int e = tid; e < total; e += NT) {
        rows[e] = (r < npts && c < K) ? g[(size_t)r * K + c] : sentinel;
        c += NT;
        while (c >= Kp) { c -= Kp; ++r; }
      }
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
      const double s0 = spts[3 * pt], s1 = spts[3 * pt + 1], s2 = spts[3 * pt + 2];
      const double T0 = (s0 * Rt[0] + s1 * Rt[1] + s2 * Rt[2]) + tt[0];   // SVNICP.cpp:62-64
      const double T1 = (s0 * Rt[3] + s1 * Rt[4] + s2 * Rt[5]) + tt[1];
      const double T2 = (s0 * Rt[6] + s1 * Rt[7] + s2 * Rt[8]) + tt[2];
      const float xf0 = (float)(T0 - anch[3 * pt]), xf1 = (float)(T1 - anch[3 * pt + 1]), xf2 = (float)(T2 - anch[3 * pt + 2]);
      const float m0 = -2.0f * xf0, m1 = -2.0f * xf1, m2 = -2.0f * xf2;
      const float4* row = rows + (size_t)pt * Kp;
      float best1 = __builtin_huge_valf(), best2 = __builtin_huge_valf();
      int kb = 0;
      float4 c0 = row[0], c1 = row[1], c2 = row[2], c3 = row[3];
      for (int k = 0; k < Kp; k += 4) {
        const float4 n0 = row[k + 4], n1 = row[k + 5], n2 = row[k + 6], n3 = row[k + 7];  // prefetch (slack at the end)
        const float sA = __builtin_fmaf(c0.z, m2, __builtin_fmaf(c0.y, m1, __builtin_fmaf(c0.x, m0, c0.w)));
        const float sB = __builtin_fmaf(c1.z, m2, __builtin_fmaf(c1.y, m1, __builtin_fmaf(c1.x, m0, c1.w)));
        const float sC = __builtin_fmaf(c2.z, m2, __builtin_fmaf(c2.y, m1, __builtin_fmaf(c2.x, m0, c2.w)));
        const float sD = __builtin_fmaf(c3.z, m2, __builtin_fmaf(c3.y, m1, __builtin_fmaf(c3.x, m0, c3.w)));
        bool lt;
        lt = sA < best1; best2 = __builtin_amdgcn_fmed3f(best1, best2, sA); best1 = __builtin_fminf(best1, sA); kb = lt ? k : kb;
        lt = sB < best1; best2 = __builtin_amdgcn_fmed3f(best1, best2, sB); best1 = __builtin_fminf(best1, sB); kb = lt ? k + 1 : kb;
        lt = sC < best1; best2 = __builtin_amdgcn_fmed3f(best1, best2, sC); best1 = __builtin_fminf(best1, sC); kb = lt ? k + 2 : kb;
        lt = sD < best1; best2 = __builtin_amdgcn_fmed3f(best1, best2, sD); best1 = __builtin_fminf(best1, sD); kb = lt ? k + 3 : kb;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      }
      const float X = __builtin_fmaxf(__builtin_fabsf(xf0), __builtin_fmaxf(__builtin_fabsf(xf1), __builtin_fabsf(xf2)));
      const float C = cmx[pt];
      const float eps2 = 2.0f * kEpsScale * C * (C + X);
      const bool ambiguous = valid && !(best2 - best1 > eps2);
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
      // software pipeline: start the gather of this point's winner, then finish the previous point
      // (whose winner was requested one search loop ago)
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
  if (fuse_single) single_particle_tail(a, u, lds);
}

// candidate table: f64 absolute coordinates (target_batch = index_select(target, sourceKNN_idx),
// SVGDICP.cpp:191-193, ONE copy), float32 local coordinates + |c'|², and C_b.  One wave per row.
__global__ __launch_bounds__(256) void k_build_table2(const int32_t* __restrict__ idx, int64_t B, int K,
                                                      const double* __restrict__ tgt, int64_t M, double* __restrict__ table,
                                                      float4* __restrict__ tablef, float* __restrict__ cmax) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  // indices are clamped into [0, M): a corrupted candidate list must never become a wild gather
  int64_t i0 = idx[b * K];
  i0 = i0 < 0 ? 0 : (i0 >= M ? M - 1 : i0);
  const double a0 = tgt[3 * i0], a1 = tgt[3 * i0 + 1], a2 = tgt[3 * i0 + 2];
  float cm = 0.0f;
  for (int k = lane; k < K; k += kWave) {
    int64_t i = idx[b * K + k];
    i = i < 0 ? 0 : (i >= M ? M - 1 : i);
    const double x = tgt[3 * i], y = tgt[3 * i + 1], z = tgt[3 * i + 2];
    double* o = table + ((size_t)b * K + k) * 3;
    o[0] = x; o[1] = y; o[2] = z;
    const float cx = (float)(x - a0), cy = (float)(y - a1), cz = (float)(z - a2);
    const float cc = (float)(((double)cx * cx + (double)cy * cy) + (double)cz * cz);
    tablef[(size_t)b * K + k] = make_float4(cx, cy, cz, cc);
    cm = __builtin_fmaxf(cm, __builtin_fmaxf(__builtin_fabsf(cx), __builtin_fmaxf(__builtin_fabsf(cy), __builtin_fabsf(cz))));
  }
  for (int off = 32; off > 0; off >>= 1) cm = __builtin_fmaxf(cm, __shfl_xor(cm, off, kWave));
  if (lane == 0) cmax[b] = cm;   // NaN coordinates give NaN scores => every step takes the exact path
}

// (k_reduce_partials lives in particle_update.hip: its last workgroup goes on with the sums-dependent half of the Stein step)

template <int PW, int WP>
hipError_t launch_t(const AccumPlan& plan, const AccumArgs& a, const UpdateArgs* single, hipStream_t st) {
  auto kern = plan.f32 ? k_stein_accumulate_f32<PW, WP> : k_stein_accumulate<PW, WP>;
  if (plan.smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.smem);
    if (e != hipSuccess) return e;
  }
  UpdateArgs u{};
  if (single) u = *single;
  hipLaunchKernelGGL(kern, dim3(plan.grid_x, plan.grid_y), dim3(NT), plan.smem, st, a, u, (single && plan.f32 == 1) ? 1 : 0);
  return hipGetLastError();
}

}  // namespace

template <int PW, int WP>
static int occ_t(const AccumPlan& pl) {
  int n = 0;
  hipError_t e = pl.f32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_accumulate_f32<PW, WP>, NT, pl.smem)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_accumulate<PW, WP>, NT, pl.smem);
  if (e != hipSuccess || n < 1) n = 2;
  return n > 8 ? 8 : n;
}
static int occupancy_blocks(const AccumPlan& pl) {
  switch (pl.PW) {
    case 8: return occ_t<8, 1>(pl);
    case 16: return occ_t<16, 1>(pl);
    case 32: return occ_t<32, 1>(pl);
    default: return pl.WP == 1 ? occ_t<64, 1>(pl) : pl.WP == 2 ? occ_t<64, 2>(pl) : occ_t<64, 4>(pl);
  }
}

AccumPlan plan_accumulate(int n_particles, int64_t B, int K, int num_cus, int f32, const Tuning& tune) {
  AccumPlan pl{};
  // MFMA tiles: 16 particles wide, 128 candidate rows; correspondence = full keeps the split accumulate kernel for any particle count
  if (f32 == 3 && (K > 128 || (n_particles <= 8 && !tune.full_corr))) f32 = 1;
  pl.f32 = f32;
  int PW = f32 == 3 ? 16 : 8;
  while (PW < 64 && PW < n_particles) PW <<= 1;
  int WP = 1;
  if (PW == 64) { WP = (n_particles + 63) / 64; if (WP >= 3) WP = 4; }
  pl.PW = PW; pl.WP = WP; pl.K = K;
  const int per_wg = PW * WP;
  pl.grid_y = (n_particles + per_wg - 1) / per_wg;
  pl.Ppad = pl.grid_y * per_wg;
  const int BW = 64 / PW, WB = 4 / WP;
  const int pass = BW * WB;                  // points per workgroup pass
  pl.RS = (3 * K) | 1;
  if (f32 == 3) {  // split variant: no LDS tiles; each kernel gets one resident round of workgroups (search: two)
    const int64_t steps = (B + pass - 1) / pass;
    auto size_grid = [&](int wg_per_cu, int* gx, int* ppb, int min_steps) {
      int64_t want = (int64_t)num_cus * wg_per_cu / (pl.grid_y > 0 ? pl.grid_y : 1);
      if (want < 1) want = 1;
      int64_t g = steps < want ? steps : want;
      // small clouds (the scan-to-map loop hands over ~1000 source points): a workgroup gets at least min_steps wave steps —
      // each accumulate workgroup writes 22 sums per particle that k_reduce_partials must read back
      if (min_steps > 1 && steps / g < min_steps) { g = steps / min_steps; if (g < 1) g = 1; }
      const int64_t spb = (steps + g - 1) / g;     // wave steps per workgroup
      *ppb = (int)(spb * pass);
      *gx = (int)((B + *ppb - 1) / *ppb);
    };
    pl.smem = (size_t)(WB - 1) * per_wg * kNSums * sizeof(double);
    int occ_s = 4, occ_a = 3;
    split_occupancy_blocks(PW, WP, K, pl.smem, &occ_s, &occ_a);
    if (tune.wgpcu_search >= 1 && tune.wgpcu_search <= 64 && tune.wgpcu_accum >= 1 && tune.wgpcu_accum <= 16) {
      occ_s = tune.wgpcu_search; occ_a = tune.wgpcu_accum;   // profiling knob
    } else {  // measured at C3: the barrier-free search kernel balances best with about four rounds of smaller workgroups
      if (PW == 64) occ_s *= 4;          // (round 3, tile tracking: 4.03 ms per registration at two rounds, 3.91 at four, 4.10 at six and eight);
                                         // with fewer than 64 particles per group — C2 — ONE round: 1.03 ms against 1.18 at two and
                                         // 1.31 at four (every workgroup pays its pose prologue and its exact-pass epilogue);
                                         // the accumulate kernel pays per workgroup in k_reduce_partials: 4 per CU
      if (occ_a > 4) occ_a = 4;
    }
    size_grid(occ_a, &pl.grid_x, &pl.pts_per_block, tune.accum_min_steps > 0 ? tune.accum_min_steps : 4);
    // few pairs (the scan-to-map loop's sizes): at most kSmallChainBlocks accumulate workgroups, so that the update kernels can
    // add their records themselves and k_reduce_partials is not launched (api.hip: small_chain)
    pl.small = (pl.grid_y == 1 && (int64_t)B * n_particles <= (1 << 19) && tune.small_chain) ? 1 : 0;
    if (pl.small && pl.grid_x > kSmallChainBlocks) {
      const int64_t spb = (steps + kSmallChainBlocks - 1) / kSmallChainBlocks;
      pl.pts_per_block = (int)(spb * pass);
      pl.grid_x = (int)((B + pl.pts_per_block - 1) / pl.pts_per_block);
    }
    size_grid(occ_s, &pl.sgrid_x, &pl.spts_per_block, 1);
    pl.TP = pass; pl.n_tiles = 0; pl.tiles_per_block = 0;
    return pl;
  }
  // bytes per staged source point: baseline = padded f64 row + point; f32 variant = K float4 + point + anchor + C_b
  const size_t per_pt = f32 ? ((size_t)((K + 3) & ~3) * 16 + 24 + 24 + 4) : ((size_t)(pl.RS + 3) * 8);
  int TP = pass;
  while (TP < 16) TP += pass;                // at least 16 points per tile …
  while (TP > pass && (size_t)TP * per_pt > (f32 ? 32u : 60u) * 1024) TP -= pass;  // … within a modest LDS footprint
  if (tune.tp >= pass && (size_t)tune.tp * per_pt <= 140u * 1024) TP = tune.tp / pass * pass;  // profiling knob
  pl.TP = TP;
  const size_t tile_bytes = (size_t)TP * per_pt + 4 * 16 + 16;
  const size_t red_bytes = (size_t)(WB - 1) * per_wg * kNSums * sizeof(double);
  pl.smem = tile_bytes > red_bytes ? tile_bytes : red_bytes;
  pl.n_tiles = (B + TP - 1) / TP;
  // one resident round of workgroups: a second, partial round would leave most of the chip idle
  int wg_per_cu = occupancy_blocks(pl);
  if (tune.wgpcu_accum >= 1 && tune.wgpcu_accum <= 8) wg_per_cu = tune.wgpcu_accum;  // profiling knob
  int64_t want = (int64_t)num_cus * wg_per_cu / (pl.grid_y > 0 ? pl.grid_y : 1);
  if (want < 1) want = 1;
  int64_t gx = pl.n_tiles < want ? pl.n_tiles : want;
  if (gx < 1) gx = 1;
  pl.tiles_per_block = (int)((pl.n_tiles + gx - 1) / gx);
  if (pl.tiles_per_block < 1) pl.tiles_per_block = 1;
  pl.grid_x = (int)((pl.n_tiles + pl.tiles_per_block - 1) / pl.tiles_per_block);
  if (pl.grid_x < 1) pl.grid_x = 1;
  return pl;
}

// single != nullptr (one particle, fused f32 kernel, no exchange between ranks): the kernel's last workgroup also reduces
// the partial sums and runs the Stein step — the caller launches neither k_reduce_partials nor an update kernel
bool accumulate_can_fuse_single(const AccumPlan& plan) { return plan.f32 == 1; }
int single_particle_grid(int64_t B) { return (int)((B + 63) / 64); }
hipError_t launch_accumulate(const AccumPlan& plan, AccumArgs a, const UpdateArgs* single, hipStream_t st) {
  a.TP = plan.TP; a.RS = plan.RS; a.tiles_per_block = plan.tiles_per_block; a.n_tiles = plan.n_tiles;
  a.Ppad = plan.Ppad;
  a.pts_per_block = plan.pts_per_block; a.spts_per_block = plan.spts_per_block;
  if (plan.f32 == 3) return launch_accumulate_split(plan, a, st);
  if (single && plan.f32 == 1) {   // one particle: lanes along the source points, reduce + Stein step in the last workgroup
    hipLaunchKernelGGL(k_icp_single, dim3(single_particle_grid(a.B)), dim3(NT), 0, st, a, *single);
    return hipGetLastError();
  }
  switch (plan.PW) {
    case 8: return launch_t<8, 1>(plan, a, single, st);
    case 16: return launch_t<16, 1>(plan, a, single, st);
    case 32: return launch_t<32, 1>(plan, a, single, st);
    default:
      if (plan.WP == 1) return launch_t<64, 1>(plan, a, single, st);
      if (plan.WP == 2) return launch_t<64, 2>(plan, a, single, st);
      return launch_t<64, 4>(plan, a, single, st);
  }
}

hipError_t launch_build_table2(const int32_t* idx, int64_t B, int K, const double* tgt, int64_t M, double* table,
                               float4* tablef, float* cmax, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_build_table2, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, idx, B, K, tgt, M, table, tablef, cmax);
  return hipGetLastError();
}

}  // namespace svnicp
