// stein_common.hpp — device helpers shared by the stage-B kernels (stein_iter.hip, stein_split.hip).
#pragma once
#include "kernels.hpp"

namespace svnicp {
namespace {

constexpr int NT = 256;

// Workgroups are dealt to the eight XCDs round-robin (workgroup i runs on XCD i mod 8) and every XCD has its own L2.
// xcd_block() renumbers the blocks so that each XCD works on ONE contiguous eighth of the block range: neighbouring source
// points (scan order is spatially coherent) then gather their winners' coordinates through the same L2 instead of all eight.
__device__ __forceinline__ int xcd_block(int i, int n) {
  constexpr int X = 8;
  const int x = i % X, j = i / X, q = n / X, r = n % X;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

__device__ __forceinline__ double rdlane_f64(double v, int l) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// one (particle, source point) result waiting for its winner's f64 coordinates
struct Pending {
  double T0, T1, T2;   // transformed source point
  double q0, q1, q2;   // winner candidate (f64), loaded one search-loop ago
  int pt;              // point index inside the tile (source point re-read from LDS)
};

__device__ __forceinline__ void accumulate_point(const Pending& pd, const double* spts, double max_dist, int svgd, double* acc) {
  const double dx = pd.T0 - pd.q0, dy = pd.T1 - pd.q1, dz = pd.T2 - pd.q2;
  const double best = (dx * dx + dy * dy) + dz * dz;   // exact d² of the winner (knn_cpu.cpp:43-50 order)
  double w = 1.0, e0 = 0.0, e1 = 0.0, e2 = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0;
  if (best < max_dist) {  // point_filter, SVGDICP.cpp:331-333
    const double n = sqrt(best);                        // SVNICP.cpp:120
    const double wq = max_dist / (max_dist + 3 * n);
    w = wq * wq;                                        // SVNICP.cpp:122
    e0 = w * dx; e1 = w * dy; e2 = w * dz;              // SVNICP.cpp:119,123
    n0 = spts[3 * pd.pt]; n1 = spts[3 * pd.pt + 1]; n2 = spts[3 * pd.pt + 2];
  } else if (best != best) {   // masking is a multiplication in the reference (SVGDICP.cpp:331-333): a NaN row stays NaN
    w = best; e0 = best; e1 = best; e2 = best;
  }
  const double w0 = w * n0, w1 = w * n1, w2 = w * n2;
  acc[0] += w;
  acc[1] += w0; acc[2] += w1; acc[3] += w2;
  // SVGD mode needs count_nonzero(mask·Ts summed over xyz) (SVGDICP.cpp:404) instead of Σw·s_x²
  acc[4] = svgd ? acc[4] + ((best < max_dist && ((pd.T0 + pd.T1) + pd.T2) != 0.0) ? 1.0 : 0.0) : fma(w0, n0, acc[4]);
  acc[5] = fma(w0, n1, acc[5]); acc[6] = fma(w0, n2, acc[6]);
  acc[7] = fma(w1, n1, acc[7]); acc[8] = fma(w1, n2, acc[8]); acc[9] = fma(w2, n2, acc[9]);
  acc[10] += e0; acc[11] += e1; acc[12] += e2;
  acc[13] = fma(e0, n0, acc[13]); acc[14] = fma(e0, n1, acc[14]); acc[15] = fma(e0, n2, acc[15]);
  acc[16] = fma(e1, n0, acc[16]); acc[17] = fma(e1, n1, acc[17]); acc[18] = fma(e1, n2, acc[18]);
  acc[19] = fma(e2, n0, acc[19]); acc[20] = fma(e2, n1, acc[20]); acc[21] = fma(e2, n2, acc[21]);
}


}  // namespace
}  // namespace svnicp
