// stein_split.hip — Stage B as two kernels per iteration: winner search on the bf16 matrix pipe, then float64 accumulation;
// and k_build_table3, which lays the candidate table out for the search.
//
//   k_stein_search_bf16   finds, for every (source point, particle), the index of the nearest of the K candidates
//                         (SVGDICP.cpp:300-329 with knn.cu:204-251, K = 1) and writes one byte per pair: float32 scores
//                         from exact bf16x3 operand splits on v_mfma_f32_16x16x32_bf16, a proven error bound that
//                         decides whether the pick is the float64 argmin, an exact float64 pass for the undecided pairs;
//                         no workgroup barrier in the loop, operands straight from global memory, ~112 VGPRs;
//   k_stein_accumulate_w  re-derives Ts in f64, gathers the winner, applies point_filter / weight and accumulates the 22
//                         sums (SVGDICP.cpp:331-333, SVNICP.cpp:116-157) with the loads of the next points in flight.
// Lane ↔ particle in both; the byte array is [B][Ppad] so a wave writes/reads 64 consecutive bytes.
// Reference kernels of the same contract (tests hold this pair to their correspondences bit for bit):
// k_stein_accumulate (float64 throughout) and k_stein_accumulate_f32 (float32 VALU search; the product path for shards of
// <= 8 particles and for K > 128) in stein_iter.hip.  Retired in round 3: the fused f32-MFMA kernel of round 1
// (stein_mfma.hip) and the f32-operand search kernel — no product configuration reached them.
#include "stein_split_device.hpp"

namespace svnicp {

namespace {

// ---------------------------------------------------------------------------------------------
// candidate table of the search kernel: one wave per source point.  Candidates relative to the point's first candidate as
// float32 (c'x, c'y, c'z, |c'|²) in MFMA A-operand order (lane = 16·component + candidate mod 16, one float4 per four 16-row
// blocks; rows past K are finite sentinels), the origin of the local frame and C_b = max |c'|₂ rounded up (the error
// bound's C2).  (Candidates 96…99 — the tile the owner lanes score themselves when K is 97…100 — are row block 6 like any other.)  Replaces the I copies of target_batch [B,K,3] of SVGDICP.cpp:191-198.
// ---------------------------------------------------------------------------------------------
constexpr float kSentinelCC = 1.0e30f; // padded rows: finite, so packed words never become NaN patterns
__global__ __launch_bounds__(256) void k_build_table3(const int32_t* __restrict__ idx, int64_t B, int K,
                                                      const double* __restrict__ tgt, int64_t M, double* __restrict__ table,
                                                      double* __restrict__ anchor, float4* __restrict__ tablea,
                                                      float* __restrict__ cmax) {
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


// (row blocks, tail) with an instantiation: K <= 16, 32, 64, 96, 100 (96 + four VALU candidates), 112, 128
inline int row_code_for(int K) { return K <= 16 ? 1 : K <= 32 ? 2 : K <= 64 ? 4 : K <= 96 ? 6 : K <= 100 ? 60 : K <= 112 ? 7 : 8; }

template <int PW, int WP, int NRB, bool TAIL>
hipError_t launch_s(const AccumPlan& plan, const AccumArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((k_stein_search_bf16<PW, WP, NRB, TAIL>), dim3(plan.sgrid_x, plan.grid_y), dim3(NT), 0, st, a);
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
  if (!a.svgd && !a.corr && !a.full_idx) hipLaunchKernelGGL((k_stein_accumulate_w<PW, WP, true>), dim3(plan.grid_x, plan.grid_y), dim3(NT), plan.smem, st, a);
  else if (a.svgd && !a.corr && !a.full_idx) hipLaunchKernelGGL((k_stein_accumulate_w<PW, WP, true, true>), dim3(plan.grid_x, plan.grid_y), dim3(NT), plan.smem, st, a);
  else hipLaunchKernelGGL((k_stein_accumulate_w<PW, WP, false>), dim3(plan.grid_x, plan.grid_y), dim3(NT), plan.smem, st, a);
  return hipGetLastError();
}

template <int PW, int WP, int NRB, bool TAIL>
hipError_t occ_search(int* n) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, k_stein_search_bf16<PW, WP, NRB, TAIL>, NT, 0);
}
template <int PW, int WP>
void occ_split(int K, size_t smem, int* search, int* accum) {
  int n = 0;
  hipError_t e;
  switch (row_code_for(K)) {
    case 1: e = occ_search<PW, WP, 1, false>(&n); break;
    case 2: e = occ_search<PW, WP, 2, false>(&n); break;
    case 4: e = occ_search<PW, WP, 4, false>(&n); break;
    case 6: e = occ_search<PW, WP, 6, false>(&n); break;
    case 60: e = occ_search<PW, WP, 6, true>(&n); break;
    case 7: e = occ_search<PW, WP, 7, false>(&n); break;
    default: e = occ_search<PW, WP, 8, false>(&n); break;
  }
  *search = (e != hipSuccess || n < 1) ? 4 : (n > 8 ? 8 : n);
  n = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stein_accumulate_w<PW, WP, true>, NT, smem);
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

hipError_t launch_build_table3(const int32_t* idx, int64_t B, int K, const double* tgt, int64_t M, double* table,
                               double* anchor, float4* tablea, float* cmax, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_build_table3, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, idx, B, K, tgt, M, table, anchor, tablea,
                     cmax);
  return hipGetLastError();
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

__global__ __launch_bounds__(256) void k_transform_cloud(const double* __restrict__ src, int64_t B, const double* __restrict__ pose12,
                                                         double* __restrict__ q, const int* __restrict__ ctl) {
  if (ctl[0]) return;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double s0 = src[3 * b], s1 = src[3 * b + 1], s2 = src[3 * b + 2];
  q[3 * b] = (s0 * pose12[0] + s1 * pose12[1] + s2 * pose12[2]) + pose12[9];        // SVNICP.cpp:62-64, as in the stage-B kernels
  q[3 * b + 1] = (s0 * pose12[3] + s1 * pose12[4] + s2 * pose12[5]) + pose12[10];
  q[3 * b + 2] = (s0 * pose12[6] + s1 * pose12[7] + s2 * pose12[8]) + pose12[11];
}

hipError_t launch_transform_cloud(const double* src, int64_t B, const double* pose12, double* q, const int* ctl, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_transform_cloud, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, src, B, pose12, q, ctl);
  return hipGetLastError();
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
