// spatial_prep.hip — spatial ordering for the pruned stage-A kernel (knn_tiles.hip).
//
// Targets and (transformed) queries are sorted along a 30-bit Morton curve (rocPRIM device radix
// sort; the pair loop, not the sort, is the hot op).  Targets are then stored tile-wise: 512
// consecutive curve points form a tile with a float32 bounding box rounded OUTWARD, so that a
// box-to-box / point-to-box distance is a rigorous lower bound of every pair distance inside.
// Nothing here changes results: the ordering only decides which tiles can be skipped.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.hpp"

namespace svnicp {

namespace {

// order-preserving map double -> uint64 (for atomicMin/atomicMax on signed values)
__device__ __forceinline__ unsigned long long enc_f64(double v) {
  const long long b = __double_as_longlong(v);
  return b < 0 ? ~(unsigned long long)b : ((unsigned long long)b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64(unsigned long long e) {
  const unsigned long long b = (e & 0x8000000000000000ull) ? (e & 0x7fffffffffffffffull) : ~e;
  return __longlong_as_double((long long)b);
}
__device__ __forceinline__ float next_down(float f) {  // largest float < f (f finite)
  if (f == 0.0f) return -1.401298464e-45f;
  int b = __float_as_int(f);
  b += (f > 0.0f) ? -1 : 1;
  return __int_as_float(b);
}
__device__ __forceinline__ float next_up(float f) {
  if (f == 0.0f) return 1.401298464e-45f;
  int b = __float_as_int(f);
  b += (f > 0.0f) ? 1 : -1;
  return __int_as_float(b);
}
__device__ __forceinline__ float f32_floor(double v) { float f = (float)v; return ((double)f > v) ? next_down(f) : f; }
__device__ __forceinline__ float f32_ceil(double v) { float f = (float)v; return ((double)f < v) ? next_up(f) : f; }

// bbox[0..2] = min xyz, bbox[3..5] = max xyz (encoded); NaN coordinates are ignored
__global__ __launch_bounds__(256) void k_bbox(const double* __restrict__ pts, int64_t n, unsigned long long* __restrict__ bbox) {
  double lo[3] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val()};
  double hi[3] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { const double v = pts[3 * i + d]; lo[d] = fmin(lo[d], v); hi[d] = fmax(hi[d], v); }  // fmin/fmax skip NaN
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    for (int off = 32; off > 0; off >>= 1) {
      lo[d] = fmin(lo[d], __shfl_xor(lo[d], off, kWave));
      hi[d] = fmax(hi[d], __shfl_xor(hi[d], off, kWave));
    }
  }
  // one set of atomics per workgroup (same-address atomics from every wave serialise at L2)
  __shared__ double s_lo[4][3], s_hi[4][3];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { s_lo[wave][d] = lo[d]; s_hi[wave][d] = hi[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    double l = s_lo[0][d], h = s_hi[0][d];
    for (int w = 1; w < 4; ++w) { l = fmin(l, s_lo[w][d]); h = fmax(h, s_hi[w][d]); }
    if (l <= h) { atomicMin(&bbox[d], enc_f64(l)); atomicMax(&bbox[3 + d], enc_f64(h)); }
  }
}


__device__ __forceinline__ unsigned int spread10(unsigned int v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// 30-bit Morton key of point i (optionally transformed by R0,t0) inside the encoded bbox
__global__ void k_morton_keys(const double* __restrict__ pts, int64_t i0, int64_t n, int transform, Pose0 pose,
                              const unsigned long long* __restrict__ bbox, unsigned int* __restrict__ keys,
                              int32_t* __restrict__ vals) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int64_t i = i0 + e;
  double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  if (transform) {
    const double* R = pose.R0;
    const double x = (p[0] * R[0] + p[1] * R[1] + p[2] * R[2]) + pose.t0[0];
    const double y = (p[0] * R[3] + p[1] * R[4] + p[2] * R[5]) + pose.t0[1];
    const double z = (p[0] * R[6] + p[1] * R[7] + p[2] * R[8]) + pose.t0[2];
    p[0] = x; p[1] = y; p[2] = z;
  }
  unsigned int c[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double lo = dec_f64(bbox[d]), hi = dec_f64(bbox[3 + d]);
    double u = (hi > lo) ? (p[d] - lo) / (hi - lo) : 0.0;
    if (!(u > 0.0)) u = 0.0;          // also NaN
    if (u > 0.999999) u = 0.999999;
    c[d] = (unsigned int)(u * 1024.0);
  }
  keys[e] = spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2);
  vals[e] = (int32_t)i;
}

// one workgroup per 512-slot tile: gather the curve-ordered targets, write SoA (f64 + f32) and the
// original index, reduce the tile's outward-rounded float32 box and the global max |coordinate|
__global__ __launch_bounds__(256) void k_targets_sorted(const double* __restrict__ tgt, int64_t M,
                                                        const int32_t* __restrict__ order, double* __restrict__ tx,
                                                        double* __restrict__ ty, double* __restrict__ tz,
                                                        float* __restrict__ txf, float* __restrict__ tyf,
                                                        float* __restrict__ tzf, int32_t* __restrict__ torig,
                                                        float* __restrict__ tile_box /* [6][n_tiles] */, int n_tiles,
                                                        unsigned long long* __restrict__ emax_bits) {
  __shared__ float red[4][6];
  const int tile = blockIdx.x;
  float lo[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()};
  float hi[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
  double e = 0.0;
  for (int r = threadIdx.x; r < 512; r += 256) {
    const int64_t j = (int64_t)tile * 512 + r;
    const double nan = __builtin_nan("");
    double x = nan, y = nan, z = nan;
    int32_t o = 0;
    if (j < M) {
      o = order[j];
      x = tgt[3 * (int64_t)o]; y = tgt[3 * (int64_t)o + 1]; z = tgt[3 * (int64_t)o + 2];
      const double m = fmax(fabs(x), fmax(fabs(y), fabs(z)));
      e = fmax(e, (m == m) ? m : __builtin_huge_val());  // NaN input disables the float32 filter
      if (x == x) { lo[0] = __builtin_fminf(lo[0], f32_floor(x)); hi[0] = __builtin_fmaxf(hi[0], f32_ceil(x)); }
      if (y == y) { lo[1] = __builtin_fminf(lo[1], f32_floor(y)); hi[1] = __builtin_fmaxf(hi[1], f32_ceil(y)); }
      if (z == z) { lo[2] = __builtin_fminf(lo[2], f32_floor(z)); hi[2] = __builtin_fmaxf(hi[2], f32_ceil(z)); }
    }
    tx[j] = x; ty[j] = y; tz[j] = z;
    txf[j] = (float)x; tyf[j] = (float)y; tzf[j] = (float)z;
    torig[j] = o;
  }
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = __builtin_fminf(lo[d], __shfl_xor(lo[d], off, kWave));
      hi[d] = __builtin_fmaxf(hi[d], __shfl_xor(hi[d], off, kWave));
    }
    e = fmax(e, __shfl_xor(e, off, kWave));
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { red[wave][d] = lo[d]; red[wave][3 + d] = hi[d]; }
    if (e > 0.0) atomicMax(emax_bits, (unsigned long long)__double_as_longlong(e));
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int d = threadIdx.x;
    float v = red[0][d];
    for (int w = 1; w < 4; ++w) v = d < 3 ? __builtin_fminf(v, red[w][d]) : __builtin_fmaxf(v, red[w][d]);
    tile_box[(size_t)d * n_tiles + tile] = v;  // empty tile: lo = +inf, hi = -inf  => never needed
  }
}

}  // namespace

size_t sort_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned int*)nullptr, (unsigned int*)nullptr, (int32_t*)nullptr,
                                  (int32_t*)nullptr, n, 0, 30, (hipStream_t)0);
  return bytes;
}

// order[0..n) = indices i0..i0+n of pts sorted along the Morton curve of `bbox`
hipError_t launch_morton_order(const double* pts, int64_t i0, int64_t n, int transform, const Pose0& pose,
                               const unsigned long long* bbox, unsigned int* keys_a, unsigned int* keys_b,
                               int32_t* vals_a, int32_t* order, void* temp, size_t temp_bytes, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_morton_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pts, i0, n, transform, pose, bbox,
                     keys_a, vals_a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_a, keys_b, vals_a, order, (size_t)n, 0, 30, st);
}

hipError_t launch_bbox(const double* pts, int64_t n, unsigned long long* bbox, hipStream_t st) {
  hipError_t e = hipMemsetAsync(bbox, 0xff, 3 * sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(bbox + 3, 0, 3 * sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  int64_t nb = (n + 255) / 256;
  if (nb > 256) nb = 256;  // grid-stride; one set of atomics per workgroup (1024 workgroups: 27 us, most of it the 6 x 1024 same-address atomics)
  hipLaunchKernelGGL(k_bbox, dim3((unsigned)nb), dim3(256), 0, st, pts, n, bbox);
  return hipGetLastError();
}

hipError_t launch_targets_sorted(const double* tgt, int64_t M, int64_t Mp, const int32_t* order, double* tx, double* ty,
                                 double* tz, float* txf, float* tyf, float* tzf, int32_t* torig, float* tile_box,
                                 unsigned long long* emax_bits, hipStream_t st) {
  hipError_t e = hipMemsetAsync(emax_bits, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  const int n_tiles = (int)(Mp / 512);
  hipLaunchKernelGGL(k_targets_sorted, dim3(n_tiles), dim3(256), 0, st, tgt, M, order, tx, ty, tz, txf, tyf, tzf, torig,
                     tile_box, n_tiles, emax_bits);
  return hipGetLastError();
}

}  // namespace svnicp
