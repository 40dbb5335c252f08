// argmin_rates.hip — cost of per-candidate argmin tracking idioms on gfx950 (wave-instruction time per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 2048;

template <int OP> __global__ __launch_bounds__(256) void k(float* out, const float4* __restrict__ tab, float seed) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ float4 rows[128];
  if (threadIdx.x < 128) rows[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const float m0 = seed + (tid & 7) * 1e-3f, m1 = seed * 0.5f, m2 = seed * 0.25f;
  float best1 = 1e30f, best2 = 1e30f; int kb = 0; unsigned ub1 = 0xffffffffu, ub2 = 0xffffffffu;
  const float bias = 3.0f;
  for (int it = 0; it < ITERS / 128; ++it) {
#pragma unroll 4
    for (int kk = 0; kk < 128; ++kk) {
      const float4 c = rows[kk];
      if constexpr (OP == 0) {          // fma only
        const float sc = __builtin_fmaf(c.z, m2, __builtin_fmaf(c.y, m1, __builtin_fmaf(c.x, m0, c.w)));
        best1 = __builtin_fminf(best1, sc);
      } else if constexpr (OP == 1) {   // current: cmp + cndmask + med3 + min
        const float sc = __builtin_fmaf(c.z, m2, __builtin_fmaf(c.y, m1, __builtin_fmaf(c.x, m0, c.w)));
        const bool lt = sc < best1;
        best2 = __builtin_amdgcn_fmed3f(best1, best2, sc);
        best1 = __builtin_fminf(best1, sc);
        kb = lt ? kk : kb;
      } else if constexpr (OP == 2) {   // packed key: (bits & ~127) | k, unsigned min / med3
        const float sc = __builtin_fmaf(c.z, m2, __builtin_fmaf(c.y, m1, __builtin_fmaf(c.x, m0, c.w + bias)));
        const unsigned key = (__float_as_uint(sc) & 0xffffff80u) | (unsigned)kk;
        const unsigned lo = key < ub1 ? key : ub1;           // v_min_u32
        const unsigned hi = key < ub1 ? ub1 : key;           // v_max_u32
        ub2 = hi < ub2 ? hi : ub2;                           // v_min_u32
        ub1 = lo;
      } else if constexpr (OP == 3) {   // med3 + min only (no index)
        const float sc = __builtin_fmaf(c.z, m2, __builtin_fmaf(c.y, m1, __builtin_fmaf(c.x, m0, c.w)));
        best2 = __builtin_amdgcn_fmed3f(best1, best2, sc);
        best1 = __builtin_fminf(best1, sc);
      }
    }
  }
  out[tid] = best1 + best2 + kb + (float)(ub1 & 127) + (float)ub2;
}

template <int OP> int run(const char* name, int wps, float* d, float4* tab, int per) {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * wps;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, tab, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, tab, 1.0f);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double cand = (double)blocks * 4 * ITERS;  // wave-candidates
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f ns per candidate per SIMD (%d VALU/cand => %.2f ns each)\n", name, wps, ms,
         ms * 1e6 / (cand / (p.multiProcessorCount * 4.0)), per, ms * 1e6 / (cand / (p.multiProcessorCount * 4.0)) / per);
  return 0;
}
int main() {
  float* d; float4* tab; CHECK(hipMalloc(&d, 4 << 20)); CHECK(hipMalloc(&tab, 128 * 16)); CHECK(hipMemset(tab, 0, 128 * 16));
  for (int w : {4, 6, 8}) {
    run<0>("3fma+min", w, d, tab, 4); run<3>("3fma+med3+min", w, d, tab, 5);
    run<1>("3fma+cmp+cndmask+med3+min", w, d, tab, 7); run<2>("add+3fma+andor+umin/umax/umin", w, d, tab, 8);
  }
  return 0;
}
