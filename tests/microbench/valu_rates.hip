// valu_rates.hip — measures sustained VALU issue rates on gfx950 for the instruction mix the NN
// kernels use (calibrates the roofline peaks quoted in DESIGN.md / bench.py).  Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

template <int OP> __global__ __launch_bounds__(256) void k(float* out, float seed) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (OP == 0) {  // v_fma_f32
    float a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + tid;
    for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fmaf(a[c], 1.0000001f, 0.5f); }
    float s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c]; out[tid] = s;
  } else if constexpr (OP == 1) {  // v_pk_fma_f32
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = f2{seed + c + tid, seed - c};
    const f2 m = {1.0000001f, 0.9999999f}, b = {0.5f, 0.25f};
    for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_elementwise_fma(a[c], m, b); }
    float s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c].x + a[c].y; out[tid] = s;
  } else if constexpr (OP == 2) {  // v_fma_f64
    double a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + tid;
    for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fma(a[c], 1.0000001, 0.5); }
    double s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c]; out[tid] = (float)s;
  } else if constexpr (OP == 3) {  // v_add_f64
    double a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + tid;
    for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) a[c] = a[c] + 0.5; }
    double s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c]; out[tid] = (float)s;
  } else if constexpr (OP == 4) {  // v_mul_f64
    double a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + tid;
    for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) a[c] = a[c] * 1.0000001; }
    double s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c]; out[tid] = (float)s;
  } else if constexpr (OP == 5) {  // v_cmp_le_f32 + s_or (ballot)
    float a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + tid;
    unsigned long long acc = 0;
    for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) acc |= __ballot(a[c] <= (float)(i + c)); }
    out[tid] = (float)(acc & 0xffff);
  } else if constexpr (OP == 6) {  // v_cmp_le_f64
    double a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + tid;
    unsigned long long acc = 0;
    for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) acc |= __ballot(a[c] <= (double)(i + c)); }
    out[tid] = (float)(acc & 0xffff);
  } else {  // v_add_f32
    float a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + tid;
    for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) a[c] = a[c] + 0.5f; }
    float s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c]; out[tid] = s;
  }
}

template <int OP> int run(const char* name, int waves_per_simd, float* d) {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double wave_instr = (double)blocks * 4 * ITERS * CHAINS;         // per launch
  const double simds = cus * 4.0;
  const double ns_per_instr_per_simd = ms * 1e6 / (wave_instr / simds);
  printf("%-14s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instruction per SIMD  => %.2f T lane-ops/s chip-wide (%.2f cycles @2.4GHz)\n",
         name, waves_per_simd, ms, ns_per_instr_per_simd, wave_instr * 64 / (ms * 1e-3) / 1e12, ns_per_instr_per_simd * 2.4);
  return 0;
}

int main() {
  float* d; CHECK(hipMalloc(&d, 256 * 1024 * 16 * sizeof(float)));
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", w, d); run<7>("v_add_f32", w, d); run<1>("v_pk_fma_f32", w, d); run<2>("v_fma_f64", w, d);
    run<3>("v_add_f64", w, d); run<4>("v_mul_f64", w, d); run<5>("v_cmp_f32", w, d); run<6>("v_cmp_f64", w, d);
  }
  return 0;
}
