// f64_mfma_mix.hip — would moving the 22 Gauss-Newton sums of the accumulate kernel onto v_mfma_f64_16x16x4_f64 pay?
// Times, per wave step of 64 (point, particle) pairs, (a) 73 dependent-ish float64 vector instructions (the kernel as it
// is), (b) 53 vector instructions + 4 f64 MFMAs (sums on the matrix pipe, B operand prepared once per two particle
// blocks), (c) 45 + 4, (d) the 4 MFMAs alone — wall time of a launch that fills the chip with W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o f64_mfma_mix f64_mfma_mix.hip && ./f64_mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int STEPS = 2048;

template <int NV, int NM>
__global__ __launch_bounds__(256) void k(double* out, double seed) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3 + i;
  v4d acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = v4d{0, 0, 0, 0};
  for (int s = 0; s < STEPS; ++s) {
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i & 7] = __builtin_fma(x[(i + 1) & 7], 1.0000001, x[(i + 3) & 7] * 0.5);   // 2 f64 instructions each
#pragma unroll
    for (int g = 0; g < NM; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[g], x[g + 4], acc[g], 0, 0, 0);
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += x[i];
#pragma unroll
  for (int g = 0; g < 4; ++g) r += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NV, int NM>
int run(const char* tag, int wgs_per_cu, double* d) {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int grid = p.multiProcessorCount * wgs_per_cu;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<NV, NM>), dim3(grid), dim3(256), 0, 0, d, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NV, NM>), dim3(grid), dim3(256), 0, 0, d, 1.0);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double steps = (double)STEPS * wgs_per_cu;   // wave steps per SIMD (one wave of each workgroup per SIMD)
  printf("%-34s %d waves/SIMD: %7.1f ns per wave step per SIMD\n", tag, wgs_per_cu, 1e6 * (ms / 5) / steps);
  return 0;
}

int main() {
  double* d; CHECK(hipMalloc(&d, sizeof(double) * 256 * 256 * 8));
  for (int w = 3; w <= 4; ++w) {
    run<36, 0>("73 f64 VALU", w, d);              // 36 x 2 + loop = ~73
    run<26, 4>("53 f64 VALU + 4 MFMA f64", w, d);
    run<22, 4>("45 f64 VALU + 4 MFMA f64", w, d);
    run<0, 4>("4 MFMA f64 only", w, d);
    run<26, 0>("53 f64 VALU", w, d);
  }
  return 0;
}
