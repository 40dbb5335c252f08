// f64_seed_accuracy.hip — relative accuracy of v_rsq_f64 and v_rcp_f64 on gfx950: the accumulate kernel
// (svn-icp_amd/csrc/stein_split.hip, k_stein_accumulate_w) refines these seeds by hand and its error bound starts here.
//   hipcc --offload-arch=gfx950 -O3 -o f64_seed_accuracy f64_seed_accuracy.hip && ./f64_seed_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>

__global__ void k(const double* x, double* rsq, double* rcp, double* root, double* quot, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  rsq[i] = __builtin_amdgcn_rsq(v);
  rcp[i] = __builtin_amdgcn_rcp(v);
  {  // the kernel's refined square root
    const double r0 = __builtin_amdgcn_rsq(v + 0x1p-1000);
    const double sa = v * r0, h0 = 0.5 * r0;
    const double ea = fma(-h0, sa, 0.5);
    const double sb = fma(sa, ea, sa);
    root[i] = fma(fma(-sb, sb, v), h0, sb);
  }
  {  // the kernel's refined reciprocal
    const double y0 = __builtin_amdgcn_rcp(v);
    const double y1 = fma(y0, fma(-v, y0, 1.0), y0);
    quot[i] = fma(y1, fma(-v, y1, 1.0), y1);
  }
}

int main() {
  const int n = 1 << 22;
  std::vector<double> hx(n), a(n), b(n), c(n), d(n);
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + U(rng), (int)(rng() % 80) - 60);   // 2^-60 .. 2^20
  double *dx, *d0, *d1, *d2, *d3;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
  hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) {
    const long double x = hx[i];
    e0 = fmax(e0, (double)fabsl((long double)a[i] * sqrtl(x) - 1.0L));
    e1 = fmax(e1, (double)fabsl((long double)b[i] * x - 1.0L));
    e2 = fmax(e2, (double)fabsl((long double)c[i] / sqrtl(x) - 1.0L));
    e3 = fmax(e3, (double)fabsl((long double)d[i] * x - 1.0L));
  }
  printf("v_rsq_f64 max relative error %.3e (2^%.1f)   v_rcp_f64 %.3e (2^%.1f)\n", e0, log2(e0), e1, log2(e1));
  printf("refined sqrt  max relative error %.3e (%.2f x 2^-53)   refined reciprocal %.3e (%.2f x 2^-53)\n", e2, e2 / ldexp(1.0, -53), e3, e3 / ldexp(1.0, -53));
  return 0;
}
