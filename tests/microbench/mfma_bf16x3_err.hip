// mfma_bf16x3_err.hip — how accurately does v_mfma_f32_16x16x32_bf16 reproduce the f32 score
//   S = beta + cc + c . m        (c, m float32 3-vectors, cc, beta float32)
// when every float32 operand is split EXACTLY into three bf16 pieces (a = a1 + a2 + a3) and the six products
// a_i b_j with i + j <= 4 of each component go into the K slots?  Compared with the exact value in float64 of the
// SAME float32 inputs; reported in units of u = 2^-24 times (i) the sum of the magnitudes of the terms and
// (ii) (C + X)^2 — the two normalisations the search kernel's error bound is written in (stein_split.hip).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_bf16x3_err mfma_bf16x3_err.hip && ./mfma_bf16x3_err
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk(float lo, float hi) {
  f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
// A pattern [a1,a1,a1,a2,a2,a3,0,0]
__device__ __forceinline__ bf8 split_a(float x) {
  const unsigned r0 = pk(x, x);
  const float e1 = x - __uint_as_float(r0 << 16);
  const unsigned r1 = pk(x, e1);
  const float e2 = e1 - __uint_as_float(r1 & 0xffff0000u);
  const unsigned r2 = pk(e1, e2);
  return __builtin_bit_cast(bf8, (u4){r0, r1, r2, 0u});
}
// B pattern [b1,b2,b3,b1,b2,b1,0,0]
__device__ __forceinline__ bf8 split_b(float x) {
  const unsigned q0 = pk(x, x);                         // [b1,b1]
  const float e1 = x - __uint_as_float(q0 << 16);
  const unsigned r0 = pk(x, e1);                        // [b1,b2]
  const float e2 = e1 - __uint_as_float(r0 & 0xffff0000u);
  const unsigned r1 = pk(e2, x);                        // [b3,b1]
  const unsigned r2 = pk(e1, x);                        // [b2,b1]
  return __builtin_bit_cast(bf8, (u4){r0, r1, r2, 0u});
}

// one wave per tile: c [16][4] (cx,cy,cz,cc), m [16][4] (mx,my,mz,beta); out [16 rows][16 cols]
__global__ __launch_bounds__(64) void k(const float4* __restrict__ c, const float4* __restrict__ m, float* __restrict__ out, int exact_check, int* bad) {
  const int lane = threadIdx.x, g = lane >> 4, r = lane & 15;
  const float4* ct = c + (size_t)blockIdx.x * 16;
  const float4* mt = m + (size_t)blockIdx.x * 16;
  const float4 cr = ct[r], mr = mt[r];
  const float va = g == 0 ? cr.x : g == 1 ? cr.y : g == 2 ? cr.z : cr.w;
  const float vb = g == 0 ? mr.x : g == 1 ? mr.y : g == 2 ? mr.z : 1.0f;
  const bf8 A = split_a(va), B = split_b(vb);
  if (exact_check) {  // the three pieces must add up to the float32 value exactly
    const u4 a = __builtin_bit_cast(u4, A);
    const float p1 = __uint_as_float(a[0] << 16), p2 = __uint_as_float(a[1] & 0xffff0000u), p3 = __uint_as_float(a[2] & 0xffff0000u);
    if ((double)p1 + (double)p2 + (double)p3 != (double)va) atomicAdd(bad, 1);
  }
  const v4f cin = {mr.w, mr.w, mr.w, mr.w};
  const v4f d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, cin, 0, 0, 0);
  float* o = out + (size_t)blockIdx.x * 256;
#pragma unroll
  for (int v = 0; v < 4; ++v) o[(4 * g + v) * 16 + r] = d[v];
}

int main() {
  const int tiles = 1 << 16;
  std::vector<float4> hc((size_t)tiles * 16), hm((size_t)tiles * 16);
  std::vector<float> ho((size_t)tiles * 256);
  float4 *dc, *dm; float* dout; int* dbad;
  CHECK(hipMalloc(&dc, hc.size() * sizeof(float4))); CHECK(hipMalloc(&dm, hm.size() * sizeof(float4)));
  CHECK(hipMalloc(&dout, ho.size() * sizeof(float))); CHECK(hipMalloc(&dbad, sizeof(int)));
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const double u = ldexp(1.0, -24);
  struct Case { const char* name; double C, X; int near; } cases[] = {
      {"C=0.3 X=0.3", 0.3, 0.3, 0}, {"C=0.3 X=0.3 near (x~c)", 0.3, 0.3, 1}, {"C=1e-3 X=1e-3", 1e-3, 1e-3, 0},
      {"C=0.05 X=2.0", 0.05, 2.0, 0}, {"C=2.0 X=0.05", 2.0, 0.05, 0}, {"C=300 X=300 near", 300.0, 300.0, 1},
      {"C=1e-5 X=1e3", 1e-5, 1e3, 0}};
  for (const Case& cs : cases) {
    for (int t = 0; t < tiles; ++t) {
      for (int i = 0; i < 16; ++i) {
        const float cx = (float)(cs.C * U(rng)), cy = (float)(cs.C * U(rng)), cz = (float)(cs.C * U(rng));
        const float cc = (float)(((double)cx * cx + (double)cy * cy) + (double)cz * cz);
        hc[(size_t)t * 16 + i] = make_float4(cx, cy, cz, cc);
      }
      for (int j = 0; j < 16; ++j) {
        float x0, x1, x2;
        if (cs.near) {  // particle next to candidate j: heavy cancellation in the score
          const float4 q = hc[(size_t)t * 16 + j];
          x0 = q.x + (float)(1e-3 * cs.C * U(rng)); x1 = q.y + (float)(1e-3 * cs.C * U(rng)); x2 = q.z + (float)(1e-3 * cs.C * U(rng));
        } else { x0 = (float)(cs.X * U(rng)); x1 = (float)(cs.X * U(rng)); x2 = (float)(cs.X * U(rng)); }
        const float X = fmaxf(fabsf(x0), fmaxf(fabsf(x1), fabsf(x2)));
        const float E = 48.0f * (float)u * (float)((cs.C + X) * (cs.C + X));
        const float beta = fmaf(x0, x0, fmaf(x1, x1, x2 * x2)) + 4.0f * E;
        hm[(size_t)t * 16 + j] = make_float4(-2.0f * x0, -2.0f * x1, -2.0f * x2, beta);
      }
    }
    CHECK(hipMemcpy(dc, hc.data(), hc.size() * sizeof(float4), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dm, hm.data(), hm.size() * sizeof(float4), hipMemcpyHostToDevice));
    CHECK(hipMemset(dbad, 0, sizeof(int)));
    hipLaunchKernelGGL(k, dim3(tiles), dim3(64), 0, 0, dc, dm, dout, 1, dbad);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ho.data(), dout, ho.size() * sizeof(float), hipMemcpyDeviceToHost));
    int bad = 0; CHECK(hipMemcpy(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost));
    double worst_t = 0, worst_cx = 0, mean_t = 0; size_t n = 0;
    for (int t = 0; t < tiles; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          const float4 c = hc[(size_t)t * 16 + i], m = hm[(size_t)t * 16 + j];
          const double exact = (double)m.w + (double)c.w + ((double)c.x * m.x + (double)c.y * m.y + (double)c.z * m.z);
          const double mag = fabs((double)m.w) + fabs((double)c.w) + fabs((double)c.x * m.x) + fabs((double)c.y * m.y) + fabs((double)c.z * m.z);
          const double err = fabs((double)ho[(size_t)t * 256 + i * 16 + j] - exact);
          const double Cm = fmax(fabs(c.x), fmax(fabs(c.y), fabs(c.z))), Xm = 0.5 * fmax(fabs(m.x), fmax(fabs(m.y), fabs(m.z)));
          worst_t = fmax(worst_t, err / (u * mag));
          worst_cx = fmax(worst_cx, err / (u * (Cm + Xm) * (Cm + Xm)));
          mean_t += err / (u * mag); ++n;
        }
    printf("%-26s split-not-exact lanes %d | max err = %.3f u*sum|terms| (mean %.3f) = %.3f u*(C+X)^2 over %zu scores\n", cs.name, bad,
           worst_t, mean_t / n, worst_cx, n);
  }
  return 0;
}
