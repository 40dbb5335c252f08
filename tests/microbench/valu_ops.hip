// valu_ops.hip — sustained issue cost of single VALU instructions on gfx950 (SIMD cycles per wave64
// instruction at 1/2/4/8 waves per SIMD), by inline asm so the compiler cannot rewrite them.
//   hipcc --offload-arch=gfx950 -O3 -o valu_ops valu_ops.hip && ./valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int CH = 8;  // independent chains

#define OP3(name, text)                                                                          \
  struct name { static __device__ __forceinline__ void go(float& a, float b, float c) {         \
    asm volatile(text : "+v"(a) : "v"(b), "v"(c)); } static const char* nm() { return #name; } };

OP3(fma_f32,      "v_fma_f32 %0, %0, %1, %2")
OP3(add_f32,      "v_add_f32 %0, %0, %1")
OP3(min_f32,      "v_min_f32 %0, %0, %1")
OP3(min_i32,      "v_min_i32 %0, %0, %1")
OP3(min_u32,      "v_min_u32 %0, %0, %1")
OP3(min3_f32,     "v_min3_f32 %0, %0, %1, %2")
OP3(min3_i32,     "v_min3_i32 %0, %0, %1, %2")
OP3(med3_f32,     "v_med3_f32 %0, %0, %1, %2")
OP3(med3_i32,     "v_med3_i32 %0, %0, %1, %2")
OP3(and_or_b32,   "v_and_or_b32 %0, %0, %1, %2")
OP3(and_or_imm,   "v_and_or_b32 %0, %0, %1, 5")
OP3(bfi_b32,      "v_bfi_b32 %0, %1, %0, %2")
OP3(perm_b32,     "v_perm_b32 %0, %0, %1, %2")
OP3(and_b32,      "v_and_b32 %0, %0, %1")
OP3(or_b32,       "v_or_b32 %0, %0, %1")
OP3(add_u32,      "v_add_u32 %0, %0, %1")
OP3(lshl_or,      "v_lshl_or_b32 %0, %0, 1, %1")
OP3(mov_b32,      "v_mov_b32 %0, %1")
OP3(max_f32,      "v_max_f32 %0, %0, %1")
OP3(mul_f32,      "v_mul_f32 %0, %0, %1")
OP3(cndmask,      "v_cndmask_b32 %0, %0, %1, vcc")
OP3(cmp_lt_f32,   "v_cmp_lt_f32 vcc, %0, %1")
OP3(pk_min_f16,   "v_pk_min_f16 %0, %0, %1")
OP3(pk_add_f16,   "v_pk_add_f16 %0, %0, %1")
OP3(cvt_pkrtz,    "v_cvt_pkrtz_f16_f32 %0, %0, %1")
OP3(min_dpp,      "v_min_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OP3(mov_dpp,      "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OP3(sad_u32,      "v_sad_u32 %0, %0, %1, %2")
OP3(minimum3,     "v_minimum3_f32 %0, %0, %1, %2")
OP3(bitop3,       "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8")
OP3(bitop3_imm,   "v_bitop3_b32 %0, %0, 31, 5 bitop3:0xba")
OP3(sub_f32,      "v_sub_f32 %0, %0, %1")
OP3(fmac_f32,     "v_fmac_f32 %0, %1, %2")
OP3(xor_b32,      "v_xor_b32 %0, %0, %1")
OP3(lshlrev,      "v_lshlrev_b32 %0, 1, %0")
OP3(lshrrev,      "v_lshrrev_b32 %0, 1, %0")
OP3(ashrrev,      "v_ashrrev_i32 %0, 1, %0")
OP3(sub_u32,      "v_sub_u32 %0, %0, %1")
OP3(add3_u32,     "v_add3_u32 %0, %0, %1, %2")
OP3(cvt_pk_bf16,  "v_cvt_pk_bf16_f32 %0, %0, %1")
OP3(max_u32,      "v_max_u32 %0, %0, %1")
OP3(max_u16,      "v_max_u16 %0, %0, %1")
OP3(min_f16,      "v_min_f16 %0, %0, %1")
OP3(alignbit,     "v_alignbit_b32 %0, %0, %1, 16")
OP3(mad_u32_u24,  "v_mad_u32_u24 %0, %0, %1, %2")
OP3(cvt_f32_i32,  "v_cvt_f32_i32 %0, %0")
OP3(xad_u32,      "v_xad_u32 %0, %0, %1, %2")
OP3(fma_f16,      "v_fma_f16 %0, %0, %1, %2")
OP3(pk_fma_f16,   "v_pk_fma_f16 %0, %0, %1, %2")

template <class O>
__global__ __launch_bounds__(256) void k32(float* out, long long* cyc, float seed) {
  float a[CH];
  for (int c = 0; c < CH; ++c) a[c] = seed + c + threadIdx.x;
  const float b = seed * 3.0f, cc = seed * 0.5f + threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) O::go(a[c], b, cc);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CH; ++c) s += a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

#define OP64(name, text)                                                                         \
  struct name { static __device__ __forceinline__ void go(double& a, double b, double c) {      \
    asm volatile(text : "+v"(a) : "v"(b), "v"(c)); } static const char* nm() { return #name; } };
OP64(fma_f64,  "v_fma_f64 %0, %0, %1, %2")
OP64(add_f64,  "v_add_f64 %0, %0, %1")
OP64(mul_f64,  "v_mul_f64 %0, %0, %1")
OP64(min_f64,  "v_min_f64 %0, %0, %1")
OP64(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
OP64(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
OP64(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
OP64(pk_mov_b32, "v_pk_mov_b32 %0, %1, %2")
OP64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")

template <class O>
__global__ __launch_bounds__(256) void k64(float* out, long long* cyc, float seed) {
  double a[CH];
  for (int c = 0; c < CH; ++c) a[c] = seed + c + threadIdx.x;
  const double b = seed * 3.0, cc = seed * 0.5 + threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) O::go(a[c], b, cc);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < CH; ++c) s += a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <class K> int run(K kern, const char* name, float* d, long long* dc) {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("%-14s", name);
  for (int w : {1, 2, 4, 8}) {
    const int blocks = cus * w;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0f);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    const int reps = 3;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), dc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double wave_cyc = (double)h[h.size() / 2];
    const double n = (double)ITERS * CH;
    // shader cycles per instruction per SIMD (w waves interleave on one SIMD); wall-derived GHz for reference
    printf("  w%d: %5.2f cyc/inst/SIMD (%.2f GHz)", w, wave_cyc / n / w, wave_cyc / (ms * 1e-3) / 1e9);
  }
  printf("\n");
  return 0;
}

#define R32(O) run(k32<O>, O::nm(), d, dc)
#define R64(O) run(k64<O>, O::nm(), d, dc)

int main() {
  float* d; long long* dc;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&dc, 256 * 8 * 4 * sizeof(long long)));
  R32(fma_f32); R32(add_f32); R32(mul_f32); R32(min_f32); R32(max_f32); R32(min_i32); R32(min_u32);
  R32(min3_f32); R32(min3_i32); R32(med3_f32); R32(med3_i32); R32(minimum3);
  R32(and_or_b32); R32(and_or_imm); R32(bfi_b32); R32(perm_b32); R32(bitop3); R32(and_b32); R32(or_b32); R32(add_u32); R32(lshl_or);
  R32(mov_b32); R32(cndmask); R32(cmp_lt_f32); R32(pk_min_f16); R32(pk_add_f16); R32(cvt_pkrtz); R32(min_dpp); R32(mov_dpp); R32(sad_u32);
  R32(bitop3_imm); R32(sub_f32); R32(fmac_f32); R32(xor_b32); R32(lshlrev); R32(lshrrev); R32(ashrrev); R32(sub_u32); R32(add3_u32);
  R32(cvt_pk_bf16); R32(max_u32); R32(max_u16); R32(min_f16); R32(alignbit); R32(mad_u32_u24); R32(cvt_f32_i32); R32(xad_u32); R32(fma_f16); R32(pk_fma_f16);
  R64(fma_f64); R64(add_f64); R64(mul_f64); R64(min_f64); R64(pk_fma_f32); R64(pk_add_f32); R64(pk_mul_f32); R64(pk_mov_b32); R64(lshl_add_u64);
  return 0;
}
