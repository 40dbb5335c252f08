// search_loop.hip — what one "tile" (16 candidates x 16 particles -> 4 scores per lane) of the stage-B
// search loop costs on gfx950, by scoring pipe and tracking idiom.  Calibrates the design choices
// in DESIGN.md §4.2.  Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -o search_loop search_loop.hip && ./search_loop
// Output: SIMD cycles (s_memtime) per 4 scores per lane, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int STEPS = 256;     // wave steps per launch
constexpr int TILES = 24;      // tiles per step (6 row blocks x 4 column blocks)

__device__ __forceinline__ float pack_slot(float v, unsigned int mask, unsigned int bits) {
  return __uint_as_float((__float_as_uint(v) & ~mask) | bits);
}
__device__ __forceinline__ float imin_f(float a, float b) {
  const int x = (int)__float_as_uint(a), y = (int)__float_as_uint(b);
  return __uint_as_float((unsigned int)(x < y ? x : y));
}
__device__ __forceinline__ float imin3_f(float a, float b, float c) {
  int x = (int)__float_as_uint(a), y = (int)__float_as_uint(b), z = (int)__float_as_uint(c);
  int r;
  asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return __uint_as_float((unsigned int)r);
}

// TRACK: 0 none (keep alive), 1 pack+med3+min (3/score), 2 pack+min (2/score), 3 pack + med3 + min3 on pairs (2.5/score),
//        4 min3 on raw pairs only (0.5/score), 5 med3+min no pack (2/score)
template <int TRACK>
__device__ __forceinline__ void track4(v4f d, int rb, float& b1, float& b2) {
  if constexpr (TRACK == 0) {
    asm volatile("" :: "v"(d));
  } else if constexpr (TRACK == 1) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float pk = pack_slot(d[v], 0x1fu, (unsigned int)(rb * 4 + v));
      b2 = __builtin_amdgcn_fmed3f(b1, b2, pk);
      b1 = imin_f(b1, pk);
    }
  } else if constexpr (TRACK == 2) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float pk = pack_slot(d[v], 0x1fu, (unsigned int)(rb * 4 + v));
      b1 = imin_f(b1, pk);
    }
  } else if constexpr (TRACK == 3) {
    float pk[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) pk[v] = pack_slot(d[v], 0x1fu, (unsigned int)(rb * 4 + v));
    // second-min: med3 per score against the OLD min is not valid across a pair; use the two-sorted-pairs form
#pragma unroll
    for (int v = 0; v < 4; v += 2) {
      const float lo = imin_f(pk[v], pk[v + 1]);
      const float hi = __uint_as_float((unsigned int)max((int)__float_as_uint(pk[v]), (int)__float_as_uint(pk[v + 1])));
      const float mx = __uint_as_float((unsigned int)max((int)__float_as_uint(b1), (int)__float_as_uint(lo)));
      b2 = imin3_f(mx, b2, hi);
      b1 = imin_f(b1, lo);
    }
  } else if constexpr (TRACK == 4) {
    b1 = imin3_f(b1, d[0], d[1]);
    b1 = imin3_f(b1, d[2], d[3]);
  } else if constexpr (TRACK == 5) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      b2 = __builtin_amdgcn_fmed3f(b1, b2, d[v]);
      b1 = imin_f(b1, d[v]);
    }
  }
}

// PIPE: 0 none (scores from registers), 1 v_mfma_f32_16x16x4_f32, 2 v_mfma_f32_16x16x32_f16, 3 v_mfma_f32_32x32x16_f16
//       (one instruction = 16 scores per lane = 4 "tiles"), 4 VALU fma (3 per score)
template <int PIPE, int TRACK>
__global__ __launch_bounds__(256, 4) void k(float* out, long long* cyc, float seed) {
  const int lane = threadIdx.x & 63;
  float av[6]; float bvv[4], bee[4];
  h8 ah[6], bh[4];
  for (int i = 0; i < 6; ++i) { av[i] = seed * (lane + i + 1); for (int j = 0; j < 8; ++j) ah[i][j] = (_Float16)(seed * (lane + i + j)); }
  for (int i = 0; i < 4; ++i) { bvv[i] = seed * (2 * lane + i); bee[i] = 3.0f + i; for (int j = 0; j < 8; ++j) bh[i][j] = (_Float16)(seed * (lane - i + j)); }
  float b1[4], b2[4];
  for (int i = 0; i < 4; ++i) { b1[i] = 1e30f; b2[i] = 1e30f; }
  v4f sc[4];
  for (int i = 0; i < 4; ++i) sc[i] = v4f{seed + i, seed * 2 + i, seed * 3 + i, seed * 4 + i};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < STEPS; ++s) {
    if constexpr (PIPE == 0) {
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        asm volatile("" : "+v"(sc[i & 3]));
        track4<TRACK>(sc[i & 3], i % 6, b1[i / 6], b2[i / 6]);
      }
    } else if constexpr (PIPE == 1) {
      auto tile = [&](int i) -> v4f {
        const int cb = i / 6, rb = i % 6;
        const v4f cin = {bee[cb], bee[cb], bee[cb], bee[cb]};
        return __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb], bvv[cb], cin, 0, 0, 0);
      };
      v4f dcur = tile(0);
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        v4f dnext = dcur;
        if (i + 1 < TILES) dnext = tile(i + 1);
        track4<TRACK>(dcur, i % 6, b1[i / 6], b2[i / 6]);
        dcur = dnext;
      }
      asm volatile("" : "+v"(av[0]), "+v"(bvv[0]));
    } else if constexpr (PIPE == 2) {
      auto tile = [&](int i) -> v4f {
        const int cb = i / 6, rb = i % 6;
        const v4f cin = {bee[cb], bee[cb], bee[cb], bee[cb]};
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rb], bh[cb], cin, 0, 0, 0);
      };
      v4f dcur = tile(0);
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        v4f dnext = dcur;
        if (i + 1 < TILES) dnext = tile(i + 1);
        track4<TRACK>(dcur, i % 6, b1[i / 6], b2[i / 6]);
        dcur = dnext;
      }
      asm volatile("" : "+v"(ah[0]), "+v"(bh[0]));
    } else if constexpr (PIPE == 3) {
      // 6 big tiles of 32 candidates x 32 particles = 16 scores per lane each (same 96 scores per lane per step)
      auto tile = [&](int i) -> v16f {
        v16f cin;
#pragma unroll
        for (int j = 0; j < 16; ++j) cin[j] = bee[i & 3];
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[i & 3], cin, 0, 0, 0);
      };
      v16f dcur = tile(0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        v16f dnext = dcur;
        if (i + 1 < 6) dnext = tile(i + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f d = {dcur[4 * q], dcur[4 * q + 1], dcur[4 * q + 2], dcur[4 * q + 3]};
          track4<TRACK>(d, q, b1[i & 3], b2[i & 3]);
        }
        dcur = dnext;
      }
      asm volatile("" : "+v"(ah[0]), "+v"(bh[0]));
    } else if constexpr (PIPE == 4) {
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        const int cb = i / 6, rb = i % 6;
        v4f d;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          asm volatile("" : "+v"(sc[v]));
          d[v] = __builtin_fmaf(sc[v][0], bvv[cb], __builtin_fmaf(sc[v][1], av[rb], __builtin_fmaf(sc[v][2], bee[cb], sc[v][3])));
        }
        track4<TRACK>(d, rb, b1[cb], b2[cb]);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 4; ++i) r += b1[i] + b2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int PIPE, int TRACK> int run(const char* name, float* d, long long* dc) {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("%-34s", name);
  for (int w : {1, 2, 4, 5, 8}) {
    const int blocks = cus * w;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<PIPE, TRACK>), dim3(blocks), dim3(256), 0, 0, d, dc, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 3;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<PIPE, TRACK>), dim3(blocks), dim3(256), 0, 0, d, dc, 1.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), dc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double wave_cyc = (double)h[h.size() / 2];          // median wave lifetime in shader cycles
    const double tiles_per_simd = (double)w * STEPS * TILES;  // w waves share one SIMD
    // cycles per tile per SIMD from the wave clock (all w waves run concurrently for ~wave_cyc), and from wall time @2.4 GHz
    printf("  w%d: %6.1f cyc (wall %6.1f)", w, wave_cyc / (STEPS * TILES) , ms * 1e-3 * 2.4e9 / tiles_per_simd);
  }
  printf("\n");
  return 0;
}

int main() {
  float* d; long long* dc;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&dc, 256 * 8 * 4 * sizeof(long long)));
  printf("columns: waves/SIMD; first number = median wave lifetime / tiles issued by that wave (divide by w for per-SIMD throughput), wall = SIMD cycles per tile at 2.4 GHz\n");
  run<0, 1>("regs   + pack/med3/min (3/score)", d, dc);
  run<0, 2>("regs   + pack/min (2/score)", d, dc);
  run<0, 4>("regs   + min3 raw (0.5/score)", d, dc);
  run<0, 5>("regs   + med3/min nopack (2/score)", d, dc);
  run<1, 0>("mfma f32 16x16x4 only", d, dc);
  run<1, 1>("mfma f32 16x16x4 + 3/score", d, dc);
  run<2, 0>("mfma f16 16x16x32 only", d, dc);
  run<2, 1>("mfma f16 16x16x32 + 3/score", d, dc);
  run<2, 2>("mfma f16 16x16x32 + 2/score", d, dc);
  run<3, 0>("mfma f16 32x32x16 only", d, dc);
  run<3, 1>("mfma f16 32x32x16 + 3/score", d, dc);
  run<4, 0>("valu fma only (3/score)", d, dc);
  run<4, 1>("valu fma + 3/score", d, dc);
  return 0;
}
