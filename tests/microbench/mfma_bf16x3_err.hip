// mfma_bf16x3_err.hip — what does v_mfma_f32_16x16x32_bf16 do to the sum of the products it is given?
//
// The nearest-of-K search (svn-icp_amd/csrc/stein_split.hip, k_stein_search_bf16) computes the float32 score
//   S = cc + c . m        (c, m float32 3-vectors, cc float32)
// on the bf16 matrix pipe: every float32 operand is split EXACTLY into three bf16 pieces (a = a1 + a2 + a3) and the six
// products a_i b_j with i + j <= 4 of each component (three for cc) fill 21 of the 32 K slots; the other 11 slots and the
// accumulator input are zero.  The kernel's exactness certificate budgets the matrix pipe's accumulation error at
// 48·u·Σ|products| (u = 2^-24), the worst case of any faithfully-rounding or truncating adder tree — this program looks
// for inputs on which the hardware is worse than HALF of that, in two parts:
//   part 1  the kernel's own operand construction (split_a3 / split_b3) on random, cancelling and extreme-ratio inputs;
//           error against the exact sum of the 21 products that were fed (read back from the operand registers);
//   part 2  ARBITRARY bf16 values in the 21 live slots (a superset of what the split can produce): one large term with
//           twenty half-ulp / just-below-ulp terms, graded magnitudes with alternating signs, cancelling pairs, random
//           exponents over 40 binades — each sequence in every rotation over the live slots;
// and prints a few probes of the adder's behaviour (ties, sticky bits, truncation) for the record.
// Last line (parsed by tests/test_gpu_parity.py::test_mfma_bf16x3_error_budget):
//   RESULT split_max=<x> adversarial_max=<y> split_inexact=<n>       (x, y in units of u·Σ|products|)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_bf16x3_err mfma_bf16x3_err.hip && ./mfma_bf16x3_err
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdint>
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
// the kernel's operand construction, verbatim (stein_split.hip: split_a3 / split_b3)
__device__ __forceinline__ u4 split_a3(float x) {   // [a1 a1 | a1 a2 | a2 a3 | a2 a3]
  const unsigned r0 = pk(x, x);
  const float e1 = x - __uint_as_float(r0 & 0xffff0000u);
  const unsigned r1 = pk(x, e1);
  const float e2 = e1 - __uint_as_float(r1 & 0xffff0000u);
  const unsigned r2 = pk(e1, e2);
  return (u4){r0, r1, r2, r2};
}
__device__ __forceinline__ u4 split_b3(float x) {   // [b1 b2 | b3 b1 | b2 b1 | 0 0]
  const unsigned r0 = pk(x, x);
  const float e1 = x - __uint_as_float(r0 & 0xffff0000u);
  const unsigned q0 = pk(x, e1);
  const float e2 = e1 - __uint_as_float(q0 & 0xffff0000u);
  const unsigned q1 = pk(e2, x);
  const unsigned q2 = pk(e1, x);
  return (u4){q0, q1, q2, 0u};
}

// part 1: one wave per tile: c [16][4] (cx,cy,cz,cc), m [16][4] (mx,my,mz,·); out [16 rows][16 cols]; the operand
// registers of every lane are written back so that the host sums exactly the products the matrix pipe was given
__global__ __launch_bounds__(64) void k_split(const float4* __restrict__ c, const float4* __restrict__ m, float* __restrict__ out,
                                              u4* __restrict__ opa, u4* __restrict__ opb) {
  const int lane = threadIdx.x, g = lane >> 4, r = lane & 15;
  const float4 cr = c[(size_t)blockIdx.x * 16 + r], mr = m[(size_t)blockIdx.x * 16 + r];
  const float va = g == 0 ? cr.x : g == 1 ? cr.y : g == 2 ? cr.z : cr.w;
  const float vb = g == 0 ? mr.x : g == 1 ? mr.y : g == 2 ? mr.z : 1.0f;
  const u4 A = split_a3(va), B = split_b3(vb);
  opa[(size_t)blockIdx.x * 64 + lane] = A;
  opb[(size_t)blockIdx.x * 64 + lane] = B;
  const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
  const v4f d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, A), __builtin_bit_cast(bf8, B), zero, 0, 0, 0);
  float* o = out + (size_t)blockIdx.x * 256;
#pragma unroll
  for (int v = 0; v < 4; ++v) o[(4 * g + v) * 16 + r] = d[v];
}

// part 2: raw operands, A [tile][16 rows][32 k], B [tile][16 cols][32 k] as bf16 bit patterns
__global__ __launch_bounds__(64) void k_raw(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ out) {
  const int lane = threadIdx.x, g = lane >> 4, r = lane & 15;
  const uint16_t* ap = A + ((size_t)blockIdx.x * 16 + r) * 32 + 8 * g;
  const uint16_t* bp = B + ((size_t)blockIdx.x * 16 + r) * 32 + 8 * g;
  u4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = (unsigned)ap[2 * i] | ((unsigned)ap[2 * i + 1] << 16); b[i] = (unsigned)bp[2 * i] | ((unsigned)bp[2 * i + 1] << 16); }
  const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
  const v4f d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), zero, 0, 0, 0);
  float* o = out + (size_t)blockIdx.x * 256;
#pragma unroll
  for (int v = 0; v < 4; ++v) o[(4 * g + v) * 16 + r] = d[v];
}

static double bf16_to_double(uint16_t h) { uint32_t w = (uint32_t)h << 16; float f; memcpy(&f, &w, 4); return (double)f; }
static uint16_t double_to_bf16_exact(double v, bool* exact) {   // v must be representable (8 significant bits, bf16 exponent range)
  float f = (float)v; uint32_t w; memcpy(&w, &f, 4);
  if (exact) *exact = ((w & 0xffffu) == 0) && ((double)f == v);
  return (uint16_t)(w >> 16);
}
// the 21 live K slots of the kernel's layout: lane group g < 3 (k = 8g .. 8g+5), group 3 holds the cc row: its B operand
// is the split of 1.0 = [1 0 | 0 1 | 0 1 | 0 0], so slots 24, 27, 29 are live
static const int kLive[21] = {0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 16, 17, 18, 19, 20, 21, 24, 27, 29};

int main() {
  const double u = ldexp(1.0, -24);
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  double split_max = 0.0, adv_max = 0.0;
  long split_inexact = 0;

  // ---------------- part 1 ----------------
  {
    const int tiles = 1 << 15;
    std::vector<float4> hc((size_t)tiles * 16), hm((size_t)tiles * 16);
    std::vector<float> ho((size_t)tiles * 256);
    std::vector<u4> ha((size_t)tiles * 64), hb((size_t)tiles * 64);
    float4 *dc, *dm; float* dout; u4 *da, *db;
    CHECK(hipMalloc(&dc, hc.size() * sizeof(float4))); CHECK(hipMalloc(&dm, hm.size() * sizeof(float4)));
    CHECK(hipMalloc(&dout, ho.size() * sizeof(float))); CHECK(hipMalloc(&da, ha.size() * sizeof(u4))); CHECK(hipMalloc(&db, hb.size() * sizeof(u4)));
    struct Case { const char* name; double C, X; int near; } cases[] = {
        {"C=0.3 X=0.3", 0.3, 0.3, 0}, {"C=0.3 X=0.3 near (x~c)", 0.3, 0.3, 1}, {"C=1e-3 X=1e-3", 1e-3, 1e-3, 0},
        {"C=0.05 X=2.0", 0.05, 2.0, 0}, {"C=2.0 X=0.05", 2.0, 0.05, 0}, {"C=300 X=300 near", 300.0, 300.0, 1},
        {"C=1e-5 X=1e3", 1e-5, 1e3, 0}, {"C=0.3 X=0.3 all-ones mantissas", 0.3, 0.3, 2}};
    for (const Case& cs : cases) {
      for (int t = 0; t < tiles; ++t) {
        for (int i = 0; i < 16; ++i) {
          float cx = (float)(cs.C * U(rng)), cy = (float)(cs.C * U(rng)), cz = (float)(cs.C * U(rng));
          if (cs.near == 2) {  // significands 0x7fffff / 0x7f7f7f: every bf16 piece rounds up and leaves a negative remainder
            auto ones = [&](float v, uint32_t pat) { uint32_t w; memcpy(&w, &v, 4); w = (w & 0xff800000u) | pat; memcpy(&v, &w, 4); return v; };
            cx = ones(cx, 0x7fffffu); cy = ones(cy, 0x7f7f7fu); cz = ones(cz, 0x7fff7fu);
          }
          const float cc = (float)(((double)cx * cx + (double)cy * cy) + (double)cz * cz);
          hc[(size_t)t * 16 + i] = make_float4(cx, cy, cz, cc);
        }
        for (int j = 0; j < 16; ++j) {
          float x0, x1, x2;
          if (cs.near == 1) {  // particle next to candidate j: heavy cancellation in the score
            const float4 q = hc[(size_t)t * 16 + j];
            x0 = q.x + (float)(1e-3 * cs.C * U(rng)); x1 = q.y + (float)(1e-3 * cs.C * U(rng)); x2 = q.z + (float)(1e-3 * cs.C * U(rng));
          } else { x0 = (float)(cs.X * U(rng)); x1 = (float)(cs.X * U(rng)); x2 = (float)(cs.X * U(rng)); }
          hm[(size_t)t * 16 + j] = make_float4(-2.0f * x0, -2.0f * x1, -2.0f * x2, 0.0f);
        }
      }
      CHECK(hipMemcpy(dc, hc.data(), hc.size() * sizeof(float4), hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dm, hm.data(), hm.size() * sizeof(float4), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_split, dim3(tiles), dim3(64), 0, 0, dc, dm, dout, da, db);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(ho.data(), dout, ho.size() * sizeof(float), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(ha.data(), da, ha.size() * sizeof(u4), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hb.data(), db, hb.size() * sizeof(u4), hipMemcpyDeviceToHost));
      double worst_fed = 0, worst_f32 = 0, worst_cx = 0; size_t n = 0; long inexact = 0;
      for (int t = 0; t < tiles; ++t) {
        // unpack the operands: A row i (lanes 16g + i), B column j (lanes 16g + j)
        double Ad[16][32], Bd[16][32];
        for (int l = 0; l < 64; ++l) {
          const int g = l >> 4, r = l & 15;
          const u4 a = ha[(size_t)t * 64 + l], b = hb[(size_t)t * 64 + l];
          for (int q = 0; q < 4; ++q) {
            Ad[r][8 * g + 2 * q] = bf16_to_double((uint16_t)(a[q] & 0xffffu)); Ad[r][8 * g + 2 * q + 1] = bf16_to_double((uint16_t)(a[q] >> 16));
            Bd[r][8 * g + 2 * q] = bf16_to_double((uint16_t)(b[q] & 0xffffu)); Bd[r][8 * g + 2 * q + 1] = bf16_to_double((uint16_t)(b[q] >> 16));
          }
        }
        for (int i = 0; i < 16; ++i) {   // the three pieces must add up to the float32 value exactly
          const float4 c = hc[(size_t)t * 16 + i];
          const float comp[4] = {c.x, c.y, c.z, c.w};
          for (int g = 0; g < 4; ++g)
            if (Ad[i][8 * g] + Ad[i][8 * g + 3] + Ad[i][8 * g + 5] != (double)comp[g]) ++inexact;
        }
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j) {
            double fed = 0, mag = 0;
            for (int k = 0; k < 32; ++k) { const double p = Ad[i][k] * Bd[j][k]; fed += p; mag += fabs(p); }
            const float4 c = hc[(size_t)t * 16 + i], m = hm[(size_t)t * 16 + j];
            const double exact = (double)c.w + ((double)c.x * m.x + (double)c.y * m.y + (double)c.z * m.z);
            const double got = (double)ho[(size_t)t * 256 + i * 16 + j];
            const double Cm = fmax(fabs(c.x), fmax(fabs(c.y), fabs(c.z))), Xm = 0.5 * fmax(fabs(m.x), fmax(fabs(m.y), fabs(m.z)));
            if (mag > 0) worst_fed = fmax(worst_fed, fabs(got - fed) / (u * mag));
            if (mag > 0) worst_f32 = fmax(worst_f32, fabs(got - exact) / (u * mag));
            worst_cx = fmax(worst_cx, fabs(got - exact) / (u * (Cm + Xm) * (Cm + Xm)));
            ++n;
          }
      }
      printf("%-32s split-not-exact %ld | accumulation error max %.3f u*sum|products fed| ; against the float32 score (incl. dropped products) "
             "%.3f u*sum|products| = %.3f u*(C+X)^2 ; %zu scores\n", cs.name, inexact, worst_fed, worst_f32, worst_cx, n);
      split_max = fmax(split_max, worst_fed);
      split_inexact += inexact;
    }
    (void)hipFree(dc); (void)hipFree(dm); (void)hipFree(dout); (void)hipFree(da); (void)hipFree(db);
  }

  // ---------------- part 2 ----------------
  {
    // a sequence = 21 term values t_0..t_20, each an exact product a·b of two bf16 numbers; placed in the live slots in
    // every rotation.  Row i of a tile = rotation (16·tile_in_family + i) mod 21 of the family's sequence; column j uses
    // b-factors scaled by 2^-(j mod 4) (exact), so 16 x 16 distinct sums per tile.
    struct Seq { const char* name; std::vector<double> a, b; };
    std::vector<Seq> seqs;
    auto p2 = [](int e) { return ldexp(1.0, e); };
    {  // one large term and twenty exact half-ulps of it
      Seq s{"1 + 20 x 2^-24 (half-ulp ties)", {}, {}};
      s.a.push_back(1.0); s.b.push_back(1.0);
      for (int i = 0; i < 20; ++i) { s.a.push_back(p2(-12)); s.b.push_back(p2(-12)); }
      seqs.push_back(s);
    }
    {  // … and twenty terms just below one ulp: a truncating sequential adder loses all of them
      Seq s{"1 + 20 x 0.996*2^-23 (just below one ulp)", {}, {}};
      s.a.push_back(1.0); s.b.push_back(1.0);
      for (int i = 0; i < 20; ++i) { s.a.push_back(p2(-12) * (255.0 / 128.0)); s.b.push_back(p2(-12) * (255.0 / 256.0) * 1.0); }
      seqs.push_back(s);
    }
    {  // the same with the large term negative and significand all ones
      Seq s{"-(2-2^-7)^2 + 20 x small positive", {}, {}};
      s.a.push_back(-(2.0 - p2(-7))); s.b.push_back(2.0 - p2(-7));
      for (int i = 0; i < 20; ++i) { s.a.push_back(p2(-11) * (1.0 + (i % 7) / 8.0)); s.b.push_back(p2(-11) * (1.0 + (i % 5) / 4.0)); }
      seqs.push_back(s);
    }
    {  // graded magnitudes, alternating signs: every addition in a sequential order rounds
      Seq s{"graded 2^-k, alternating signs, full significands", {}, {}};
      for (int i = 0; i < 21; ++i) { s.a.push_back((i & 1 ? -1.0 : 1.0) * p2(-i) * (255.0 / 128.0)); s.b.push_back(255.0 / 256.0 + 0.0); }
      seqs.push_back(s);
    }
    {  // graded by 2^-2 steps (three terms per 8-bit window, like the a_i b_j pieces of the kernel)
      Seq s{"graded 2^-8 groups (the kernel's piece structure)", {}, {}};
      const int ea[6] = {0, -8, -16, -8, -16, -16};
      for (int c = 0; c < 3; ++c) for (int q = 0; q < 6; ++q) { s.a.push_back((q & 1 ? -1.0 : 1.0) * p2(ea[q]) * (1.0 + (37 * (c * 6 + q) % 128) / 128.0)); s.b.push_back(1.0 + (91 * (c * 6 + q) % 128) / 128.0); }
      for (int q = 0; q < 3; ++q) { s.a.push_back(-p2(-8 * q) * (1.0 + (53 * q % 128) / 128.0)); s.b.push_back(1.0); }
      seqs.push_back(s);
    }
    {  // cancelling pairs: +x, -x(1 - 2^-7), then small leftovers
      Seq s{"cancelling pairs", {}, {}};
      for (int i = 0; i < 10; ++i) { const double x = p2(-(i % 3)) * (1.0 + (i * 29 % 128) / 128.0); s.a.push_back(x); s.b.push_back(1.5); s.a.push_back(-x); s.b.push_back(1.5 - p2(-7)); }
      s.a.push_back(p2(-20)); s.b.push_back(1.0);
      seqs.push_back(s);
    }
    for (int rep = 0; rep < 64; ++rep) {  // random signs, exponents over 40 binades, full random significands
      Seq s{"random exponents over 40 binades", {}, {}};
      for (int i = 0; i < 21; ++i) {
        const double sa = (rng() & 1) ? -1.0 : 1.0;
        s.a.push_back(sa * p2(-(int)(rng() % 20)) * (1.0 + (double)(rng() % 128) / 128.0));
        s.b.push_back(p2(-(int)(rng() % 20)) * (1.0 + (double)(rng() % 128) / 128.0));
      }
      seqs.push_back(s);
    }
    const int tiles_per_seq = 2;   // 2 x 16 rows >= 21 rotations
    const int tiles = (int)seqs.size() * tiles_per_seq;
    std::vector<uint16_t> hA((size_t)tiles * 16 * 32, 0), hB((size_t)tiles * 16 * 32, 0);
    std::vector<float> ho((size_t)tiles * 256);
    bool all_exact = true;
    for (size_t si = 0; si < seqs.size(); ++si)
      for (int tt = 0; tt < tiles_per_seq; ++tt) {
        const size_t t = si * tiles_per_seq + tt;
        for (int i = 0; i < 16; ++i) {
          const int rot = (16 * tt + i) % 21;
          for (int q = 0; q < 21; ++q) {
            bool ex;
            hA[(t * 16 + i) * 32 + kLive[(q + rot) % 21]] = double_to_bf16_exact(seqs[si].a[q], &ex); all_exact &= ex;
          }
        }
        for (int j = 0; j < 16; ++j) {
          // the b-factor of term q sits in the same slot as its a-factor only for ONE rotation; to keep every (row, column)
          // sum a sum of the family's products, column j carries b-factors in rotation (16·tt + j) mod 21 as well and rows
          // and columns are compared on the diagonal of rotations only (i == j) — off-diagonal entries are still valid
          // 21-term sums of products of the family's factors and are checked too
          const int rot = (16 * tt + j) % 21;
          for (int q = 0; q < 21; ++q) {
            bool ex;
            hB[(t * 16 + j) * 32 + kLive[(q + rot) % 21]] = double_to_bf16_exact(seqs[si].b[q] * ldexp(1.0, -(j % 4)), &ex); all_exact &= ex;
          }
        }
      }
    if (!all_exact) { printf("internal error: an adversarial factor is not a bf16 number\n"); return 2; }
    uint16_t *dA, *dB; float* dout;
    CHECK(hipMalloc(&dA, hA.size() * 2)); CHECK(hipMalloc(&dB, hB.size() * 2)); CHECK(hipMalloc(&dout, ho.size() * sizeof(float)));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_raw, dim3(tiles), dim3(64), 0, 0, dA, dB, dout);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ho.data(), dout, ho.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<double> fam_max(seqs.size(), 0.0);
    for (size_t t = 0; t < (size_t)tiles; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double fed = 0, mag = 0;
          for (int k = 0; k < 32; ++k) { const double p = bf16_to_double(hA[(t * 16 + i) * 32 + k]) * bf16_to_double(hB[(t * 16 + j) * 32 + k]); fed += p; mag += fabs(p); }
          if (mag == 0) continue;
          const double r = fabs((double)ho[t * 256 + i * 16 + j] - fed) / (u * mag);
          fam_max[t / tiles_per_seq] = fmax(fam_max[t / tiles_per_seq], r);
        }
    double rnd = 0;
    for (size_t si = 0; si < seqs.size(); ++si) {
      adv_max = fmax(adv_max, fam_max[si]);
      if (si + 64 < seqs.size()) printf("adversarial: %-52s max error %.3f u*sum|products|\n", seqs[si].name, fam_max[si]);
      else rnd = fmax(rnd, fam_max[si]);
    }
    printf("adversarial: %-52s max error %.3f u*sum|products| (64 sequences)\n", "random exponents over 40 binades", rnd);

    // probes: what the adder does with ties, sticky bits and truncation (one row each, B = ones in the live slots)
    struct Probe { const char* name; std::vector<double> t; };
    std::vector<Probe> probes = {
        {"1 + 2^-24                     (tie)", {1.0, p2(-24)}},
        {"1 + 2^-24 + 2^-24             (two half-ulps: exact sum 1 + 2^-23)", {1.0, p2(-24), p2(-24)}},
        {"1 + 2^-24 + 2^-25             (0.75 ulp)", {1.0, p2(-24), p2(-25)}},
        {"1 + 2^-24 + 2^-40             (tie + sticky)", {1.0, p2(-24), p2(-40)}},
        {"1 - 2^-25                     (tie below one)", {1.0, -p2(-25)}},
        {"1 - 2^-26                     (quarter-ulp below one)", {1.0, -p2(-26)}},
        {"1 + 16 x 2^-28                (sixteen 1/16-ulps: exact sum 1 + 2^-24)", {1.0, p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28), p2(-28)}},
        {"1 + 20 x 2^-27                (exact sum 1 + 2.5 x 2^-24)", {1.0, p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27), p2(-27)}},
        {"2^20 + 1 + 1 + 1 - 2^20       (cancellation after absorption)", {p2(20), 1.0, 1.0, 1.0, -p2(20)}},
        {"2^30 + 1 - 2^30               (needs > 24 bits)", {p2(30), 1.0, -p2(30)}},
    };
    std::vector<uint16_t> pA((size_t)16 * 32, 0), pB((size_t)16 * 32, 0);
    for (size_t r = 0; r < probes.size() && r < 16; ++r)
      for (size_t q = 0; q < probes[r].t.size(); ++q) pA[r * 32 + kLive[q]] = double_to_bf16_exact(probes[r].t[q], nullptr);
    for (int j = 0; j < 16; ++j) for (int q = 0; q < 21; ++q) pB[(size_t)j * 32 + kLive[q]] = double_to_bf16_exact(1.0, nullptr);
    CHECK(hipMemcpy(dA, pA.data(), pA.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, pB.data(), pB.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_raw, dim3(1), dim3(64), 0, 0, dA, dB, dout);
    CHECK(hipDeviceSynchronize());
    float po[256];
    CHECK(hipMemcpy(po, dout, sizeof po, hipMemcpyDeviceToHost));
    for (size_t r = 0; r < probes.size() && r < 16; ++r) {
      double ex = 0; for (double v : probes[r].t) ex += v;
      printf("probe: %-70s -> %.10g  (exact %.10g, difference %+.3f u)\n", probes[r].name, (double)po[r * 16], ex, ((double)po[r * 16] - ex) / u);
    }
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dout);
  }
  printf("RESULT split_max=%.4f adversarial_max=%.4f split_inexact=%ld\n", split_max, adv_max, split_inexact);
  return 0;
}
