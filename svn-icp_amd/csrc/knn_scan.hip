// knn_scan.hip — Stage A, fast variant: exact brute-force top-K with an f32 pre-filter.
//
// Same result contract as knn_topk.hip (reference: SVGDICP::knn_source_cloud, src/core/SVGDICP.cpp:201-215;
// CPU semantics src/core/knn/knn_cpu.cpp:35-67): K smallest by (f64 dist², idx), ascending, with
// dist² = ((dx·dx)+dy·dy)+dz·dz unfused.  Every (query, target) pair is still visited — this is
// brute force — but the per-pair work is 6 f32 VALU ops + 1 compare, and f64 arithmetic is spent
// only on the few hundred survivors per query:
//
//  1. seed   For each query the wave scans the first Ms slots of the (golden-ratio permuted)
//            target stream keeping one running f32 minimum per (query, lane) in LDS.  The j-th
//            smallest of a query's 64 lane minima becomes its threshold τ: with Ms/Mp = 1/8 and
//            j = 32 about 3·K targets lie below τ, and fewer than K do with probability ~1e-6.
//  2. scan   Fixed τ, all Mp slots: d32 = fma(dz,dz,fma(dy,dy,dx·dx)) on float32 copies against
//            τ32 = an upper bound of d32 over every pair whose TRUE distance is <= τ
//            (derivation below), so the filter never loses a pair with d² <= τ.  Survivors are
//            (query, slot) words appended to a wave-level LDS queue (ballot/mbcnt, counter in an
//            SGPR: no per-query state is touched in the hot loop) and flushed in batches into
//            per-query slot pools in HBM.
//  3. select Per query: exact f64 d² for its pool (gathers from the L2-resident target), count
//            those with d² <= τ.  If that count is >= K the true top-K is inside the pool (every
//            pair with d² <= τ passed the filter), the wave bitonic-sorts the survivors by
//            (d², idx) in LDS and writes the K best.  Otherwise (count < K or pool overflow) the
//            query is appended to a fail list and redone by the streaming kernel of knn_topk.hip —
//            correctness never depends on the seed being good, only speed does.
//
// Filter bound.  u = 2^-24, E >= every |coordinate| involved.  q~, t~ = float32 roundings:
// |δ~_i| <= (|δ_i| + 2uE)(1+u) per axis, so Σδ~² <= (1+u)²(‖δ‖ + 2√3·uE)², and the three rounded
// f32 operations add (1+u)³:  d32 <= (1+u)^5 (√d_real + 2√3·uE)².  The f64-computed d² is within
// (1 ± 2^-50) of d_real.  τ32 below uses those constants with an extra 1e-6 relative slack and is
// rounded up to float32.  Underflow only makes d32 smaller; overflow needs a huge true distance.
#include "kernels.hpp"

namespace svnicp {

namespace {

constexpr int T = 8;              // f32 target points per lane per step
constexpr int STEP = kWave * T;   // 512 slots per wave step  (Mp is a multiple of 512 for this kernel)
constexpr int QW = 64;            // queries per wave
constexpr int WAVES = 4;
constexpr int QCAP = 2048;        // wave queue capacity (entries of 8 B)

struct alignas(16) QF { float x, y, z, thr; };          // f32 query + filter threshold
struct alignas(16) QD { double x, y, z, tau; };         // f64 query + exact threshold

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool ent_less(double da, int ia, double db, int ib) {
  return (da < db) || (da == db && ia < ib);
}

__device__ void bitonic_sort(double* sd, int* si, int S, int lane) {
  for (int k = 2; k <= S; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = lane; e < (S >> 1); e += kWave) {
        const int a = ((e & ~(j - 1)) << 1) | (e & (j - 1));
        const int b = a | j;
        const bool up = (a & k) == 0;
        const double da = sd[a], db = sd[b];
        const int ia = si[a], ib = si[b];
        if (ent_less(db, ib, da, ia) == up) { sd[a] = db; si[a] = ib; sd[b] = da; si[b] = ia; }
      }
      wave_sync();
    }
  }
}

__device__ __forceinline__ float f32_round_up(double v) {
  float f = (float)v;                       // round to nearest
  if ((double)f < v) f = __int_as_float(__float_as_int(f) + 1);  // v >= 0 here: next float up
  return f;
}

// conservative f32 filter threshold for exact threshold tau (see header)
__device__ __forceinline__ float filter_threshold(double tau, double E) {
  if (!(tau < __builtin_huge_val())) return __builtin_huge_valf();
  const double u = 5.9604644775390625e-08;  // 2^-24
  const double r = sqrt(tau) * (1.0 + 1e-15) + 3.4641016151377553 * u * E;
  const double b = (r * r) * (1.0 + 5.0 * u + 1e-6);
  return f32_round_up(b);
}

__global__ __launch_bounds__(256, 2) void k_knn_scan(KnnScanArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps q0/nq/loop control scalar
  const int K = a.K, S2 = a.S2;
  // per-wave LDS: QD[64] | QF[64] | cnt[64] | scratch (seed minima / queue / sort arrays)
  constexpr size_t kScratch = 16384;  // >= 64*64*4 (seed), QCAP*8 (queue), S2max*12 (sort, S2 <= 1024)
  constexpr size_t per_wave = sizeof(QD) * QW + sizeof(QF) * QW + sizeof(int) * QW + kScratch;
  unsigned char* base = smem + per_wave * wave;
  QD* qd = reinterpret_cast<QD*>(base);
  QF* qf = reinterpret_cast<QF*>(base + sizeof(QD) * QW);
  int* cnt = reinterpret_cast<int*>(base + (sizeof(QD) + sizeof(QF)) * QW);
  unsigned char* scratch = base + (sizeof(QD) + sizeof(QF) + sizeof(int)) * QW;
  float* lmin = reinterpret_cast<float*>(scratch);              // [QW][64]
  int2* queue = reinterpret_cast<int2*>(scratch);               // [QCAP] {slot, q}
  double* sd = reinterpret_cast<double*>(scratch);              // [S2]
  int* si = reinterpret_cast<int*>(scratch + sizeof(double) * (size_t)S2);

  const int64_t q0 = a.b_lo + ((int64_t)blockIdx.x * WAVES + wave) * QW;
  if (q0 >= a.b_hi) return;  // no block-level barriers in this kernel
  const int nq = (a.b_hi - q0) < QW ? (int)(a.b_hi - q0) : QW;
  const double Et = __longlong_as_double((long long)*a.emax_bits);  // max |target coordinate|

  {  // queries: q = R0·s + t0 (SVGDICP.cpp:204), f64 and f32 copies
    const int64_t b = q0 + lane;
    QD s;
    if (b < a.b_hi) {
      const double sx = a.src[3 * b], sy = a.src[3 * b + 1], sz = a.src[3 * b + 2];
      const double* R = a.pose.R0;
      s.x = (sx * R[0] + sy * R[1] + sz * R[2]) + a.pose.t0[0];
      s.y = (sx * R[3] + sy * R[4] + sz * R[5]) + a.pose.t0[1];
      s.z = (sx * R[6] + sy * R[7] + sz * R[8]) + a.pose.t0[2];
    } else { s.x = s.y = s.z = 0.0; }
    s.tau = 0.0;
    qd[lane] = s;
    QF f; f.x = (float)s.x; f.y = (float)s.y; f.z = (float)s.z; f.thr = -1.0f;
    qf[lane] = f;
    cnt[lane] = 0;
  }
  wave_sync();

  // ------------------------------------------------------------------ 1. seed
  {
    for (int e = lane; e < QW * kWave; e += kWave) lmin[e] = __builtin_huge_valf();
    wave_sync();
    float x[T], y[T], z[T];
    for (int64_t tile = 0; tile < a.Ms; tile += STEP) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        x[t] = a.txf[tile + t * kWave + lane]; y[t] = a.tyf[tile + t * kWave + lane]; z[t] = a.tzf[tile + t * kWave + lane];
      }
      for (int q = 0; q < nq; ++q) {
        const float4 qq = *reinterpret_cast<const float4*>(&qf[q]);
        float m = lmin[q * kWave + lane];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float dx = qq.x - x[t], dy = qq.y - y[t], dz = qq.z - z[t];
          const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          m = __builtin_fminf(m, d);  // NaN padding is ignored by fmin
        }
        lmin[q * kWave + lane] = m;
      }
    }
    wave_sync();
    // j-th smallest of each query's 64 lane minima -> tau
    for (int q = 0; q < nq; ++q) {
      const float v = lmin[q * kWave + lane];
      int rank = 0;
      for (int m = 0; m < kWave; ++m) {
        const float o = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), m));
        rank += (o < v || (o == v && m < lane)) ? 1 : 0;
      }
      const unsigned long long hit = __ballot(rank == a.seed_rank);
      const int src_lane = hit ? (int)__builtin_ctzll(hit) : 0;
      const float tv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
      if (lane == 0) {
        const double tau = (double)tv;  // +inf when fewer than seed_rank+1 lanes saw a point
        const double qa = fmax(fabs(qd[q].x), fmax(fabs(qd[q].y), fabs(qd[q].z)));
        qd[q].tau = tau;
        qf[q].thr = filter_threshold(tau, fmax(Et, qa));
      }
    }
    wave_sync();
  }

  // ------------------------------------------------------------------ 2. scan
  int qcount = 0;  // wave-uniform
  auto flush = [&]() {
    wave_sync();
    for (int e = lane; e < qcount; e += kWave) {
      const int2 ent = queue[e];
      const int pos = atomicAdd(&cnt[ent.y], 1);
      if (pos < S2) a.pool[(q0 + ent.y) * (int64_t)S2 + pos] = ent.x;
    }
    qcount = 0;
    wave_sync();
  };
  {
    float x[T], y[T], z[T], nx[T], ny[T], nz[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { x[t] = a.txf[t * kWave + lane]; y[t] = a.tyf[t * kWave + lane]; z[t] = a.tzf[t * kWave + lane]; }
    for (int64_t tile = 0; tile < a.Mp; tile += STEP) {
      const int64_t nt = (tile + STEP < a.Mp) ? tile + STEP : tile;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        nx[t] = a.txf[nt + t * kWave + lane]; ny[t] = a.tyf[nt + t * kWave + lane]; nz[t] = a.tzf[nt + t * kWave + lane];
      }
      float4 qq = *reinterpret_cast<const float4*>(&qf[0]);
      for (int q = 0; q < nq; ++q) {
        const float4 cur = qq;
        qq = *reinterpret_cast<const float4*>(&qf[(q + 1 < nq) ? q + 1 : q]);  // prefetch next query (hides LDS latency)
        float d[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float dx = cur.x - x[t], dy = cur.y - y[t], dz = cur.z - z[t];
          d[t] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        }
        // one compare per step: a VALU→SGPR ballot per target costs more than the distance itself
        // (measured, tests/microbench/valu_rates.hip); fmin ignores the NaN padding
        float dm = __builtin_fminf(__builtin_fminf(d[0], d[1]), d[2]);
#pragma unroll
        for (int t = 3; t + 1 < T; t += 2) dm = __builtin_fminf(__builtin_fminf(dm, d[t]), d[t + 1]);
        if constexpr ((T - 3) % 2 == 1) dm = __builtin_fminf(dm, d[T - 1]);
        if (__ballot(dm <= cur.w)) {  // wave-uniform, rare
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const unsigned long long m = __ballot(d[t] <= cur.w);
            if (m) {
              const unsigned int lo = (unsigned int)m, hi = (unsigned int)(m >> 32);
              const int pos = qcount + (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
              if ((m >> lane) & 1ull) queue[pos] = make_int2((int)(tile + t * kWave + lane), q);
              qcount += __popcll(m);
            }
          }
          qcount = __builtin_amdgcn_readfirstlane(qcount);
          if (qcount > QCAP - STEP) flush();
        }
      }
#pragma unroll
      for (int t = 0; t < T; ++t) { x[t] = nx[t]; y[t] = ny[t]; z[t] = nz[t]; }
    }
    flush();
  }

  // ------------------------------------------------------------------ 3. select
  for (int q = 0; q < nq; ++q) {
    const int64_t b = q0 + q;
    const int n = __builtin_amdgcn_readfirstlane(cnt[q]);
    const double qx = qd[q].x, qy = qd[q].y, qz = qd[q].z, tau = qd[q].tau;
    bool ok = n <= S2;
    int m = 0;  // survivors with exact d2 <= tau, compacted into sd/si
    if (ok) {
      for (int e0 = 0; e0 < n; e0 += kWave) {
        const int e = e0 + lane;
        bool pass = false;
        double d = 0.0;
        int orig = 0;
        if (e < n) {
          const int slot = a.pool[b * (int64_t)S2 + e];
          orig = a.torig[slot];
          const double dx = qx - a.tx[slot], dy = qy - a.ty[slot], dz = qz - a.tz[slot];
          d = (dx * dx + dy * dy) + dz * dz;  // knn_cpu.cpp:43-50 order, unfused
          pass = d <= tau;
        }
        const unsigned long long pm = __ballot(pass);
        if (pass) {
          const int pos = m + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)pm, 0u));
          sd[pos] = d; si[pos] = orig;
        }
        m += __popcll(pm);
      }
      ok = m >= K;
    }
    if (!ok) {  // bad seed or overflow: redo this query with the streaming kernel
      if (lane == 0) {
        const int slot = atomicAdd(a.fail_count, 1);
        a.fail_list[slot] = (int32_t)b;
      }
      wave_sync();
      continue;
    }
    int S = 128;
    while (S < m) S <<= 1;
    for (int e = m + lane; e < S; e += kWave) { sd[e] = __builtin_huge_val(); si[e] = 0x7fffffff; }
    wave_sync();
    bitonic_sort(sd, si, S, lane);
    for (int e = lane; e < K; e += kWave) { a.out_idx[b * K + e] = si[e]; a.out_d2[b * K + e] = sd[e]; }
    wave_sync();
  }
}

// pseudo-random bijection of [0, Mp): three multiply/xorshift rounds on `bits` = ceil(log2 Mp) bits
// (each round is invertible mod 2^bits), cycle-walked back into range.
__device__ __forceinline__ unsigned long long slot_of(unsigned long long i, unsigned long long Mp, int bits) {
  const unsigned long long mask = (bits >= 64) ? ~0ull : ((1ull << bits) - 1ull);
  const int sh = (bits + 1) / 2;
  unsigned long long x = i;
  do {
    x = (x * 0x9E3779B97F4A7C15ull) & mask; x ^= x >> sh;
    x = (x * 0xBF58476D1CE4E5B9ull) & mask; x ^= x >> sh;
    x = (x * 0x94D049BB133111EBull) & mask; x ^= x >> sh;
  } while (x >= Mp);
  return x;
}

// f64 permuted SoA + f32 copies + original index + max |coordinate| (as uint64 bits of a
// non-negative double).  Scatter through a pseudo-random bijection: a LiDAR point's spatial
// neighbours sit at structured index offsets (±1 beam, ±64·k columns), and any linear map
// i → i·G mod Mp keeps that structure in the lane id (slot mod 64) — measured: 99.8 % of the seeded
// thresholds were useless.  With a hash every tile, every lane's subsequence and every prefix of
// the slot stream is a uniform random sample of every neighbourhood.
__global__ void k_targets_soa2(const double* __restrict__ tgt, int64_t M, int64_t Mp, int bits,
                               double* __restrict__ tx, double* __restrict__ ty, double* __restrict__ tz,
                               float* __restrict__ txf, float* __restrict__ tyf, float* __restrict__ tzf,
                               int32_t* __restrict__ torig, unsigned long long* __restrict__ emax_bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double e = 0.0;
  if (i < Mp) {
    const int64_t j = (int64_t)slot_of((unsigned long long)i, (unsigned long long)Mp, bits);
    const double nan = __builtin_nan("");
    const bool in = i < M;
    const double x = in ? tgt[3 * i] : nan, y = in ? tgt[3 * i + 1] : nan, z = in ? tgt[3 * i + 2] : nan;
    tx[j] = x; ty[j] = y; tz[j] = z;
    txf[j] = (float)x; tyf[j] = (float)y; tzf[j] = (float)z;
    torig[j] = (int32_t)i;
    if (in) {
      e = fmax(fabs(x), fmax(fabs(y), fabs(z)));
      if (!(e == e)) e = __builtin_huge_val();  // NaN input: disable the filter (threshold becomes +inf)
    }
  }
  for (int off = 32; off > 0; off >>= 1) e = fmax(e, __shfl_xor(e, off, kWave));
  if ((threadIdx.x & 63) == 0 && e > 0.0) atomicMax(emax_bits, (unsigned long long)__double_as_longlong(e));
}

}  // namespace

int64_t knn_scan_padded_targets(int64_t M) { return ((M + STEP - 1) / STEP) * STEP; }

hipError_t launch_targets_soa2(const double* tgt, int64_t M, int64_t Mp, double* tx, double* ty, double* tz,
                               float* txf, float* tyf, float* tzf, int32_t* torig, unsigned long long* emax_bits,
                               hipStream_t st) {
  hipError_t e = hipMemsetAsync(emax_bits, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  const int64_t nb = (Mp + 255) / 256;
  int bits = 1;
  while ((1ll << bits) < Mp) ++bits;
  hipLaunchKernelGGL(k_targets_soa2, dim3((unsigned)nb), dim3(256), 0, st, tgt, M, Mp, bits, tx, ty, tz, txf, tyf, tzf,
                     torig, emax_bits);
  return hipGetLastError();
}

// seed parameters: sample Ms ≈ Mp·12/K slots (a multiple of STEP), threshold = lane-minimum of rank
// ≈ 2.6·K·Ms/Mp.  Returns false when the fast variant does not apply (large K, small M).
bool knn_scan_plan(int64_t Mp, int K, int64_t* Ms, int* seed_rank, int* S2) {
  if (K > 200 || Mp < 16 * STEP || (Mp % STEP) != 0) return false;
  double F = 12.5 / (double)K;
  if (F > 0.25) F = 0.25;
  int64_t ms = (int64_t)((double)Mp * F / STEP + 0.5) * STEP;
  if (ms < STEP) ms = STEP;
  if (ms > Mp) ms = Mp;
  int j = (int)(2.6 * (double)K * (double)ms / (double)Mp + 0.5);
  if (j < 4) j = 4;
  if (j > 44) j = 44;
  *Ms = ms;
  *seed_rank = j - 1;
  *S2 = 1024;
  return true;
}

hipError_t launch_knn_scan(const KnnScanArgs& a, hipStream_t st) {
  const int64_t nq = a.b_hi - a.b_lo;
  if (nq <= 0) return hipSuccess;
  const int64_t nb = (nq + (int64_t)QW * WAVES - 1) / ((int64_t)QW * WAVES);
  const size_t per_wave = sizeof(QD) * QW + sizeof(QF) * QW + sizeof(int) * QW + 16384;
  const size_t smem = per_wave * WAVES;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_scan),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_knn_scan, dim3((unsigned)nb), dim3(256), smem, st, a);
  return hipGetLastError();
}

}  // namespace svnicp
