// knn_tiles.hip — Stage A, pruned variant: exact top-K with spatially sorted target tiles.
//
// Same result contract as knn_topk.hip / knn_scan.hip (reference: SVGDICP::knn_source_cloud,
// src/core/SVGDICP.cpp:201-215; CPU semantics src/core/knn/knn_cpu.cpp:35-67): the K smallest by
// (f64 dist², original index), ascending, dist² = ((dx·dx)+dy·dy)+dz·dz unfused.  It is still a
// brute-force search in the sense that every pair that COULD be among a query's K nearest is
// evaluated; what changes is that whole 512-point target tiles are skipped when their bounding box
// is provably farther than the query's threshold.
//
//  0. spatial_prep.hip sorted targets and queries along a Morton curve; a wave owns 64 consecutive
//     (hence nearby) queries, a tile is 512 consecutive targets with an outward-rounded f32 box.
//  1. seed   The wave picks the ≤16 tiles nearest to its query box and scans them keeping TWO
//     running f32 minima per (query, lane).  The 128 values of a query belong to 128 distinct
//     targets, so their K-th smallest (K ≤ 128), inflated by the f32 error bound, is a threshold τ
//     with AT LEAST K targets inside — a guarantee, not an estimate.
//  2. scan   Tiles whose box-to-box lower bound exceeds the wave's largest threshold are skipped
//     (one vector test per 64 tiles); inside a kept tile, queries whose point-to-box bound exceeds
//     their own threshold are skipped (one vector test per tile).  The surviving (query, tile)
//     pairs run the f32 pre-filter loop of knn_scan.hip; survivors go to the per-query pools.
//  3. select exact f64 d² of the pool; when a query has many survivors its threshold is first
//     tightened from the survivors themselves (same two-minima guarantee); keep d² <= τ, bitonic
//     sort by (d², original index), write the K best to the query's ORIGINAL row.  Pool overflow
//     (thousands of duplicates) falls back to the streaming kernel, as in knn_scan.hip.
//
// All bounds are conservative: boxes are rounded outward, f32 lower bounds are shrunk by 1e-6
// relative, thresholds are inflated by the error terms derived in knn_scan.hip.
#include "kernels.hpp"

namespace svnicp {

namespace {

constexpr int T = 8;
constexpr int STEP = kWave * T;   // 512 slots per tile
constexpr int QW = 64;
constexpr int WAVES = 4;
constexpr int QCAP = 1024;
constexpr int SL = 512;            // LDS sort capacity of the select phase
constexpr int NSQ = 4;            // seed tiles chosen per query (the wave scans the union)
constexpr int MAX_TILES = 8192;   // bitmap capacity (M <= 4M points)
constexpr int QB = 16;            // queries per seed batch (LDS: QB*64*2 floats)
constexpr double kU = 5.9604644775390625e-08;  // 2^-24

struct alignas(16) QF { float x, y, z, thr; };
struct alignas(16) QD { double x, y, z, tau; };

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ bool ent_less(double da, int ia, double db, int ib) {
  return (da < db) || (da == db && ia < ib);
}
__device__ void bitonic_sort(double* sd, int* si, int S, int lane) {
  for (int k = 2; k <= S; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = lane; e < (S >> 1); e += kWave) {
        const int a = ((e & ~(j - 1)) << 1) | (e & (j - 1)), b = a | j;
        const bool up = (a & k) == 0;
        const double da = sd[a], db = sd[b];
        const int ia = si[a], ib = si[b];
        if (ent_less(db, ib, da, ia) == up) { sd[a] = db; si[a] = ib; sd[b] = da; si[b] = ia; }
      }
      wave_sync();
    }
}
__device__ void bitonic_sort128_f32(float* v, int lane) {  // 128 floats, one compare-exchange per lane per round
  for (int k = 2; k <= 128; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int a = ((lane & ~(j - 1)) << 1) | (lane & (j - 1)), b = a | j;
      const bool up = (a & k) == 0;
      const float va = v[a], vb = v[b];
      if ((vb < va) == up) { v[a] = vb; v[b] = va; }
      wave_sync();
    }
}
__device__ __forceinline__ float next_down(float f) {
  if (f == 0.0f) return -1.401298464e-45f;
  int b = __float_as_int(f); b += (f > 0.0f) ? -1 : 1; return __int_as_float(b);
}
__device__ __forceinline__ float next_up(float f) {
  if (f == 0.0f) return 1.401298464e-45f;
  int b = __float_as_int(f); b += (f > 0.0f) ? 1 : -1; return __int_as_float(b);
}
__device__ __forceinline__ float f32_floor(double v) { float f = (float)v; return ((double)f > v) ? next_down(f) : f; }
__device__ __forceinline__ float f32_ceil(double v) { float f = (float)v; return ((double)f < v) ? next_up(f) : f; }
__device__ __forceinline__ float f32_round_up(double v) {
  float f = (float)v;
  if ((double)f < v) f = __int_as_float(__float_as_int(f) + 1);  // v >= 0
  return f;
}
// f32 pre-filter threshold for exact threshold tau (bound derived in knn_scan.hip)
__device__ __forceinline__ float filter_threshold(double tau, double E) {
  if (!(tau < __builtin_huge_val())) return __builtin_huge_valf();
  const double r = sqrt(tau) * (1.0 + 1e-15) + 3.4641016151377553 * kU * E;
  return f32_round_up((r * r) * (1.0 + 5.0 * kU + 1e-6));
}
// lower bound of the squared distance between two boxes (or a degenerate box = point), shrunk
__device__ __forceinline__ float box_lb2(float alo0, float alo1, float alo2, float ahi0, float ahi1, float ahi2,
                                         float blo0, float blo1, float blo2, float bhi0, float bhi1, float bhi2) {
  const float g0 = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo0 - ahi0, alo0 - bhi0));
  const float g1 = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo1 - ahi1, alo1 - bhi1));
  const float g2 = __builtin_fmaxf(0.0f, __builtin_fmaxf(blo2 - ahi2, alo2 - bhi2));
  return (g0 * g0 + g1 * g1 + g2 * g2) * 0.999999f;
}

// per-query record handed from k_knn_tiles (rank / seed / scan) to k_knn_select
struct alignas(16) QRec { double x, y, z, tau; int cnt; int row; float thr, tb; };  // thr / tb: f32 pre-filter and point-to-box thresholds

// survivor e of query position r: the first kTilesBase live in the query's own row, the rest in chunks of the shared arena
__device__ __forceinline__ int pool_read(const KnnTilesArgs& a, int64_t r, int e) {
  if (e < kTilesBase) return a.pool[r * kTilesBase + e];
  const int o = e - kTilesBase;
  const int id = a.chunk_tab[r * kTilesChunks + (o / kTilesChunk)];
  return a.arena[(int64_t)id * kTilesChunk + (o % kTilesChunk)];
}

// select: exact f64 distances of the survivors of query position r, keep d² <= tau, order by (d², original
// index), write the K best to the query's original row.  One wave; sd/si = SL-entry LDS scratch of that wave.
__device__ void select_query(const KnnTilesArgs& a, int64_t r, const QRec& rec, double* sd, int* si, int lane) {
  const int K = a.K, S2 = a.S2;
  const int64_t b = rec.row;
  const int n_first = rec.cnt;
  const double qx = rec.x, qy = rec.y, qz = rec.z;
  double tau = rec.tau;
  if (a.stat_n && lane == 0) a.stat_n[b] = n_first;
  const int n = n_first;
  bool ok = n <= S2;
  int m = 0;
  constexpr int NBR = 8;  // survivor batches of 64 held in registers
  static_assert(NBR * kWave == kTilesBase, "the register path reads the query's own pool row only");
  if (ok && n <= NBR * kWave) {
    // common case: every survivor's slot, then every survivor's coordinates, are requested back to back —
    // two memory round trips per query instead of two per 64 survivors — and d² stays in registers
    int slot[NBR], orig[NBR];
    double d[NBR];
#pragma unroll
    for (int j = 0; j < NBR; ++j) {
      const int e = j * kWave + lane;
      slot[j] = (j * kWave < n && e < n) ? a.pool[r * (int64_t)kTilesBase + e] : -1;   // n <= NBR * 64 = kTilesBase
    }
#pragma unroll
    for (int j = 0; j < NBR; ++j) {
      d[j] = __builtin_huge_val(); orig[j] = 0;
      if (j * kWave < n && slot[j] >= 0) {
        orig[j] = a.torig[slot[j]];
        const double dx = qx - a.tx[slot[j]], dy = qy - a.ty[slot[j]], dz = qz - a.tz[slot[j]];
        d[j] = (dx * dx + dy * dy) + dz * dz;  // knn_cpu.cpp:43-50 order, unfused
      }
    }
    if (n > 2 * kWave && n >= K) {
      // More survivors than the 128-entry ranking below takes: lower the threshold until between K and 128 of
      // the register-resident distances pass.  Any such threshold is exact for the top-K; it is found by a
      // bracketed search on the value (first guess from count ~ d^3, then bisection), a handful of probes of
      // ballot + popcount each.  Ties that keep more than 128 entries at every threshold leave tau unchanged.
      auto count_le = [&](double t) -> int {
        int c = 0;
#pragma unroll
        for (int j = 0; j < NBR; ++j)
          if (j * kWave < n) c += __popcll(__ballot(slot[j] >= 0 && d[j] <= t));
        return c;
      };
      double lo = 0.0, hi = tau;
      if (!(hi < __builtin_huge_val())) {  // no finite threshold yet: start from the largest survivor
        double mx = 0.0;
#pragma unroll
        for (int j = 0; j < NBR; ++j) if (slot[j] >= 0) mx = fmax(mx, d[j]);
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, kWave));
        hi = mx;
      }
      const double target = 0.5 * (K + 2 * kWave);
      double t = hi * (double)__builtin_powf((float)(target / n), 0.6666667f);
      for (int it = 0; it < 48; ++it) {
        if (!(t > lo && t < hi)) t = 0.5 * (lo + hi);
        if (!(t > lo && t < hi)) break;  // bracket exhausted (adjacent doubles)
        const int c = count_le(t);
        if (c < K) lo = t;
        else { hi = t; if (c <= 2 * kWave) break; }
        t = 0.5 * (lo + hi);
      }
      tau = hi < tau ? hi : tau;  // count(d² <= hi) >= K always holds for hi
    }
#pragma unroll
    for (int j = 0; j < NBR; ++j) {
      if (j * kWave < n) {  // wave-uniform
        const bool pass = slot[j] >= 0 && d[j] <= tau;
        const unsigned long long pm = __ballot(pass);
        const int pos = m + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)pm, 0u));
        if (pass && pos < SL) { sd[pos] = d[j]; si[pos] = orig[j]; }
        m += __popcll(pm);
      }
    }
    ok = m >= K && m <= SL;
  } else if (ok) {
    if (n > SL / 2) {
      double m1 = __builtin_huge_val(), m2 = __builtin_huge_val();
      for (int e = lane; e < n; e += kWave) {
        const int slot = pool_read(a, r, e);
        const double dx = qx - a.tx[slot], dy = qy - a.ty[slot], dz = qz - a.tz[slot];
        const double d = (dx * dx + dy * dy) + dz * dz;
        m2 = d < m1 ? m1 : (d < m2 ? d : m2);
        m1 = d < m1 ? d : m1;
      }
      sd[2 * lane] = m1; sd[2 * lane + 1] = m2; si[2 * lane] = 0; si[2 * lane + 1] = 0;
      wave_sync();
      bitonic_sort(sd, si, 128, lane);
      const double t2 = sd[K - 1];
      wave_sync();
      tau = t2 < tau ? t2 : tau;
    }
    for (int e0 = 0; e0 < n; e0 += kWave) {
      const int e = e0 + lane;
      bool pass = false;
      double d = 0.0;
      int orig = 0;
      if (e < n) {
        const int slot = pool_read(a, r, e);
        orig = a.torig[slot];
        const double dx = qx - a.tx[slot], dy = qy - a.ty[slot], dz = qz - a.tz[slot];
        d = (dx * dx + dy * dy) + dz * dz;  // knn_cpu.cpp:43-50 order, unfused
        pass = d <= tau;
      }
      const unsigned long long pm = __ballot(pass);
      const int pos = m + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)pm, 0u));
      if (pass && pos < SL) { sd[pos] = d; si[pos] = orig; }
      m += __popcll(pm);
    }
    ok = m >= K && m <= SL;
  }
  if (!ok) {
    if (lane == 0) {
      const int slot = atomicAdd(a.fail_count, 1);
      a.fail_list[slot] = (int32_t)b;
      // tau has K witnesses, so the fallback may start from it; fewer than K survivors means it cannot be trusted
      if (a.fail_tau) a.fail_tau[slot] = (n <= S2 && m < K) ? __builtin_huge_val() : tau;
    }
    wave_sync();
    return;
  }
  if (m <= 2 * kWave) {
    for (int e = m + lane; e < 2 * kWave + 8; e += kWave) { sd[e] = __builtin_huge_val(); si[e] = 0x7fffffff; }
    // common case (the bisection above leaves K entries plus exact ties): no sort — every lane ranks its (up to
    // two) entries against all m by (d², index) with wave-broadcast LDS reads and stores them at their rank
    wave_sync();
    const int e0 = lane, e1 = lane + kWave;
    const double d0 = e0 < m ? sd[e0] : 0.0, d1 = e1 < m ? sd[e1] : 0.0;
    const int i0 = e0 < m ? si[e0] : 0, i1 = e1 < m ? si[e1] : 0;
    // rank = #(d_j < d) + #(d_j == d and i_j < i).  The first count is compare + add-with-carry per entry (no scalar
    // mask logic, which is slow behind vector compares on this chip).  Exact ties are rare: they show up as two entries
    // with the same first count (a flag per rank in LDS), and only then does the second count run.
    int* flag = si + 2 * kWave + 8;   // [128] behind the entries
    flag[lane] = 0; flag[lane + kWave] = 0;
    int r0 = 0, r1 = 0;
    for (int j0 = 0; j0 < m; j0 += 8) {  // eight broadcast reads in flight; entries past m are +inf (never less)
      double dj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) dj[u] = sd[j0 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        r0 += dj[u] < d0 ? 1 : 0;
        r1 += dj[u] < d1 ? 1 : 0;
      }
    }
    wave_sync();
    if (e0 < m) atomicAdd(&flag[r0], 1);
    if (e1 < m) atomicAdd(&flag[r1], 1);
    wave_sync();
    if (__ballot((e0 < m && flag[r0] > 1) || (e1 < m && flag[r1] > 1))) {  // exact ties: order them by original index
      for (int j = 0; j < m; ++j) {
        const double dj = sd[j];
        const int ij = si[j];
        r0 += (dj == d0 && ij < i0) ? 1 : 0;
        r1 += (dj == d1 && ij < i1) ? 1 : 0;
      }
    }
    if (e0 < m && r0 < K) { a.out_idx[b * K + r0] = i0; a.out_d2[b * K + r0] = d0; }
    if (e1 < m && r1 < K) { a.out_idx[b * K + r1] = i1; a.out_d2[b * K + r1] = d1; }
    wave_sync();
  } else {
    int S = 256;
    while (S < m) S <<= 1;
    for (int e = m + lane; e < S; e += kWave) { sd[e] = __builtin_huge_val(); si[e] = 0x7fffffff; }
    wave_sync();
    bitonic_sort(sd, si, S, lane);
    for (int e = lane; e < K; e += kWave) { a.out_idx[b * K + e] = si[e]; a.out_d2[b * K + e] = sd[e]; }
    wave_sync();
  }
}

__global__ __launch_bounds__(256) void k_knn_seed(KnnTilesArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = a.K, n_tiles = a.n_tiles;
  // one workgroup = one group of 64 queries: wave 0 loads and ranks (lane = query), then each of the four waves seeds one
  // batch of QB = 16 queries against the tiles the group picked, then wave 0 turns the K-th values into thresholds.
  // (One wave per group did all four batches in sequence: 2048 long-running waves at C3, two per SIMD, the slowest 1.6x
  // the median, set the kernel time.)
  constexpr size_t kShared = sizeof(QD) * QW + sizeof(QF) * QW + 4 * sizeof(int) * QW + sizeof(unsigned int) * (MAX_TILES / 32);
  constexpr size_t kScratch = sizeof(float2) * QB * kWave + 512;   // per wave: two minima per (query of the batch, lane); rank: 64 boxes
  static_assert(QB * WAVES == QW, "one batch per wave");
  QD* qd = reinterpret_cast<QD*>(smem);
  QF* qf = reinterpret_cast<QF*>(smem + sizeof(QD) * QW);
  int* cnt = reinterpret_cast<int*>(smem + (sizeof(QD) + sizeof(QF)) * QW);
  int* qb = cnt + QW;                                   // original row of each query
  float* thrb = reinterpret_cast<float*>(qb + QW);      // box-test threshold per query
  int* spare = reinterpret_cast<int*>(thrb + QW);       // K-th value of each query (float bits)
  unsigned int* tbits = reinterpret_cast<unsigned int*>(spare + QW);   // tile bitmap [MAX_TILES/32]
  unsigned char* scratch = smem + kShared + kScratch * wave;
  float2* lm = reinterpret_cast<float2*>(scratch);                 // seed: [QB][64] two minima

  const unsigned int grp = blockIdx.x;   // natural order: neighbouring groups share tiles in L2 (a strided order, as in the scan kernel, cost 5 % at C5)
  const int64_t q0 = a.b_lo + (int64_t)grp * QW;  // first query position of the workgroup
  const int nq = (a.b_hi - q0) < QW ? (int)(a.b_hi - q0) : QW;
  const double Et = __longlong_as_double((long long)*a.emax_bits);

  unsigned long long tphase = a.phase_cycles ? __builtin_readcyclecounter() : 0ull;
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto phase_mark = [&](int i) {  // debug: wave cycles per phase, summed over waves at the end (SVNICP_DEBUG)
    if (!a.phase_cycles) return;
    const unsigned long long now = __builtin_readcyclecounter();
    pacc[i] += now - tphase;
    tphase = now;
  };
  // ---- 0. the group's queries (curve order) and their common box ----
  float wlo0 = 0.f, wlo1 = 0.f, wlo2 = 0.f, whi0 = 0.f, whi1 = 0.f, whi2 = 0.f;
  const int nwords = (n_tiles + 31) >> 5;
  if (wave == 0) {
    const bool act = lane < nq;
    const int64_t b = act ? (int64_t)a.qorder[q0 - a.b_lo + lane] : 0;
    QD s; s.x = s.y = s.z = 0.0; s.tau = 0.0;
    if (act) {
      const double sx = a.src[3 * b], sy = a.src[3 * b + 1], sz = a.src[3 * b + 2];
      const double* R = a.pose.R0;
      s.x = (sx * R[0] + sy * R[1] + sz * R[2]) + a.pose.t0[0];   // SVGDICP.cpp:204
      s.y = (sx * R[3] + sy * R[4] + sz * R[5]) + a.pose.t0[1];
      s.z = (sx * R[6] + sy * R[7] + sz * R[8]) + a.pose.t0[2];
    }
    qd[lane] = s;
    QF f; f.x = (float)s.x; f.y = (float)s.y; f.z = (float)s.z; f.thr = -1.0f;
    qf[lane] = f;
    cnt[lane] = 0; qb[lane] = (int)b; thrb[lane] = -1.0f;
    const float inf = __builtin_huge_valf();
    wlo0 = act ? f32_floor(s.x) : inf; wlo1 = act ? f32_floor(s.y) : inf; wlo2 = act ? f32_floor(s.z) : inf;
    whi0 = act ? f32_ceil(s.x) : -inf; whi1 = act ? f32_ceil(s.y) : -inf; whi2 = act ? f32_ceil(s.z) : -inf;
    for (int off = 32; off > 0; off >>= 1) {
      wlo0 = __builtin_fminf(wlo0, __shfl_xor(wlo0, off, kWave)); whi0 = __builtin_fmaxf(whi0, __shfl_xor(whi0, off, kWave));
      wlo1 = __builtin_fminf(wlo1, __shfl_xor(wlo1, off, kWave)); whi1 = __builtin_fmaxf(whi1, __shfl_xor(whi1, off, kWave));
      wlo2 = __builtin_fminf(wlo2, __shfl_xor(wlo2, off, kWave)); whi2 = __builtin_fmaxf(whi2, __shfl_xor(whi2, off, kWave));
    }
  }
  wave_sync();
  const float* bx_lo0 = a.tile_box, *bx_lo1 = a.tile_box + n_tiles, *bx_lo2 = a.tile_box + 2 * (size_t)n_tiles;
  const float* bx_hi0 = a.tile_box + 3 * (size_t)n_tiles, *bx_hi1 = a.tile_box + 4 * (size_t)n_tiles,
             * bx_hi2 = a.tile_box + 5 * (size_t)n_tiles;

  // ---- 1. seed: nearest tiles, two minima per (query, lane), K-th of 128 -> guaranteed threshold ----
  {
    // each query (lane) picks its own NSQ nearest tiles: point-to-box bound, ties (inside several
    // boxes) broken toward the nearest box centre; the workgroup scans the union of all picks
    if (wave == 0) {
      for (int e = lane; e < nwords; e += kWave) tbits[e] = 0u;
      wave_sync();
      const float px = qf[lane].x, py = qf[lane].y, pz = qf[lane].z;
      unsigned long long best[NSQ];
#pragma unroll
      for (int i = 0; i < NSQ; ++i) best[i] = ~0ull;
      // 64 tile boxes at a time are fetched lane-per-tile (coalesced) into LDS and then read back as wave
      // broadcasts: every query scores every tile without a scalar-load round trip per tile
      float4* boxs = reinterpret_cast<float4*>(scratch);  // [64][2] (the seed buffers are not in use yet)
      // Groups of 64 consecutive tiles (compact along the curve) are visited nearest first, by the lower bound between
      // the wave's query box and the group's box, and the visit stops when that bound exceeds every query's current
      // fourth-best score: no tile of a farther group can enter any top-NSQ list, so the picks equal those of a full
      // pass — at a cost that no longer grows with the number of tiles (2 M-point targets: 4096 tiles).
      const int n_groups = (n_tiles + kWave - 1) / kWave;  // <= 128
      unsigned int gkey[2] = {0xffffffffu, 0xffffffffu};     // lane's groups: bound (7 low mantissa bits cut) | group index
      for (int g = 0; g < n_groups; ++g) {
        const int tl = g * kWave + lane;
        const float inf = __builtin_huge_valf();
        float lo0 = inf, lo1 = inf, lo2 = inf, hi0 = -inf, hi1 = -inf, hi2 = -inf;
        if (tl < n_tiles) { lo0 = bx_lo0[tl]; lo1 = bx_lo1[tl]; lo2 = bx_lo2[tl]; hi0 = bx_hi0[tl]; hi1 = bx_hi1[tl]; hi2 = bx_hi2[tl]; }
        for (int off = 32; off > 0; off >>= 1) {
          lo0 = __builtin_fminf(lo0, __shfl_xor(lo0, off, kWave)); hi0 = __builtin_fmaxf(hi0, __shfl_xor(hi0, off, kWave));
          lo1 = __builtin_fminf(lo1, __shfl_xor(lo1, off, kWave)); hi1 = __builtin_fmaxf(hi1, __shfl_xor(hi1, off, kWave));
          lo2 = __builtin_fminf(lo2, __shfl_xor(lo2, off, kWave)); hi2 = __builtin_fmaxf(hi2, __shfl_xor(hi2, off, kWave));
        }
        const float glb = box_lb2(wlo0, wlo1, wlo2, whi0, whi1, whi2, lo0, lo1, lo2, hi0, hi1, hi2);  // empty boxes give +inf or NaN
        const unsigned int key = (glb < inf) ? ((__float_as_uint(glb) & ~127u) | (unsigned int)g) : 0xffffffffu;
        if (lane == (g & (kWave - 1))) gkey[g >> 6] = key;
      }
      for (;;) {
        unsigned int kmin = gkey[0] < gkey[1] ? gkey[0] : gkey[1];
        for (int off = 32; off > 0; off >>= 1) { const unsigned int o = __shfl_xor(kmin, off, kWave); kmin = o < kmin ? o : kmin; }
        if (kmin == 0xffffffffu) break;
        float w4 = -__builtin_huge_valf();  // largest fourth-best score among the wave's queries (+inf while a list is not full)
        if (lane < nq) w4 = best[NSQ - 1] == ~0ull ? __builtin_huge_valf() : __uint_as_float((unsigned int)(best[NSQ - 1] >> 32));
        for (int off = 32; off > 0; off >>= 1) w4 = __builtin_fmaxf(w4, __shfl_xor(w4, off, kWave));
        if (__uint_as_float(kmin & ~127u) > w4) break;
        const int gsel = (int)(kmin & 127u);
        if (lane == (gsel & (kWave - 1))) gkey[gsel >> 6] = 0xffffffffu;
        const int t0 = gsel * kWave;
        const int tl = t0 + lane;
        if (tl < n_tiles) {
          boxs[2 * lane] = make_float4(bx_lo0[tl], bx_lo1[tl], bx_lo2[tl], bx_hi0[tl]);
          boxs[2 * lane + 1] = make_float4(bx_hi1[tl], bx_hi2[tl], 0.f, 0.f);
        }
        wave_sync();
        const int cntt = (n_tiles - t0) < kWave ? (n_tiles - t0) : kWave;
#pragma unroll 4
        for (int i = 0; i < cntt; ++i) {
          const float4 b0 = boxs[2 * i], b1 = boxs[2 * i + 1];
          const float l0 = b0.x, l1 = b0.y, l2 = b0.z, h0 = b0.w, h1 = b1.x, h2 = b1.y;
          const int tile = t0 + i;
          const float lb = box_lb2(px, py, pz, px, py, pz, l0, l1, l2, h0, h1, h2);
          const float c0 = px - 0.5f * (l0 + h0), c1 = py - 0.5f * (l1 + h1), c2 = pz - 0.5f * (l2 + h2);
          const float score = lb + 1e-3f * (c0 * c0 + c1 * c1 + c2 * c2);
          unsigned long long key = (score < __builtin_huge_valf()) ? (((unsigned long long)__float_as_uint(score) << 32) | (unsigned int)tile) : ~0ull;
#pragma unroll
          for (int u = 0; u < NSQ; ++u) {  // insertion into the sorted top-NSQ
            const unsigned long long lo = key < best[u] ? key : best[u];
            key = key < best[u] ? best[u] : key;
            best[u] = lo;
          }
        }
        wave_sync();
      }
      if (lane < nq) {
#pragma unroll
        for (int i = 0; i < NSQ; ++i)
          if (best[i] != ~0ull) { const unsigned int t = (unsigned int)(best[i] & 0xffffffffull); atomicOr(&tbits[t >> 5], 1u << (t & 31)); }
      }
    }
    phase_mark(0);
    __syncthreads();
    phase_mark(3);
    const int qb0 = wave * QB;   // this wave's batch
    if (qb0 < nq) {
      const int nb = (nq - qb0) < QB ? (nq - qb0) : QB;
      for (int e = lane; e < QB * kWave; e += kWave) lm[e] = make_float2(__builtin_huge_valf(), __builtin_huge_valf());
      wave_sync();
      for (int wd = 0; wd < nwords; ++wd) {
        unsigned int bits = __builtin_amdgcn_readfirstlane(tbits[wd]);
        while (bits) {
          const int tile = (wd << 5) + (int)__builtin_ctz(bits);
          bits &= bits - 1;
          if (a.phase_cycles) pacc[4] += 1;
          const int64_t tb = (int64_t)tile * STEP;
          float x[T], y[T], z[T];
#pragma unroll
          for (int t = 0; t < T; ++t) { x[t] = a.txf[tb + t * kWave + lane]; y[t] = a.tyf[tb + t * kWave + lane]; z[t] = a.tzf[tb + t * kWave + lane]; }
          for (int q = 0; q < nb; ++q) {
            const float4 cur = *reinterpret_cast<const float4*>(&qf[qb0 + q]);
            float2 m = lm[q * kWave + lane];
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const float dx = cur.x - x[t], dy = cur.y - y[t], dz = cur.z - z[t];
              float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
              d = __builtin_fminf(d, __builtin_huge_valf());     // NaN padding -> +inf
              m.y = __builtin_amdgcn_fmed3f(m.x, m.y, d);        // second smallest
              m.x = __builtin_fminf(m.x, d);
            }
            lm[q * kWave + lane] = m;
          }
        }
      }
      wave_sync();
      phase_mark(5);
      for (int q = 0; q < nb; ++q) {
        const float2 m = lm[q * kWave + lane];
        // K-th smallest of the 128 lane minima: bisection on the (non-negative) float bit patterns, 31 probes of
        // two compares + ballot/popcount instead of a 28-round LDS sort
        const unsigned int kx = __float_as_uint(m.x), ky = __float_as_uint(m.y);
        unsigned int kth = 0u;
        for (int bit = 30; bit >= 0; --bit) {
          const unsigned int t_try = kth | ((1u << bit) - 1u);  // this bit 0, lower bits 1
          const int c = __popcll(__ballot(kx <= t_try)) + __popcll(__ballot(ky <= t_try));
          if (c < K) kth |= 1u << bit;
        }
        if (lane == 0) spare[qb0 + q] = (int)kth;  // K <= 128 witnesses with f32 distance <= this value
      }
      phase_mark(6);
    }
    phase_mark(1);
    __syncthreads();
    phase_mark(3);
    if (wave == 0 && lane < nq) {  // thresholds of all queries at once, one lane per query
      const double tv = (double)__uint_as_float((unsigned int)spare[lane]);
      const double E = fmax(Et, fmax(fabs(qd[lane].x), fmax(fabs(qd[lane].y), fabs(qd[lane].z))));
      double tau = __builtin_huge_val();
      if (tv < (double)__builtin_huge_valf()) {
        const double r = sqrt(tv) * (1.0 + 1e-6) + 4.0 * kU * E;  // exact distance of every witness <= r
        tau = (r * r) * (1.0 + 1e-12);
      }
      qd[lane].tau = tau;
      const float thr = filter_threshold(tau, E);
      qf[lane].thr = thr;
      float tb = __builtin_huge_valf();
      if (thr < __builtin_huge_valf()) {  // point-to-box test uses the rounded query: allow its rounding error
        const double rb = sqrt((double)thr) + 2.0 * kU * E;
        tb = f32_round_up((rb * rb) * (1.0 + 1e-6));
      }
      thrb[lane] = tb;
    }
    wave_sync();
  }

  // ---- 2. hand over: thresholds and positions go to the scan kernel (its own launch: one workgroup per 64 queries, so a
  // heavy group no longer pins the three lighter groups of its workgroup, and the hardware balances the groups) ----
  if (wave == 0 && lane < nq) {
    QRec rec;
    rec.x = qd[lane].x; rec.y = qd[lane].y; rec.z = qd[lane].z; rec.tau = qd[lane].tau;
    rec.cnt = 0; rec.row = qb[lane]; rec.thr = qf[lane].thr; rec.tb = thrb[lane];
    reinterpret_cast<QRec*>(a.qrec)[q0 + lane] = rec;
  }
  phase_mark(1);
  if (a.phase_cycles && lane == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) atomicAdd(&a.phase_cycles[i], pacc[i]);
    unsigned long long* pw = a.phase_cycles + 8 + 8 * ((size_t)grp * WAVES + wave);  // per-wave record
#pragma unroll
    for (int i = 0; i < 7; ++i) pw[i] = pacc[i];
  }
}

// ---- scan: the tiles that can matter for one group of 64 consecutive (Morton-ordered) queries ----
// Tiles whose box is farther from the group's box than the group's largest threshold are skipped (one vector test per
// 64 tiles).  The remaining tiles are drawn, one at a time, from the workgroup's ticket counter by its SW waves (a fixed
// split left waves idle at the final barrier for 23 % of the wave cycles; counters in global memory, tried so that several
// workgroups could share a group, cost more than they balanced).  Inside a tile, queries whose point-to-box bound exceeds
// their own threshold are skipped (one vector test per tile); the rest run the f32 pre-filter and the survivors go to the
// query's pool.
template <int SW>
__global__ __launch_bounds__(64 * SW) void k_knn_scan_tiles(KnnTilesArgs a) {
  __shared__ __align__(16) QF s_qf[QW];
  __shared__ int s_cnt[QW];
  __shared__ int s_ticket;
  __shared__ __align__(16) int2 s_queue[SW][QCAP];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S2 = a.S2, n_tiles = a.n_tiles;
  int2* queue = s_queue[wave];
  // groups in a strided order: neighbours on the curve have similar work, and a run of heavy groups dispatched together
  // at the end of the launch is what the kernel time would wait for
  const unsigned int ngrp = gridDim.x;
  const unsigned int grp = (unsigned int)(((unsigned long long)blockIdx.x * a.group_stride) % ngrp);
  const int64_t q0 = a.b_lo + (int64_t)grp * QW;
  const int nq = (a.b_hi - q0) < QW ? (int)(a.b_hi - q0) : QW;
  QRec* recs = reinterpret_cast<QRec*>(a.qrec);
  unsigned long long tphase = a.phase_cycles ? __builtin_readcyclecounter() : 0ull;
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto phase_mark = [&](int i) {
    if (!a.phase_cycles) return;
    const unsigned long long now = __builtin_readcyclecounter();
    pacc[i] += now - tphase;
    tphase = now;
  };
  // every wave keeps the group's queries in registers (lane = query) for the box tests; the pair loop reads them from LDS
  float gx = 0.f, gy = 0.f, gz = 0.f, gtb = -1.0f;
  float wlo0, wlo1, wlo2, whi0, whi1, whi2, tmax;
  {
    const bool act = lane < nq;
    QF f; f.x = f.y = f.z = 0.f; f.thr = -1.0f;
    const float inf = __builtin_huge_valf();
    wlo0 = wlo1 = wlo2 = inf; whi0 = whi1 = whi2 = -inf;
    if (act) {
      const QRec rec = recs[q0 + lane];
      f.x = (float)rec.x; f.y = (float)rec.y; f.z = (float)rec.z; f.thr = rec.thr;
      gtb = rec.tb;
      wlo0 = f32_floor(rec.x); wlo1 = f32_floor(rec.y); wlo2 = f32_floor(rec.z);
      whi0 = f32_ceil(rec.x); whi1 = f32_ceil(rec.y); whi2 = f32_ceil(rec.z);
    }
    gx = f.x; gy = f.y; gz = f.z;
    if (wave == 0) { s_qf[lane] = f; s_cnt[lane] = 0; }
    if (threadIdx.x == 0) s_ticket = 0;
    tmax = gtb;
    for (int off = 32; off > 0; off >>= 1) {
      wlo0 = __builtin_fminf(wlo0, __shfl_xor(wlo0, off, kWave)); whi0 = __builtin_fmaxf(whi0, __shfl_xor(whi0, off, kWave));
      wlo1 = __builtin_fminf(wlo1, __shfl_xor(wlo1, off, kWave)); whi1 = __builtin_fmaxf(whi1, __shfl_xor(whi1, off, kWave));
      wlo2 = __builtin_fminf(wlo2, __shfl_xor(wlo2, off, kWave)); whi2 = __builtin_fmaxf(whi2, __shfl_xor(whi2, off, kWave));
      tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, off, kWave));
    }
  }
  __syncthreads();
  const float* bx_lo0 = a.tile_box, *bx_lo1 = a.tile_box + n_tiles, *bx_lo2 = a.tile_box + 2 * (size_t)n_tiles;
  const float* bx_hi0 = a.tile_box + 3 * (size_t)n_tiles, *bx_hi1 = a.tile_box + 4 * (size_t)n_tiles,
             * bx_hi2 = a.tile_box + 5 * (size_t)n_tiles;
  // Survivors are queued per wave and drained 64 at a time: entries of one (query, tile row) are adjacent, and each run
  // of equal queries reserves its pool positions with one update of the query's counter (LDS) by its first lane.
  // Positions past the query's own pool row live in arena chunks: the lane that gets the first position of a chunk
  // allocates it and publishes its id in chunk_tab, the others (in this wave or another wave of the workgroup — never in
  // another workgroup: a query's counter is workgroup-local) wait for the id.  The allocating lane has finished its store
  // before any lane of its own wave starts to wait, and other waves do not hold it up, so the wait ends; it is bounded
  // anyway, and a query whose chunk could not be had (arena exhausted) is marked failed and redone exactly by the fallback.
  int qcount = 0;
  auto put = [&](int q, int pos, bool on, int slot) {
    const int64_t r = q0 + q;
    if (on && pos < kTilesBase) a.pool[r * kTilesBase + pos] = slot;
    const bool ovf = on && pos >= kTilesBase && pos < S2;
    if (__ballot(ovf)) {
      const int o = pos - kTilesBase;
      int32_t* tab = a.chunk_tab + r * kTilesChunks + (ovf ? o / kTilesChunk : 0);
      if (ovf && (o % kTilesChunk) == 0) {
        int id = atomicAdd(a.chunk_tab + a.tab_rows * (int64_t)kTilesChunks, 1) + 1;
        if (id >= a.arena_cap) id = -2;
        __hip_atomic_store(tab, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (ovf) {
        int id = -1;
        for (int spin = 0; spin < (1 << 22) && id == -1; ++spin) {
          id = __hip_atomic_load(tab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (id == -1) __builtin_amdgcn_s_sleep(2);
        }
        if (id >= 0) a.arena[(int64_t)id * kTilesChunk + (o % kTilesChunk)] = slot;
        else atomicMax(&s_cnt[q], 1 << 24);  // no chunk: more than S2 survivors on record -> the select phase hands the query to the fallback
      }
    }
  };
  auto flush = [&]() {
    wave_sync();
    for (int e0 = 0; e0 < qcount; e0 += kWave) {
      const int e = e0 + lane;
      const bool on = e < qcount;
      const int2 ent = on ? queue[e] : make_int2(0, -1 - lane);
      const int prev = __shfl_up(ent.y, 1, kWave);
      const unsigned long long lead = __ballot(on && (lane == 0 || ent.y != prev));
      const unsigned long long upto = lead & (~0ull >> (63 - lane));               // leaders at or below this lane
      const int L = 63 - (int)__builtin_clzll(upto | 1ull);
      const unsigned long long after = lead & ~((2ull << L) - 1ull);
      const int nvalid = (qcount - e0) < kWave ? (qcount - e0) : kWave;
      const int end = after ? (int)__builtin_ctzll(after) : nvalid;
      const int q = ent.y & (QW - 1);
      int base = 0;
      if (on && lane == L) base = atomicAdd(&s_cnt[q], end - L);
      base = __shfl(base, L, kWave);
      put(q, base + (lane - L), on, ent.x);
    }
    qcount = 0;
    wave_sync();
  };
  {
    auto next_ticket = [&]() { int v = 0; if (lane == 0) v = atomicAdd(&s_ticket, 1); return __builtin_amdgcn_readfirstlane(v); };
    int my = next_ticket(), seen = 0;
    for (int t0 = 0; t0 < n_tiles; t0 += kWave) {
      const int tl = t0 + lane;
      bool need = false;
      if (tl < n_tiles)
        need = box_lb2(wlo0, wlo1, wlo2, whi0, whi1, whi2, bx_lo0[tl], bx_lo1[tl], bx_lo2[tl], bx_hi0[tl], bx_hi1[tl], bx_hi2[tl]) <= tmax;
      const unsigned long long any = __ballot(need);
      const int n_here = __popcll(any);
      while (my < seen + n_here) {
        unsigned long long rem = any;
        for (int k = my - seen; k > 0; --k) rem &= rem - 1;
        const int tile = t0 + (int)__builtin_ctzll(rem);
        my = next_ticket();   // drawn early: the counter's latency hides behind this tile's work
        if (a.phase_cycles) pacc[5] += 1;
        const float l0 = bx_lo0[tile], l1 = bx_lo1[tile], l2 = bx_lo2[tile], h0 = bx_hi0[tile], h1 = bx_hi1[tile], h2 = bx_hi2[tile];
        unsigned long long qmask = __ballot(lane < nq && box_lb2(gx, gy, gz, gx, gy, gz, l0, l1, l2, h0, h1, h2) <= gtb);
        if (!qmask) continue;
        if (a.phase_cycles) pacc[6] += __popcll(qmask);
        const int64_t tb = (int64_t)tile * STEP;
        float x[T], y[T], z[T];
#pragma unroll
        for (int t = 0; t < T; ++t) { x[t] = a.txf[tb + t * kWave + lane]; y[t] = a.tyf[tb + t * kWave + lane]; z[t] = a.tzf[tb + t * kWave + lane]; }
        while (qmask) {
          const int q = (int)__builtin_ctzll(qmask);
          qmask &= qmask - 1;
          const float4 cur = *reinterpret_cast<const float4*>(&s_qf[q]);
          float d[T];
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const float dx = cur.x - x[t], dy = cur.y - y[t], dz = cur.z - z[t];
            d[t] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          }
          float dm = __builtin_fminf(__builtin_fminf(d[0], d[1]), d[2]);
#pragma unroll
          for (int t = 3; t + 1 < T; t += 2) dm = __builtin_fminf(__builtin_fminf(dm, d[t]), d[t + 1]);
          if constexpr ((T - 3) % 2 == 1) dm = __builtin_fminf(dm, d[T - 1]);
          if (__ballot(dm <= cur.w)) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const bool hit = d[t] <= cur.w;
              const unsigned long long m = __ballot(hit);
              if (m) {
                const int pos = qcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                if (hit) queue[pos] = make_int2((int)(tb + t * kWave + lane), q);
                qcount += __popcll(m);
              }
            }
            qcount = __builtin_amdgcn_readfirstlane(qcount);
            if (qcount > QCAP - STEP) flush();
          }
        }
      }
      seen += n_here;
    }
    flush();
  }
  phase_mark(2);
  __syncthreads();  // every wave's survivors are counted before the counts are published
  phase_mark(3);
  if (wave == 0 && lane < nq) recs[q0 + lane].cnt = s_cnt[lane];
  if (a.phase_cycles && lane == 0) {
    atomicAdd(&a.phase_cycles[2], pacc[2]); atomicAdd(&a.phase_cycles[3], pacc[3]);
    atomicAdd(&a.phase_cycles[5], pacc[5]); atomicAdd(&a.phase_cycles[6], pacc[6]);
    atomicMax(&a.phase_cycles[7], pacc[2]);   // slowest scan wave
  }
}

// one wave per query position (grid-stride): see select_query
__global__ __launch_bounds__(256) void k_knn_select(KnnTilesArgs a) {
  __shared__ __align__(16) double s_sd[WAVES][SL];
  __shared__ int s_si[WAVES][SL];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const QRec* recs = reinterpret_cast<const QRec*>(a.qrec);
  for (int64_t r = a.b_lo + (int64_t)blockIdx.x * WAVES + wave; r < a.b_hi; r += (int64_t)gridDim.x * WAVES) {
    const QRec rec = recs[r];
    select_query(a, r, rec, s_sd[wave], s_si[wave], lane);
  }
}

}  // namespace

bool knn_tiles_applicable(int64_t Mp, int K) { return K <= 128 && Mp >= 16 * STEP && (Mp % STEP) == 0 && Mp / STEP <= MAX_TILES; }

hipError_t launch_knn_tiles(const KnnTilesArgs& a, hipStream_t st) {
  const int64_t nq = a.b_hi - a.b_lo;
  if (nq <= 0) return hipSuccess;
  const int64_t nb = (nq + QW - 1) / QW;
  if (a.n_tiles > MAX_TILES) return hipErrorInvalidValue;
  const size_t smem = sizeof(QD) * QW + sizeof(QF) * QW + 4 * sizeof(int) * QW + sizeof(unsigned int) * (MAX_TILES / 32) +
                      (sizeof(float2) * QB * kWave + 512) * WAVES;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_seed),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_knn_seed, dim3((unsigned)nb), dim3(256), smem, st, a);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.scan_split == 4) hipLaunchKernelGGL(k_knn_scan_tiles<4>, dim3((unsigned)((nq + QW - 1) / QW)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_knn_scan_tiles<8>, dim3((unsigned)((nq + QW - 1) / QW)), dim3(512), 0, st, a);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  int64_t nbs = (nq + WAVES - 1) / WAVES;
  if (nbs > 65536) nbs = 65536;  // grid-stride beyond
  hipLaunchKernelGGL(k_knn_select, dim3((unsigned)nbs), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace svnicp
