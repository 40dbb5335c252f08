// knn_topk.hip — Stage A: exact brute-force top-K of every (R0·s + t0) against the whole target.
//
// Replaces  SVGDICP::knn_source_cloud (src/core/SVGDICP.cpp:201-215) and the PyTorch3D kernel it
// launches (KNearestNeighborKernelV1, src/core/knn/knn.cu:68-111 + MinK, include/core/utils/mink.cuh:40-115),
// with the result contract of the reference CPU path (src/core/knn/knn_cpu.cpp:35-67):
// the K smallest by (dist², index), strict '<' (lowest index wins ties), ascending; dist² is
// ((dx·dx)+dy·dy)+dz·dz in f64 with no fused multiply-add; slots beyond M keep idx 0 / dist 0.
//
// MI355X design (not the reference's one-thread-per-query / global-memory MinK):
//  * lane ↔ target point.  A wave keeps 64×T target points in VGPRs (SoA loads, 512 B coalesced
//    per instruction, next tile prefetched) and walks its 64 query points against them; the query
//    and its running threshold are wave-uniform (LDS broadcast reads).
//  * the hot loop is 8 f64 VALU ops + 1 compare per pair, no data-dependent work: a pair only
//    leaves the fast path when d² <= thr (≈ K·ln(M/K) times per query out of M).
//  * the target stream is stored in a pseudo-random permutation (k_targets_soa2, knn_scan.hip), so
//    every 256-slot tile is a uniform sample of the cloud:
//    scan-ordered input would otherwise tighten the threshold gradually and push ~100× more pairs
//    through the slow path (measured: 118 ms → 35 ms at C3).
//  * survivors are compacted with ballot/mbcnt into the query's candidate pool (global, L2
//    resident, touched rarely); when a pool is nearly full the wave bitonic-sorts it in LDS by
//    (d², idx), keeps the best K and lowers the threshold.  The filter uses '<=' on a
//    non-increasing threshold, so it is conservative and the final sort decides ties exactly
//    like the reference's (dist, idx) heap.
#include "kernels.hpp"

namespace svnicp {

namespace {

constexpr int T = 4;             // target points per lane per step
constexpr int STEP = kWave * T;  // 256 target points per wave step
constexpr int QW = 64;           // query points per wave
constexpr int WAVES = 4;         // waves per workgroup

struct alignas(16) QSlot { double x, y, z, thr; };

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool ent_less(double da, int ia, double db, int ib) {
  return (da < db) || (da == db && ia < ib);
}

// ascending bitonic sort of S (power of two) (d, i) pairs held in LDS, by one wave
__device__ void bitonic_sort(double* sd, int* si, int S, int lane) {
  for (int k = 2; k <= S; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = lane; e < (S >> 1); e += kWave) {
        const int a = ((e & ~(j - 1)) << 1) | (e & (j - 1));
        const int b = a | j;
        const bool up = (a & k) == 0;
        const double da = sd[a], db = sd[b];
        const int ia = si[a], ib = si[b];
        const bool a_gt_b = ent_less(db, ib, da, ia);
        if (a_gt_b == up) { sd[a] = db; si[a] = ib; sd[b] = da; si[b] = ia; }
      }
      wave_sync();
    }
  }
}

// sort the pool of query q, keep the best K, refresh the threshold
__device__ void merge_pool(int q, int lane, int K, int S, QSlot* qv, int* cnt, double* sd, int* si,
                           double* pool_d, int32_t* pool_i, int64_t pool_base) {
  // this wave's own earlier global stores to the pool must have landed before it reads them back
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  const int n = __builtin_amdgcn_readfirstlane(cnt[q]);
  for (int e = lane; e < S; e += kWave) {
    const bool in = e < n;
    sd[e] = in ? pool_d[pool_base + e] : __builtin_huge_val();
    si[e] = in ? pool_i[pool_base + e] : 0x7fffffff;
  }
  wave_sync();
  bitonic_sort(sd, si, S, lane);
  const int n2 = n < K ? n : K;
  for (int e = lane; e < n2; e += kWave) { pool_d[pool_base + e] = sd[e]; pool_i[pool_base + e] = si[e]; }
  if (lane == 0) {
    cnt[q] = n2;
    if (n >= K) qv[q].thr = sd[K - 1];
  }
  wave_sync();
}

__global__ __launch_bounds__(256) void k_knn_topk(KnnArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int S = a.S, K = a.K;
  // per-wave LDS carve-up
  const size_t per_wave = sizeof(QSlot) * QW + 2 * sizeof(int) * QW + (size_t)S * (sizeof(double) + sizeof(int));
  unsigned char* base = smem + per_wave * wave;
  QSlot* qv = reinterpret_cast<QSlot*>(base);
  double* sd = reinterpret_cast<double*>(base + sizeof(QSlot) * QW);
  int* si = reinterpret_cast<int*>(base + sizeof(QSlot) * QW + sizeof(double) * (size_t)S);
  int* cnt = si + S;
  int* qb = cnt + QW;  // global query index of each slot

  // direct mode: queries b_lo..b_hi, one 64-query chunk per wave.  list mode (fallback of
  // knn_scan.hip): queries qlist[0..*qlist_count), chunks strided over the fixed grid.
  // sliced list mode (few failed queries): work item w = (failed query w / slices, target slice w % slices), one
  // per wave chunk; the per-slice top-K lists go to sl_d / sl_i and k_knn_merge_slices finishes the query.  A list
  // longer than slice_max_queries is left to the plain list mode (and vice versa): both launches are always issued.
  const bool list_mode = a.qlist != nullptr;
  const bool sliced = list_mode && a.slices > 0;
  const int64_t n_listed = list_mode ? (int64_t)*a.qlist_count : 0;
  if (list_mode && ((n_listed <= a.slice_max_queries) != sliced)) return;  // the other launch owns this regime
  const int64_t n_queries = sliced ? n_listed * a.slices : list_mode ? n_listed : (a.b_hi - a.b_lo);
  const int cq = sliced ? 1 : list_mode ? a.list_qw : QW;  // queries per chunk: few in list mode, so failed queries spread over many waves
  const int64_t n_chunks = (n_queries + cq - 1) / cq;
  const int64_t tiles_total = a.Mp / STEP;
  const int64_t tiles_per_slice = sliced ? (tiles_total + a.slices - 1) / a.slices : tiles_total;
  const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
  const int64_t total_waves = (int64_t)gridDim.x * WAVES;

  for (int64_t chunk = wave_global; chunk < n_chunks; chunk += total_waves) {  // no block-level barriers inside
    const int64_t c0 = chunk * cq;
    const int nq = (n_queries - c0) < cq ? (int)(n_queries - c0) : cq;
    const int64_t pool_row0 = list_mode ? wave_global * cq : a.b_lo + c0;
    {  // load + transform this wave's queries: q = R0·s + t0 (SVGDICP.cpp:204)
      QSlot s;
      int64_t b = 0;
      if (lane < nq) {
        b = sliced ? (int64_t)a.qlist[c0 / a.slices] : list_mode ? (int64_t)a.qlist[c0 + lane] : a.b_lo + c0 + lane;
        const double sx = a.src[3 * b], sy = a.src[3 * b + 1], sz = a.src[3 * b + 2];
        const double* R = a.pose.R0;
        s.x = (sx * R[0] + sy * R[1] + sz * R[2]) + a.pose.t0[0];
        s.y = (sx * R[3] + sy * R[4] + sz * R[5]) + a.pose.t0[1];
        s.z = (sx * R[6] + sy * R[7] + sz * R[8]) + a.pose.t0[2];
        s.thr = (sliced && a.qthr) ? a.qthr[c0 / a.slices] : __builtin_huge_val();
      } else {
        s.x = s.y = s.z = 0.0;
        s.thr = -1.0;  // nothing passes d2 <= -1
      }
      qv[lane] = s;
      cnt[lane] = 0;
      qb[lane] = (int)b;
    }
    wave_sync();

    int64_t tile_lo = 0, tile_hi = a.Mp;
    if (sliced) {
      tile_lo = (c0 % a.slices) * tiles_per_slice * STEP;
      tile_hi = tile_lo + tiles_per_slice * STEP;
      if (tile_hi > a.Mp) tile_hi = a.Mp;
      if (tile_lo > tile_hi) tile_lo = tile_hi;
    }
    double x[T], y[T], z[T], nx[T], ny[T], nz[T];
    {
      const int64_t t0 = tile_lo < a.Mp ? tile_lo : 0;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        x[t] = a.tx[t0 + t * kWave + lane]; y[t] = a.ty[t0 + t * kWave + lane]; z[t] = a.tz[t0 + t * kWave + lane];
      }
    }
    for (int64_t tile = tile_lo; tile < tile_hi; tile += STEP) {
      const int64_t nt = (tile + STEP < tile_hi) ? tile + STEP : tile;  // prefetch next tile (or re-read last)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        nx[t] = a.tx[nt + t * kWave + lane]; ny[t] = a.ty[nt + t * kWave + lane]; nz[t] = a.tz[nt + t * kWave + lane];
      }
      for (int q = 0; q < nq; ++q) {
        const double2 q01 = *reinterpret_cast<const double2*>(&qv[q].x);
        const double2 q23 = *reinterpret_cast<const double2*>(&qv[q].z);
        const double qx = q01.x, qy = q01.y, qz = q23.x, thr = q23.y;
        double d[T];
        unsigned long long m[T];
        unsigned long long any = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const double dx = qx - x[t], dy = qy - y[t], dz = qz - z[t];
          d[t] = (dx * dx + dy * dy) + dz * dz;   // knn_cpu.cpp:43-50 order, unfused
          m[t] = __ballot(d[t] <= thr);
          any |= m[t];
        }
        if (any) {  // wave-uniform slow path
          const int64_t pool_base = (pool_row0 + q) * (int64_t)S;
          int n = __builtin_amdgcn_readfirstlane(cnt[q]);
#pragma unroll
          for (int t = 0; t < T; ++t) {
            if (m[t]) {
              if (n > S - kWave) {  // make room for up to 64 new entries
                if (lane == 0) cnt[q] = n;
                wave_sync();
                merge_pool(q, lane, K, S, qv, cnt, sd, si, a.pool_d, a.pool_i, pool_base);
                n = __builtin_amdgcn_readfirstlane(cnt[q]);
              }
              const unsigned int lo = (unsigned int)m[t], hi = (unsigned int)(m[t] >> 32);
              const int pos = n + (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
              if ((m[t] >> lane) & 1ull) {
                a.pool_d[pool_base + pos] = d[t];
                a.pool_i[pool_base + pos] = a.torig[tile + t * kWave + lane];  // original target index
              }
              n += __popcll(m[t]);
            }
          }
          if (lane == 0) cnt[q] = n;
          wave_sync();
        }
      }
#pragma unroll
      for (int t = 0; t < T; ++t) { x[t] = nx[t]; y[t] = ny[t]; z[t] = nz[t]; }
    }

    // final selection + output (ascending by (d2, idx); pad like torch::full(...,0), knn_cpu.cpp:25-26)
    for (int q = 0; q < nq; ++q) {
      const int64_t b = qb[q];
      merge_pool(q, lane, K, S, qv, cnt, sd, si, a.pool_d, a.pool_i, (pool_row0 + q) * (int64_t)S);
      const int n = __builtin_amdgcn_readfirstlane(cnt[q]);
      if (sliced) {  // partial result of this slice, padded with never-selected entries
        for (int e = lane; e < K; e += kWave) {
          const bool in = e < n;
          a.sl_d[c0 * K + e] = in ? sd[e] : __builtin_huge_val();
          a.sl_i[c0 * K + e] = in ? si[e] : 0x7fffffff;
        }
      } else {
        for (int e = lane; e < K; e += kWave) {
          const bool in = e < n;
          a.out_idx[b * K + e] = in ? si[e] : 0;
          a.out_d2[b * K + e] = in ? sd[e] : 0.0;
        }
      }
      wave_sync();
    }
  }
}

// sliced list mode, second half: K-way merge of the slices' sorted top-K lists of one failed query.  One wave per
// query, lane <-> slice: the lists sit in LDS, each lane holds the head of its list, and K rounds of a wave-wide
// (d², idx) minimum pop the result in order (padding as knn_cpu.cpp:25-26).
__global__ __launch_bounds__(64) void k_knn_merge_slices(KnnArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int n_listed = *a.qlist_count;
  if (n_listed > a.slice_max_queries) return;
  const int K = a.K, NS = a.slices, lane = threadIdx.x;
  const int n_in = NS * K;
  double* sd = reinterpret_cast<double*>(smem);
  int* si = reinterpret_cast<int*>(smem + sizeof(double) * (size_t)n_in);
  for (int f = blockIdx.x; f < n_listed; f += gridDim.x) {
    const int64_t b = a.qlist[f];
    wave_sync();
    for (int e = lane; e < n_in; e += kWave) { sd[e] = a.sl_d[(int64_t)f * n_in + e]; si[e] = a.sl_i[(int64_t)f * n_in + e]; }
    wave_sync();
    int pos = 0;
    double hd = lane < NS ? sd[lane * K] : __builtin_huge_val();
    int hi = lane < NS ? si[lane * K] : 0x7fffffff;
    for (int r = 0; r < K; ++r) {
      double bd = hd;
      int bi = hi, bl = lane;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(bd, off, kWave);
        const int oi = __shfl_xor(bi, off, kWave), ol = __shfl_xor(bl, off, kWave);
        // exhausted heads are (inf, INT_MAX); equal (d, idx) cannot come from two slices (distinct targets)
        if (ent_less(od, oi, bd, bi) || (od == bd && oi == bi && ol < bl)) { bd = od; bi = oi; bl = ol; }
      }
      if (lane == 0) {
        const bool in = bi != 0x7fffffff;
        a.out_idx[b * K + r] = in ? bi : 0;
        a.out_d2[b * K + r] = in ? bd : 0.0;
      }
      if (lane == bl && bi != 0x7fffffff) {
        ++pos;
        hd = pos < K ? sd[lane * K + pos] : __builtin_huge_val();
        hi = pos < K ? si[lane * K + pos] : 0x7fffffff;
      }
    }
  }
}

// target_batch = index_select(target, sourceKNN_idx) (SVGDICP.cpp:191-193), ONE copy [B][K][3]
__global__ void k_build_table(const int32_t* __restrict__ idx, int64_t n_entries, const double* __restrict__ tgt,
                              double* __restrict__ table) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const int64_t i = idx[e];
  table[3 * e] = tgt[3 * i];
  table[3 * e + 1] = tgt[3 * i + 1];
  table[3 * e + 2] = tgt[3 * i + 2];
}

}  // namespace

int knn_pool_size(int K) {
  int S = 256;
  while (S < K + 128) S <<= 1;
  return S;
}
int64_t knn_padded_targets(int64_t M) { return ((M + 511) / 512) * 512; }  // multiple of both kernels' steps

hipError_t launch_knn_topk(const KnnArgs& a, hipStream_t st) {
  int64_t nb;
  if (a.qlist) {
    nb = a.list_grid;  // fixed grid; the kernel reads the list length from device memory
  } else {
    const int64_t nq = a.b_hi - a.b_lo;
    if (nq <= 0) return hipSuccess;
    nb = (nq + (int64_t)QW * WAVES - 1) / ((int64_t)QW * WAVES);
  }
  const size_t per_wave = sizeof(QSlot) * QW + 2 * sizeof(int) * QW + (size_t)a.S * (sizeof(double) + sizeof(int));
  const size_t smem = per_wave * WAVES;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_topk),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_knn_topk, dim3((unsigned)nb), dim3(256), smem, st, a);
  return hipGetLastError();
}

int knn_slice_count(int K) {  // slices x K entries must fit the merge kernel's 8192-entry LDS sort
  int kp = 1;
  while (kp < K) kp <<= 1;
  int ns = 8192 / kp;
  return ns > 64 ? 64 : (ns < 1 ? 1 : ns);
}

hipError_t launch_knn_merge_slices(const KnnArgs& a, hipStream_t st) {
  const size_t smem = (size_t)a.slices * a.K * (sizeof(double) + sizeof(int));
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_merge_slices),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  const int nb = a.slice_max_queries < 512 ? a.slice_max_queries : 512;
  hipLaunchKernelGGL(k_knn_merge_slices, dim3(nb > 0 ? nb : 1), dim3(64), smem, st, a);
  return hipGetLastError();
}

hipError_t launch_build_table(const int32_t* idx, int64_t n_entries, const double* tgt, double* table,
                              hipStream_t st) {
  if (n_entries <= 0) return hipSuccess;
  const int64_t nb = (n_entries + 255) / 256;
  hipLaunchKernelGGL(k_build_table, dim3((unsigned)nb), dim3(256), 0, st, idx, n_entries, tgt, table);
  return hipGetLastError();
}

}  // namespace svnicp
