// voxel_map.hip — the local map of the scan-to-map loop, resident in HBM.
//
// SURVEY.md §8(f)-4 (second half).  Device counterpart of svnicp::VoxelHashMap
// (/root/reference/svn-icp/src/core/VoxelHashMap.cpp:22-101, include/core/VoxelHashMap.h): a hash map
// voxel -> at most max_points points in insertion order, with
//   AddPointCloud(cloud, pose)  (:22-42)  transform by the pose (float32 like pcl::PointXYZ), voxel index = coordinates /
//                                          voxel_size truncated toward zero (:29), append while the voxel has room, then
//   RemoveFarPointCloud(pos)    (:89-97)  drop every voxel whose FIRST point is farther than max_range,
//   GetMap(pose, r) / GetMap()  (:44-58)  all points of the voxels whose first point is closer than r (or of all voxels).
// The reference keeps it in a tsl::robin_map on the host and re-uploads the query result for every scan
// (OdometryPipeline.cpp:577-582); here the table lives in HBM and a query writes float64 rows straight into a device
// buffer the solver copies device-to-device (svnicp_set_target, SVNICP_MEM_DEVICE) — per scan only the new points go over
// PCIe.
//
// Layout: open addressing, linear probing; keys[cap] = packed voxel index (3 x 21 bits, offset 2^20) or EMPTY / TOMB;
// counts[cap]; pts[cap][max_points] float3.  Determinism: a point's slot is found (or created by atomicCAS) in parallel,
// but WHICH points a voxel keeps is decided in input order — the (slot, input index) pairs are radix-sorted (stable), a
// point's rank inside its voxel is its position in the sorted run, and it is stored at counts[slot] + rank if that is
// below max_points: exactly the points the sequential loop of the reference keeps, in the same order.  A query emits
// voxels in ascending (x, y, z) voxel index (radix sort of the selected keys), so the output is reproducible and equals
// the ordered host map of svn-icp_amd/host/registration_pipeline.hpp point for point.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/svnicp_hip.h"
#include "kernels.hpp"

namespace {

constexpr unsigned long long kEmpty = ~0ull;
constexpr unsigned long long kTomb = ~0ull - 1ull;
constexpr int kOff = 1 << 20;  // voxel indices in [-2^20, 2^20)

__device__ __forceinline__ unsigned long long hash_key(unsigned long long k) {  // splitmix64 finaliser
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return k;
}

struct MapPose { double R[9]; double t[3]; };

// transform + voxel key + find-or-create the voxel's slot
__global__ __launch_bounds__(256) void k_map_locate(const float* __restrict__ in, int64_t n, MapPose pose, float voxel,
                                                    unsigned long long* __restrict__ keys, int64_t cap,
                                                    float* __restrict__ q, unsigned int* __restrict__ slot_of,
                                                    int* __restrict__ stats) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p0 = in[3 * i], p1 = in[3 * i + 1], p2 = in[3 * i + 2];
  // pcl::transformPointCloud with gtsam's double Matrix4 (VoxelHashMap.cpp:25): every coordinate is formed in double from the
  // widened float32 point, left to right, and rounded ONCE to float32
  float c[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
    c[d] = (float)(((pose.R[3 * d] * (double)p0 + pose.R[3 * d + 1] * (double)p1) + pose.R[3 * d + 2] * (double)p2) + pose.t[d]);
  q[3 * i] = c[0]; q[3 * i + 1] = c[1]; q[3 * i + 2] = c[2];
  long long v[3];
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float f = truncf(c[d] / voxel);   // Eigen cast<int>: toward zero (VoxelHashMap.cpp:29)
    ok = ok && (f >= (float)-kOff) && (f < (float)kOff);   // also false for NaN
    v[d] = ok ? (long long)f : 0;
  }
  if (!ok) { slot_of[i] = 0xffffffffu; atomicAdd(&stats[3], 1); return; }   // outside the index range or NaN: counted, not stored
  const unsigned long long key = ((unsigned long long)(v[0] + kOff) << 42) | ((unsigned long long)(v[1] + kOff) << 21) |
                                 (unsigned long long)(v[2] + kOff);
  unsigned long long h = hash_key(key) & (unsigned long long)(cap - 1);
  for (int64_t probe = 0; probe < cap; ++probe) {   // bounded: a full table ends the loop and is reported
    unsigned long long cur = keys[h];
    if (cur == kEmpty) cur = atomicCAS(&keys[h], kEmpty, key);
    if (cur == key || cur == kEmpty) { slot_of[i] = (unsigned int)h; return; }
    h = (h + 1) & (unsigned long long)(cap - 1);
  }
  slot_of[i] = 0xffffffffu;
  atomicOr(&stats[2], 2);
}

__global__ __launch_bounds__(256) void k_map_iota(int* __restrict__ v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (int)i;
}

__device__ __forceinline__ int64_t lower_bound_u32(const unsigned int* a, int64_t n, unsigned int v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}

// sorted by (slot, input index): store the points that still fit, in input order
__global__ __launch_bounds__(256) void k_map_place(const unsigned int* __restrict__ sslot, const int* __restrict__ sidx, int64_t n,
                                                   const int* __restrict__ counts, int max_points, const float* __restrict__ q,
                                                   float* __restrict__ pts) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int s = sslot[j];
  if (s == 0xffffffffu) return;
  const int64_t rank = j - lower_bound_u32(sslot, n, s);
  const int64_t pos = (int64_t)counts[s] + rank;
  if (pos >= max_points) return;
  const int i = sidx[j];
  float* o = pts + ((size_t)s * max_points + pos) * 3;
  o[0] = q[3 * (size_t)i]; o[1] = q[3 * (size_t)i + 1]; o[2] = q[3 * (size_t)i + 2];
}

__global__ __launch_bounds__(256) void k_map_count(const unsigned int* __restrict__ sslot, int64_t n, int* __restrict__ counts,
                                                   int max_points, int* __restrict__ stats) {
  __shared__ int s_new;
  if (threadIdx.x == 0) s_new = 0;
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    const unsigned int s = sslot[j];
    if (s != 0xffffffffu && (j == 0 || sslot[j - 1] != s)) {   // one thread per run
      // a voxel next to the sensor takes thousands of points of one scan: the run's end by bisection, not by walking it
      int64_t lo = j + 1, hi = n;
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sslot[mid] <= s) lo = mid + 1; else hi = mid; }
      const int before = counts[s];
      const int64_t after = before + (lo - j);
      counts[s] = after > max_points ? max_points : (int)after;
      if (before == 0) atomicAdd(&s_new, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_new) atomicAdd(&stats[0], s_new);   // one global atomic per workgroup
}

// RemoveFarPointCloud (VoxelHashMap.cpp:89-97)
__global__ __launch_bounds__(256) void k_map_remove_far(unsigned long long* __restrict__ keys, int* __restrict__ counts,
                                                        const float* __restrict__ pts, int64_t cap, int max_points, double px,
                                                        double py, double pz, double r2, int* __restrict__ stats) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cap) return;
  const unsigned long long k = keys[s];
  if (k == kEmpty || k == kTomb || counts[s] <= 0) return;
  const float* f = pts + (size_t)s * max_points * 3;
  const double dx = (double)f[0] - px, dy = (double)f[1] - py, dz = (double)f[2] - pz;
  if (dx * dx + dy * dy + dz * dz > r2) {
    keys[s] = kTomb; counts[s] = 0;
    atomicSub(&stats[0], 1); atomicAdd(&stats[1], 1);
  }
}

// GetMap(pose, r) selection (VoxelHashMap.cpp:48-58); r2 < 0 selects every voxel (GetMap(), :44-46)
constexpr int kSelChunk = 16;   // slots per thread of k_map_select
__global__ __launch_bounds__(256) void k_map_select(const unsigned long long* __restrict__ keys, const int* __restrict__ counts,
                                                    const float* __restrict__ pts, int64_t cap, int max_points, double px, double py,
                                                    double pz, double r2, unsigned long long* __restrict__ sel_key,
                                                    unsigned int* __restrict__ sel_slot, int* __restrict__ nsel, int limit) {
  // a workgroup owns kSelChunk * 256 consecutive slots: it counts its selected voxels, reserves their output range with ONE
  // global atomic (4096 workgroups with one atomic each on the same address took 45 us) and writes them (any order: the
  // keys are sorted afterwards)
  __shared__ int s_base, s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int64_t s0 = (int64_t)blockIdx.x * (kSelChunk * 256) + threadIdx.x;
  unsigned int takes = 0;
  unsigned long long k[kSelChunk];
#pragma unroll
  for (int i = 0; i < kSelChunk; ++i) {
    const int64_t s = s0 + (int64_t)i * 256;
    k[i] = s < cap ? keys[s] : kEmpty;
  }
#pragma unroll
  for (int i = 0; i < kSelChunk; ++i) {
    const int64_t s = s0 + (int64_t)i * 256;
    if (k[i] != kEmpty && k[i] != kTomb && counts[s] > 0) {
      bool take = true;
      if (r2 >= 0.0) {
        const float* f = pts + (size_t)s * max_points * 3;
        const double dx = (double)f[0] - px, dy = (double)f[1] - py, dz = (double)f[2] - pz;
        take = dx * dx + dy * dy + dz * dz < r2;
      }
      if (take) takes |= 1u << i;
    }
  }
  int my = 0;
  if (takes) my = atomicAdd(&s_n, __popc(takes));
  __syncthreads();
  if (threadIdx.x == 0 && s_n > 0) s_base = atomicAdd(nsel, s_n);
  __syncthreads();
  int pos = s_base + my;
#pragma unroll
  for (int i = 0; i < kSelChunk; ++i) {
    if ((takes >> i) & 1u) {
      if (pos < limit) { sel_key[pos] = k[i]; sel_slot[pos] = (unsigned int)(s0 + (int64_t)i * 256); }   // limit = live voxels known to the host: never exceeded
      ++pos;
    }
  }
}

__global__ __launch_bounds__(256) void k_map_sel_counts(const unsigned long long* __restrict__ skey, const unsigned int* __restrict__ sslot, int nsel,
                                                        const int* __restrict__ counts, int* __restrict__ cnt_out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nsel) cnt_out[j] = skey[j] == kEmpty ? 0 : counts[sslot[j]];   // entries the selection did not fill sort to the end with key ~0
}

// voxel j of the sorted selection -> its points as float64 rows (ICPUtils.cpp:27-43 widening) at offs[j]
__global__ __launch_bounds__(256) void k_map_gather(const unsigned int* __restrict__ sslot, const int* __restrict__ offs,
                                                    const int* __restrict__ cnts, int nsel, int max_points,
                                                    const float* __restrict__ pts, double* __restrict__ out, float* __restrict__ out_f32) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (voxel, point slot)
  const int j = (int)(g / max_points), k = (int)(g % max_points);
  if (j >= nsel || k >= cnts[j]) return;
  const float* f = pts + ((size_t)sslot[j] * max_points + k) * 3;
  const size_t o = ((size_t)offs[j] + k) * 3;
  if (out) { out[o] = (double)f[0]; out[o + 1] = (double)f[1]; out[o + 2] = (double)f[2]; }
  if (out_f32) { out_f32[o] = f[0]; out_f32[o + 1] = f[1]; out_f32[o + 2] = f[2]; }
}

// rebuild without tombstones: re-insert every live voxel into a fresh table
__global__ __launch_bounds__(256) void k_map_rehash(const unsigned long long* __restrict__ okeys, const int* __restrict__ ocounts,
                                                    const float* __restrict__ opts, int64_t ocap, int max_points,
                                                    unsigned long long* __restrict__ keys, int* __restrict__ counts,
                                                    float* __restrict__ pts, int64_t cap, int* __restrict__ stats) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ocap) return;
  const unsigned long long key = okeys[s];
  if (key == kEmpty || key == kTomb || ocounts[s] <= 0) return;
  unsigned long long h = hash_key(key) & (unsigned long long)(cap - 1);
  for (int64_t probe = 0; probe < cap; ++probe) {
    if (atomicCAS(&keys[h], kEmpty, key) == kEmpty) {
      counts[h] = ocounts[s];
      const float* a = opts + (size_t)s * max_points * 3;
      float* b = pts + (size_t)h * max_points * 3;
      for (int i = 0; i < 3 * ocounts[s]; ++i) b[i] = a[i];
      return;
    }
    h = (h + 1) & (unsigned long long)(cap - 1);
  }
  atomicOr(&stats[2], 2);
}

template <typename T>
struct Buf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap && p) return hipSuccess;
    if (p) (void)hipFree(p);
    // grow by half beyond the request: the map gains voxels with every scan, and a hipFree + hipMalloc per call cost more
    // than the kernels of a query
    if (cap > 0) n += n / 2;
    p = nullptr; cap = 0;
    if (n == 0) n = 1;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct svnicp_map {
  int device = 0;
  hipStream_t stream = nullptr;
  double voxel = 1.0, max_range = 80.0;
  int max_points = 20;
  int64_t cap = 0;
  Buf<unsigned long long> keys, sel_key, sel_key2;
  Buf<int> counts, stats, sidx_in, sidx, sel_cnt, sel_off, nsel;
  Buf<float> pts, q, in, out_f32;
  Buf<unsigned int> slot, sslot, sel_slot, sel_slot2;
  Buf<double> out;
  Buf<unsigned char> tmp;
  int64_t last_M = 0;
  int64_t skipped = 0;   // points svnicp_map_add_cloud did not store (outside the index range or NaN), since creation / clear
  int h_stats[4] = {0, 0, 0, 0};
  std::string err;
};

namespace {
thread_local std::string g_map_error;
int mfail(svnicp_map* m, int code, const std::string& msg) { if (m) m->err = msg; else g_map_error = msg; return code; }
#define MCHK(m, expr)                                                                                      \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess)                                                                                  \
      return mfail((m), _e == hipErrorOutOfMemory ? SVNICP_ERR_NOMEM : SVNICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

int alloc_table(svnicp_map* m, int64_t cap, Buf<unsigned long long>& keys, Buf<int>& counts, Buf<float>& pts) {
  MCHK(m, keys.ensure((size_t)cap));
  MCHK(m, counts.ensure((size_t)cap));
  MCHK(m, pts.ensure((size_t)cap * m->max_points * 3));
  MCHK(m, hipMemsetAsync(keys.p, 0xff, (size_t)cap * 8, m->stream));
  MCHK(m, hipMemsetAsync(counts.p, 0, (size_t)cap * 4, m->stream));
  return 0;
}

int read_stats(svnicp_map* m) {
  MCHK(m, hipMemcpyAsync(m->h_stats, m->stats.p, sizeof m->h_stats, hipMemcpyDeviceToHost, m->stream));
  MCHK(m, hipStreamSynchronize(m->stream));
  return 0;
}

int rebuild(svnicp_map* m, int64_t new_cap) {
  Buf<unsigned long long> nk; Buf<int> nc; Buf<float> np;
  int rc = alloc_table(m, new_cap, nk, nc, np);
  if (rc) { nk.release(); nc.release(); np.release(); return rc; }
  hipLaunchKernelGGL(k_map_rehash, dim3((unsigned)((m->cap + 255) / 256)), dim3(256), 0, m->stream, m->keys.p, m->counts.p, m->pts.p,
                     m->cap, m->max_points, nk.p, nc.p, np.p, new_cap, m->stats.p);
  MCHK(m, hipGetLastError());
  const int zero = 0;
  MCHK(m, hipMemcpyAsync(m->stats.p + 1, &zero, sizeof(int), hipMemcpyHostToDevice, m->stream));   // no tombstones left
  MCHK(m, hipStreamSynchronize(m->stream));
  m->keys.release(); m->counts.release(); m->pts.release();
  m->keys = nk; m->counts = nc; m->pts = np;
  m->cap = new_cap;
  return 0;
}
}  // namespace

extern "C" {

const char* svnicp_map_last_error(const svnicp_map* m) { return m ? m->err.c_str() : g_map_error.c_str(); }

int svnicp_map_create(int device, double voxel_size, double max_range, int max_points, int64_t capacity_voxels, svnicp_map** out) {
  if (!out || !(voxel_size > 0) || max_points < 1 || max_points > 256)
    return mfail(nullptr, SVNICP_ERR_INVALID, "svnicp_map_create: need voxel_size > 0 and 1 <= max_points <= 256");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mfail(nullptr, SVNICP_ERR_NO_DEVICE, "svnicp_map_create: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return mfail(nullptr, SVNICP_ERR_INVALID, "svnicp_map_create: bad device ordinal");
  if (hipSetDevice(device) != hipSuccess) return mfail(nullptr, SVNICP_ERR_HIP, "hipSetDevice failed");
  svnicp_map* m = new svnicp_map();
  m->device = device; m->voxel = voxel_size; m->max_range = max_range; m->max_points = max_points;
  int64_t cap = 1 << 16;
  const int64_t want = capacity_voxels > 0 ? capacity_voxels : (int64_t)1 << 20;
  while (cap < want) cap <<= 1;
  m->cap = cap;
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return mfail(nullptr, SVNICP_ERR_HIP, "hipStreamCreate failed"); }
  int rc = alloc_table(m, cap, m->keys, m->counts, m->pts);
  if (!rc && (m->stats.ensure(4) != hipSuccess || m->nsel.ensure(1) != hipSuccess)) rc = SVNICP_ERR_NOMEM;
  if (!rc && hipMemsetAsync(m->stats.p, 0, 16, m->stream) != hipSuccess) rc = SVNICP_ERR_HIP;
  if (!rc && hipStreamSynchronize(m->stream) != hipSuccess) rc = SVNICP_ERR_HIP;
  if (rc) { g_map_error = m->err.empty() ? "svnicp_map_create: allocation failed" : m->err; svnicp_map_destroy(m); return rc; }
  *out = m;
  return SVNICP_OK;
}

void svnicp_map_destroy(svnicp_map* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  m->keys.release(); m->sel_key.release(); m->sel_key2.release(); m->counts.release(); m->stats.release(); m->sidx_in.release();
  m->sidx.release(); m->sel_cnt.release(); m->sel_off.release(); m->nsel.release(); m->pts.release(); m->q.release(); m->in.release();
  m->out_f32.release(); m->slot.release(); m->sslot.release(); m->sel_slot.release(); m->sel_slot2.release(); m->out.release(); m->tmp.release();
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

int svnicp_map_clear(svnicp_map* m) {
  if (!m) return SVNICP_ERR_INVALID;
  MCHK(m, hipSetDevice(m->device));
  MCHK(m, hipMemsetAsync(m->keys.p, 0xff, (size_t)m->cap * 8, m->stream));
  MCHK(m, hipMemsetAsync(m->counts.p, 0, (size_t)m->cap * 4, m->stream));
  MCHK(m, hipMemsetAsync(m->stats.p, 0, 16, m->stream));
  MCHK(m, hipStreamSynchronize(m->stream));
  std::memset(m->h_stats, 0, sizeof m->h_stats);
  m->skipped = 0;
  return SVNICP_OK;
}

int svnicp_map_skipped_points(svnicp_map* m, int64_t* out) {
  if (!m || !out) return SVNICP_ERR_INVALID;
  *out = m->skipped;
  return SVNICP_OK;
}

int svnicp_map_size(svnicp_map* m, int64_t* voxels) {
  if (!m || !voxels) return SVNICP_ERR_INVALID;
  MCHK(m, hipSetDevice(m->device));
  const int rc = read_stats(m);
  if (rc) return rc;
  *voxels = m->h_stats[0];
  return SVNICP_OK;
}

int svnicp_map_add_cloud(svnicp_map* m, const float* xyz, int64_t n, int mem_kind, const double R_rowmajor[9], const double t[3]) {
  if (!m || !R_rowmajor || !t || n < 0 || (n > 0 && !xyz) || n > 0x7fffffffLL) return mfail(m, SVNICP_ERR_INVALID, "svnicp_map_add_cloud: bad argument");
  MCHK(m, hipSetDevice(m->device));
  if (n > 0) {
    // room for this cloud in the worst case (every point a new voxel): keep the load factor below 1/2, clear tombstones
    int rc = read_stats(m);
    if (rc) return rc;
    int64_t need = m->cap;
    while ((int64_t)(m->h_stats[0] + n) * 2 > need) need <<= 1;
    if (need != m->cap || (int64_t)m->h_stats[1] * 4 > m->cap) { rc = rebuild(m, need); if (rc) return rc; }
    const float* din = xyz;
    if (mem_kind != SVNICP_MEM_DEVICE) {
      MCHK(m, m->in.ensure((size_t)n * 3));
      MCHK(m, hipMemcpyAsync(m->in.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, m->stream));
      din = m->in.p;
    }
    MCHK(m, m->q.ensure((size_t)n * 3)); MCHK(m, m->slot.ensure((size_t)n)); MCHK(m, m->sslot.ensure((size_t)n));
    MCHK(m, m->sidx_in.ensure((size_t)n)); MCHK(m, m->sidx.ensure((size_t)n));
    MapPose ps;
    for (int i = 0; i < 9; ++i) ps.R[i] = R_rowmajor[i];
    for (int i = 0; i < 3; ++i) ps.t[i] = t[i];
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_map_locate, dim3(g), dim3(256), 0, m->stream, din, n, ps, (float)m->voxel, m->keys.p, m->cap, m->q.p, m->slot.p, m->stats.p);
    MCHK(m, hipGetLastError());
    {  // stable sort of (slot, input index): a voxel's new points become a run in input order
      hipLaunchKernelGGL(k_map_iota, dim3(g), dim3(256), 0, m->stream, m->sidx_in.p, n);
      MCHK(m, hipGetLastError());
      size_t bytes = 0;
      MCHK(m, rocprim::radix_sort_pairs(nullptr, bytes, m->slot.p, m->sslot.p, m->sidx_in.p, m->sidx.p, (size_t)n, 0, 32, m->stream));
      MCHK(m, m->tmp.ensure(bytes));
      MCHK(m, rocprim::radix_sort_pairs(m->tmp.p, bytes, m->slot.p, m->sslot.p, m->sidx_in.p, m->sidx.p, (size_t)n, 0, 32, m->stream));
    }
    hipLaunchKernelGGL(k_map_place, dim3(g), dim3(256), 0, m->stream, m->sslot.p, m->sidx.p, n, m->counts.p, m->max_points, m->q.p, m->pts.p);
    MCHK(m, hipGetLastError());
    hipLaunchKernelGGL(k_map_count, dim3(g), dim3(256), 0, m->stream, m->sslot.p, n, m->counts.p, m->max_points, m->stats.p);
    MCHK(m, hipGetLastError());
  }
  hipLaunchKernelGGL(k_map_remove_far, dim3((unsigned)((m->cap + 255) / 256)), dim3(256), 0, m->stream, m->keys.p, m->counts.p, m->pts.p,
                     m->cap, m->max_points, t[0], t[1], t[2], m->max_range * m->max_range, m->stats.p);
  MCHK(m, hipGetLastError());
  const int rc = read_stats(m);
  if (rc) return rc;
  if (m->h_stats[2] || m->h_stats[3]) {
    const int flags = m->h_stats[2], zero2[2] = {0, 0};
    m->skipped += m->h_stats[3];    // points outside +-2^20 voxels or NaN: not stored, not an error — the map and its
                                    // counters are consistent, the caller's drive goes on (svnicp_map_skipped_points)
    MCHK(m, hipMemcpyAsync(m->stats.p + 2, zero2, sizeof zero2, hipMemcpyHostToDevice, m->stream));
    MCHK(m, hipStreamSynchronize(m->stream));
    // cannot happen: the table is grown to twice the voxels this cloud could add before anything is inserted
    if (flags & 2) return mfail(m, SVNICP_ERR_NOMEM, "svnicp_map_add_cloud: hash table full");
  }
  return SVNICP_OK;
}

int svnicp_map_query(svnicp_map* m, const double center[3], double max_range, int64_t* count_out) {
  if (!m || !count_out) return SVNICP_ERR_INVALID;
  MCHK(m, hipSetDevice(m->device));
  const double r2 = (center && max_range >= 0.0) ? max_range * max_range : -1.0;
  const double c0 = center ? center[0] : 0.0, c1 = center ? center[1] : 0.0, c2 = center ? center[2] : 0.0;
  // Everything is sized by the number of live voxels the host already knows (every call that changes the map ends by
  // reading the statistics), so the whole query is queued without looking at intermediate counts and synchronises once:
  // unselected entries keep the key ~0, sort to the end and count zero points.
  const size_t live = (size_t)(m->h_stats[0] > 0 ? m->h_stats[0] : 0);
  *count_out = 0; m->last_M = 0;
  if (live == 0) return SVNICP_OK;
  MCHK(m, m->sel_key.ensure(live)); MCHK(m, m->sel_key2.ensure(live)); MCHK(m, m->sel_slot.ensure(live)); MCHK(m, m->sel_slot2.ensure(live));
  MCHK(m, m->sel_cnt.ensure(live)); MCHK(m, m->sel_off.ensure(live));
  MCHK(m, m->out.ensure(live * (size_t)m->max_points * 3));
  size_t b1 = 0, b2 = 0;
  MCHK(m, rocprim::radix_sort_pairs(nullptr, b1, m->sel_key.p, m->sel_key2.p, m->sel_slot.p, m->sel_slot2.p, live, 0, 64, m->stream));
  MCHK(m, rocprim::exclusive_scan(nullptr, b2, m->sel_cnt.p, m->sel_off.p, 0, live, rocprim::plus<int>(), m->stream));
  MCHK(m, m->tmp.ensure(b1 > b2 ? b1 : b2));
  MCHK(m, hipMemsetAsync(m->nsel.p, 0, sizeof(int), m->stream));
  MCHK(m, hipMemsetAsync(m->sel_key.p, 0xff, live * sizeof(unsigned long long), m->stream));
  MCHK(m, hipMemsetAsync(m->sel_slot.p, 0, live * sizeof(unsigned int), m->stream));
  hipLaunchKernelGGL(k_map_select, dim3((unsigned)((m->cap + kSelChunk * 256 - 1) / (kSelChunk * 256))), dim3(256), 0, m->stream, m->keys.p, m->counts.p, m->pts.p, m->cap,
                     m->max_points, c0, c1, c2, r2, m->sel_key.p, m->sel_slot.p, m->nsel.p, (int)live);
  MCHK(m, hipGetLastError());
  MCHK(m, rocprim::radix_sort_pairs(m->tmp.p, b1, m->sel_key.p, m->sel_key2.p, m->sel_slot.p, m->sel_slot2.p, live, 0, 64, m->stream));
  const int nl = (int)live;
  hipLaunchKernelGGL(k_map_sel_counts, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, m->stream, m->sel_key2.p, m->sel_slot2.p, nl, m->counts.p, m->sel_cnt.p);
  MCHK(m, hipGetLastError());
  MCHK(m, rocprim::exclusive_scan(m->tmp.p, b2, m->sel_cnt.p, m->sel_off.p, 0, live, rocprim::plus<int>(), m->stream));
  const int64_t work = (int64_t)live * m->max_points;
  hipLaunchKernelGGL(k_map_gather, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, m->stream, m->sel_slot2.p, m->sel_off.p, m->sel_cnt.p, nl,
                     m->max_points, m->pts.p, m->out.p, (float*)nullptr);
  MCHK(m, hipGetLastError());
  int last[2] = {0, 0};
  MCHK(m, hipMemcpyAsync(&last[0], m->sel_off.p + (live - 1), sizeof(int), hipMemcpyDeviceToHost, m->stream));
  MCHK(m, hipMemcpyAsync(&last[1], m->sel_cnt.p + (live - 1), sizeof(int), hipMemcpyDeviceToHost, m->stream));
  MCHK(m, hipStreamSynchronize(m->stream));   // the rows are complete when the call returns (another stream may read them)
  const int64_t M = (int64_t)last[0] + last[1];
  m->last_M = M;
  *count_out = M;
  return SVNICP_OK;
}

void* svnicp_map_points_devptr(svnicp_map* m) { return m ? (void*)m->out.p : nullptr; }

int svnicp_map_download(svnicp_map* m, double* out_xyz, int64_t cap_points, int64_t* n_out) {
  if (!m || !n_out) return SVNICP_ERR_INVALID;
  MCHK(m, hipSetDevice(m->device));
  *n_out = m->last_M;
  const int64_t n = m->last_M < cap_points ? m->last_M : cap_points;
  if (n > 0 && out_xyz) {
    MCHK(m, hipMemcpyAsync(out_xyz, m->out.p, (size_t)n * 24, hipMemcpyDeviceToHost, m->stream));
    MCHK(m, hipStreamSynchronize(m->stream));
  }
  return SVNICP_OK;
}

}  // extern "C"
