// scan_prep.hip — the per-scan pre-processing of the scan-to-map loop, on the device.
//
// what OdometryPipeline::ICP_processing does to a scan before the solver sees it
// (/root/reference/svn-icp/src/core/OdometryPipeline.cpp):
//   crop_pointcloud      (:692-704)  keep min_range² < |p|² < max_range²; scan_max_range_ = largest SQUARED norm seen (:699)
//   uniform down-sample  (:684-690)  pcl::UniformSampling, leaf 0.5·voxel  -> the cloud that goes into the local map (:559)
//   uniform down-sample              leaf 1.5·voxel of THAT cloud          -> the source cloud of the registration (:560)
// The reference runs these with PCL on the host and uploads the results; here the raw float32 scan is uploaded once and the
// three clouds stay in HBM: the cropped and map clouds feed svnicp_map_add_cloud(…, SVNICP_MEM_DEVICE), the source cloud
// (float64 rows) feeds svnicp_set_source(…, SVNICP_MEM_DEVICE).
//
// UniformSampling as restated in svn-icp_amd/host/registration_pipeline.hpp (the cross-check of the tests): a grid of leaf
// size r anchored at floor(min/r); per occupied leaf the point closest to the leaf centre survives, the first one in input
// order on ties; leaves are emitted in ascending linear index.  Device form: leaf keys -> stable radix sort of (leaf, input
// index) -> runs; per run the smallest squared distance (64-bit atomicMin on the bits of a non-negative double), then the
// lowest sorted position among the points that attain it (stable sort: lowest input index) — the same winner as the
// sequential loop, independent of scheduling.  Same arithmetic, same order of additions as the host code (float32 points
// widened to float64).  HBM-bound integer/byte work; nothing here touches the matrix pipe.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/svnicp_hip.h"
#include "kernels.hpp"

namespace {

template <typename T>
struct PBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap && p) return hipSuccess;
    if (p) (void)hipFree(p);
    if (cap > 0) n += n / 2;   // scans vary in size: do not re-allocate for every small growth
    p = nullptr; cap = 0;
    if (n == 0) n = 1;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

__device__ __forceinline__ unsigned long long enc_f64(double v) {   // order-preserving double -> uint64
  const long long b = __double_as_longlong(v);
  return b < 0 ? ~(unsigned long long)b : ((unsigned long long)b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64(unsigned long long e) {
  const unsigned long long b = (e & 0x8000000000000000ull) ? (e & 0x7fffffffffffffffull) : ~e;
  return __longlong_as_double((long long)b);
}

// crop: keep flag per point, largest squared norm of ALL points (one atomic per workgroup)
__global__ __launch_bounds__(256) void k_prep_crop(const float* __restrict__ in, int64_t n, double min2, double max2, int* __restrict__ keep,
                                                   unsigned long long* __restrict__ max_n2) {
  __shared__ double s_m[4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double n2 = -1.0;
  if (i < n) {
    const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
    n2 = (double)((x * x + y * y) + z * z);           // float32, left to right, as pt.x*pt.x + pt.y*pt.y + pt.z*pt.z of
                                                      // OdometryPipeline.cpp:698 (this library is built with -ffp-contract=off)
    keep[i] = (n2 < max2 && n2 > min2) ? 1 : 0;
    if (!(n2 == n2)) n2 = -1.0;                       // a NaN point is dropped and does not count for the range
  }
  double m = n2;
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3]));
    if (m >= 0.0) atomicMax(max_n2, enc_f64(m));
  }
}

__global__ __launch_bounds__(256) void k_prep_compact(const float* __restrict__ in, int64_t n, const int* __restrict__ keep, const int* __restrict__ off,
                                                      float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !keep[i]) return;
  const size_t o = (size_t)off[i] * 3;
  out[o] = in[3 * i]; out[o + 1] = in[3 * i + 1]; out[o + 2] = in[3 * i + 2];
}

// grid bounds of a cloud: min / max of floor(p / leaf) per axis; bounds[0..2] = min, [3..5] = max (as long long)
__global__ __launch_bounds__(256) void k_ds_bounds(const float* __restrict__ in, int64_t n, double inv, long long* __restrict__ bounds) {
  __shared__ long long s_lo[4][3], s_hi[4][3];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  long long lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
  if (i < n) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { const long long c = (long long)floor((double)in[3 * i + d] * inv); lo[d] = c; hi[d] = c; }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d)
    for (int off = 32; off > 0; off >>= 1) {
      const long long a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
      lo[d] = a < lo[d] ? a : lo[d]; hi[d] = b > hi[d] ? b : hi[d];
    }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { s_lo[wave][d] = lo[d]; s_hi[wave][d] = hi[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    long long l = s_lo[0][d], h = s_hi[0][d];
    for (int w = 1; w < 4; ++w) { l = s_lo[w][d] < l ? s_lo[w][d] : l; h = s_hi[w][d] > h ? s_hi[w][d] : h; }
    if (l <= h) { atomicMin(&bounds[d], l); atomicMax(&bounds[3 + d], h); }
  }
}

// leaf key, squared distance to the leaf centre, input index
__global__ __launch_bounds__(256) void k_ds_keys(const float* __restrict__ in, int64_t n, double radius, double inv, const long long* __restrict__ bounds,
                                                 unsigned long long* __restrict__ key, unsigned long long* __restrict__ d2bits, int* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long dx = bounds[3] - bounds[0] + 1, dy = bounds[4] - bounds[1] + 1;
  long long ijk[3];
  double d2 = 0.0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {   // registration_pipeline.hpp: downsample_uniform, same expressions
    const double v = (double)in[3 * i + d];
    ijk[d] = (long long)floor(v * inv) - bounds[d];
    const double c = ((double)(ijk[d] + bounds[d]) + 0.5) * radius;
    d2 += (v - c) * (v - c);
  }
  key[i] = (unsigned long long)(ijk[0] + ijk[1] * dx + ijk[2] * dx * dy);
  d2bits[i] = (unsigned long long)__double_as_longlong(d2);   // d2 >= 0 (or NaN): the bit pattern orders like the value
  idx[i] = (int)i;
}

__global__ __launch_bounds__(256) void k_ds_run_flags(const unsigned long long* __restrict__ skey, int64_t n, int* __restrict__ flag) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) flag[j] = (j == 0 || skey[j] != skey[j - 1]) ? 1 : 0;
}

// run id of sorted position j = (exclusive prefix of the flags) + flag - 1
__global__ __launch_bounds__(256) void k_ds_min_d2(const int* __restrict__ flag, const int* __restrict__ pre, const int* __restrict__ sidx,
                                                   const unsigned long long* __restrict__ d2bits, int64_t n, unsigned long long* __restrict__ run_min) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) atomicMin(&run_min[pre[j] + flag[j] - 1], d2bits[sidx[j]]);
}
__global__ __launch_bounds__(256) void k_ds_min_pos(const int* __restrict__ flag, const int* __restrict__ pre, const int* __restrict__ sidx,
                                                    const unsigned long long* __restrict__ d2bits, int64_t n,
                                                    const unsigned long long* __restrict__ run_min, int* __restrict__ run_pos) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int r = pre[j] + flag[j] - 1;
  if (d2bits[sidx[j]] == run_min[r]) atomicMin(&run_pos[r], (int)j);
}
__global__ __launch_bounds__(256) void k_ds_gather(const float* __restrict__ in, const int* __restrict__ sidx, const int* __restrict__ run_pos, int runs,
                                                   float* __restrict__ out, double* __restrict__ out64) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= runs) return;
  const int p = run_pos[r];
  const int i = sidx[p];
  const float x = in[3 * (size_t)i], y = in[3 * (size_t)i + 1], z = in[3 * (size_t)i + 2];
  out[3 * (size_t)r] = x; out[3 * (size_t)r + 1] = y; out[3 * (size_t)r + 2] = z;
  if (out64) { out64[3 * (size_t)r] = (double)x; out64[3 * (size_t)r + 1] = (double)y; out64[3 * (size_t)r + 2] = (double)z; }   // ICPUtils.cpp:27-43
}

std::string g_prep_error;

}  // namespace

struct svnicp_prep {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  PBuf<float> in, cropped, map_cloud, source;
  PBuf<double> source64;
  PBuf<int> keep, off, idx, sidx, flag, pre, run_pos;
  PBuf<unsigned long long> key, skey, d2bits, run_min, scal;   // scal: [0] max squared norm (encoded) [1..6] grid bounds
  PBuf<char> tmp;
  int64_t n_cropped = 0, n_map = 0, n_source = 0;
};

namespace {

int pfail(svnicp_prep* p, int code, const std::string& msg) { if (p) p->err = msg; else g_prep_error = msg; return code; }
#define PCHK(p, expr)                                                                                                       \
  do {                                                                                                                      \
    const hipError_t _e = (expr);                                                                                           \
    if (_e != hipSuccess)                                                                                                   \
      return pfail((p), _e == hipErrorOutOfMemory ? SVNICP_ERR_NOMEM : SVNICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// out (+ out64) = UniformSampling(in[0..n), radius); *n_out = number of occupied leaves
int downsample(svnicp_prep* p, const float* in, int64_t n, double radius, PBuf<float>& out, PBuf<double>* out64, int64_t* n_out) {
  *n_out = 0;
  if (n <= 0) return SVNICP_OK;
  if (!(radius > 0.0)) {   // the host code returns the cloud unchanged
    PCHK(p, out.ensure((size_t)n * 3));
    PCHK(p, hipMemcpyAsync(out.p, in, (size_t)n * 12, hipMemcpyDeviceToDevice, p->stream));
    *n_out = n;
    return SVNICP_OK;
  }
  const double inv = 1.0 / radius;
  const unsigned g = (unsigned)((n + 255) / 256);
  PCHK(p, p->key.ensure((size_t)n)); PCHK(p, p->skey.ensure((size_t)n)); PCHK(p, p->d2bits.ensure((size_t)n));
  PCHK(p, p->idx.ensure((size_t)n)); PCHK(p, p->sidx.ensure((size_t)n)); PCHK(p, p->flag.ensure((size_t)n)); PCHK(p, p->pre.ensure((size_t)n));
  PCHK(p, p->run_min.ensure((size_t)n)); PCHK(p, p->run_pos.ensure((size_t)n));
  long long* bounds = reinterpret_cast<long long*>(p->scal.p + 1);
  const long long init_b[6] = {INT64_MAX, INT64_MAX, INT64_MAX, INT64_MIN, INT64_MIN, INT64_MIN};
  PCHK(p, hipMemcpyAsync(bounds, init_b, sizeof init_b, hipMemcpyHostToDevice, p->stream));
  hipLaunchKernelGGL(k_ds_bounds, dim3(g), dim3(256), 0, p->stream, in, n, inv, bounds);
  hipLaunchKernelGGL(k_ds_keys, dim3(g), dim3(256), 0, p->stream, in, n, radius, inv, bounds, p->key.p, p->d2bits.p, p->idx.p);
  PCHK(p, hipGetLastError());
  size_t b1 = 0, b2 = 0;
  PCHK(p, rocprim::radix_sort_pairs(nullptr, b1, p->key.p, p->skey.p, p->idx.p, p->sidx.p, (size_t)n, 0, 64, p->stream));
  PCHK(p, rocprim::exclusive_scan(nullptr, b2, p->flag.p, p->pre.p, 0, (size_t)n, rocprim::plus<int>(), p->stream));
  PCHK(p, p->tmp.ensure(b1 > b2 ? b1 : b2));
  PCHK(p, rocprim::radix_sort_pairs(p->tmp.p, b1, p->key.p, p->skey.p, p->idx.p, p->sidx.p, (size_t)n, 0, 64, p->stream));   // stable
  hipLaunchKernelGGL(k_ds_run_flags, dim3(g), dim3(256), 0, p->stream, p->skey.p, n, p->flag.p);
  PCHK(p, hipGetLastError());
  PCHK(p, rocprim::exclusive_scan(p->tmp.p, b2, p->flag.p, p->pre.p, 0, (size_t)n, rocprim::plus<int>(), p->stream));
  PCHK(p, hipMemsetAsync(p->run_min.p, 0xff, (size_t)n * 8, p->stream));
  PCHK(p, hipMemsetAsync(p->run_pos.p, 0x7f, (size_t)n * 4, p->stream));
  hipLaunchKernelGGL(k_ds_min_d2, dim3(g), dim3(256), 0, p->stream, p->flag.p, p->pre.p, p->sidx.p, p->d2bits.p, n, p->run_min.p);
  hipLaunchKernelGGL(k_ds_min_pos, dim3(g), dim3(256), 0, p->stream, p->flag.p, p->pre.p, p->sidx.p, p->d2bits.p, n, p->run_min.p, p->run_pos.p);
  PCHK(p, hipGetLastError());
  int last[2] = {0, 0};
  PCHK(p, hipMemcpyAsync(&last[0], p->pre.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipMemcpyAsync(&last[1], p->flag.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  const int runs = last[0] + last[1];
  PCHK(p, out.ensure((size_t)runs * 3));
  if (out64) PCHK(p, out64->ensure((size_t)runs * 3));
  hipLaunchKernelGGL(k_ds_gather, dim3((unsigned)((runs + 255) / 256)), dim3(256), 0, p->stream, in, p->sidx.p, p->run_pos.p, runs, out.p,
                     out64 ? out64->p : (double*)nullptr);
  PCHK(p, hipGetLastError());
  *n_out = runs;
  return SVNICP_OK;
}

}  // namespace

extern "C" {

int svnicp_prep_create(int device, svnicp_prep** out) {
  if (!out) return SVNICP_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    return pfail(nullptr, SVNICP_ERR_NO_DEVICE, "svnicp_prep_create: no HIP device visible (this library has no CPU path)");
  svnicp_prep* p = new svnicp_prep();
  p->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&p->stream) != hipSuccess || p->scal.ensure(8) != hipSuccess) {
    delete p;
    return pfail(nullptr, SVNICP_ERR_HIP, "svnicp_prep_create: stream / allocation failed");
  }
  *out = p;
  return SVNICP_OK;
}

void svnicp_prep_destroy(svnicp_prep* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  p->in.release(); p->cropped.release(); p->map_cloud.release(); p->source.release(); p->source64.release(); p->keep.release(); p->off.release();
  p->idx.release(); p->sidx.release(); p->flag.release(); p->pre.release(); p->run_pos.release(); p->key.release(); p->skey.release();
  p->d2bits.release(); p->run_min.release(); p->scal.release(); p->tmp.release();
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

const char* svnicp_prep_last_error(const svnicp_prep* p) { return p ? p->err.c_str() : g_prep_error.c_str(); }

int svnicp_prep_scan(svnicp_prep* p, const float* xyz, int64_t n, int mem_kind, double min_range, double max_range, double voxel_size,
                     double* scan_max_range, int64_t* n_cropped, int64_t* n_map, int64_t* n_source) {
  if (!p || !scan_max_range || !n_cropped || !n_map || !n_source || n < 0 || (n > 0 && !xyz) || n > 0x7fffffffLL)
    return pfail(p, SVNICP_ERR_INVALID, "svnicp_prep_scan: bad argument");
  PCHK(p, hipSetDevice(p->device));
  p->n_cropped = p->n_map = p->n_source = 0;
  *n_cropped = *n_map = *n_source = 0;
  if (n == 0) return SVNICP_OK;
  const float* din = xyz;
  if (mem_kind != SVNICP_MEM_DEVICE) {
    PCHK(p, p->in.ensure((size_t)n * 3));
    PCHK(p, hipMemcpyAsync(p->in.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, p->stream));
    din = p->in.p;
  }
  // ---- crop (:692-704)
  PCHK(p, p->keep.ensure((size_t)n)); PCHK(p, p->off.ensure((size_t)n));
  PCHK(p, hipMemsetAsync(p->scal.p, 0, 8, p->stream));   // encoded doubles are > 0 for every value >= -inf: 0 = nothing seen
  const unsigned g = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_prep_crop, dim3(g), dim3(256), 0, p->stream, din, n, min_range * min_range, max_range * max_range, p->keep.p, p->scal.p);
  PCHK(p, hipGetLastError());
  size_t b = 0;
  PCHK(p, rocprim::exclusive_scan(nullptr, b, p->keep.p, p->off.p, 0, (size_t)n, rocprim::plus<int>(), p->stream));
  PCHK(p, p->tmp.ensure(b));
  PCHK(p, rocprim::exclusive_scan(p->tmp.p, b, p->keep.p, p->off.p, 0, (size_t)n, rocprim::plus<int>(), p->stream));
  int last[2] = {0, 0};
  unsigned long long enc = 0;
  PCHK(p, hipMemcpyAsync(&last[0], p->off.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipMemcpyAsync(&last[1], p->keep.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipMemcpyAsync(&enc, p->scal.p, 8, hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  if (enc) {
    const unsigned long long bits = (enc & 0x8000000000000000ull) ? (enc & 0x7fffffffffffffffull) : ~enc;
    double m;
    std::memcpy(&m, &bits, 8);
    if (m > *scan_max_range) *scan_max_range = m;       // :699 (a squared norm, kept as the reference keeps it)
  }
  const int64_t nc = (int64_t)last[0] + last[1];
  PCHK(p, p->cropped.ensure((size_t)(nc > 0 ? nc : 1) * 3));
  hipLaunchKernelGGL(k_prep_compact, dim3(g), dim3(256), 0, p->stream, din, n, p->keep.p, p->off.p, p->cropped.p);
  PCHK(p, hipGetLastError());
  p->n_cropped = nc;
  // ---- the two uniform samplings (:559-560)
  int rc = downsample(p, p->cropped.p, nc, 0.5 * voxel_size, p->map_cloud, nullptr, &p->n_map);
  if (rc) return rc;
  rc = downsample(p, p->map_cloud.p, p->n_map, 1.5 * voxel_size, p->source, &p->source64, &p->n_source);
  if (rc) return rc;
  if (!(1.5 * voxel_size > 0.0) && p->n_source > 0) {   // unchanged cloud: still hand out float64 rows
    return pfail(p, SVNICP_ERR_INVALID, "svnicp_prep_scan: voxel_size must be positive");
  }
  PCHK(p, hipStreamSynchronize(p->stream));   // the clouds are complete when the call returns (other streams read them)
  *n_cropped = p->n_cropped; *n_map = p->n_map; *n_source = p->n_source;
  return SVNICP_OK;
}

const float* svnicp_prep_cropped_devptr(svnicp_prep* p) { return p ? p->cropped.p : nullptr; }
const float* svnicp_prep_map_cloud_devptr(svnicp_prep* p) { return p ? p->map_cloud.p : nullptr; }
const double* svnicp_prep_source_devptr(svnicp_prep* p) { return p ? p->source64.p : nullptr; }
const float* svnicp_prep_source_f32_devptr(svnicp_prep* p) { return p ? p->source.p : nullptr; }

int svnicp_prep_download(svnicp_prep* p, int which, float* out_xyz, int64_t cap_points, int64_t* n_out) {
  if (!p || !n_out || which < 0 || which > 2) return SVNICP_ERR_INVALID;
  PCHK(p, hipSetDevice(p->device));
  const int64_t n_all = which == 0 ? p->n_cropped : which == 1 ? p->n_map : p->n_source;
  const float* src = which == 0 ? p->cropped.p : which == 1 ? p->map_cloud.p : p->source.p;
  *n_out = n_all;
  const int64_t n = n_all < cap_points ? n_all : cap_points;
  if (n > 0 && out_xyz) {
    PCHK(p, hipMemcpyAsync(out_xyz, src, (size_t)n * 12, hipMemcpyDeviceToHost, p->stream));
    PCHK(p, hipStreamSynchronize(p->stream));
  }
  return SVNICP_OK;
}

}  // extern "C"
