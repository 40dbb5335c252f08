// particle_update.hip — the small per-iteration Stein step on the particle set, one workgroup.
//
// Replaces (SVN mode) the tail of SVNICP::stein_align per iteration (src/core/SVNICP.cpp:71-107):
//   Newton_grad_right's finalisation H + 1e-6·I, linalg::solve (SVNICP.cpp:149-162),
//   rotm_to_ypr_tensor / to_rotation_tensor (:166-215), rbf_hessian_kernel incl. torch::median (:254-266),
//   svgd_grad (:218-227) or svn_full_grad (:229-252), pose_update (:268-279), the early-stop test
//   (:95-101, evaluated on the device: no per-iteration host sync) and the particle history (:103-107).
// In the multi-GPU layout every GPU runs this kernel redundantly on ALL particles after the
// all-gather of the 22 raw sums per particle; identical inputs + identical code ⇒ identical state.
#include "kernels.hpp"
#include "update_single.hpp"
#include "stein_split_device.hpp"

namespace svnicp {

namespace {

constexpr int UT = 512;     // threads of the update workgroup (2 waves per SIMD: up to 256 VGPRs, no spills)
constexpr int KREG = 32;    // pairwise-distance keys a thread keeps in registers for the median (n <= KREG*UT)

// HBM workspace (doubles): H[P][36] b[P][6] N[P][6] x[P][6] phi[P][6] sq[P][P].  The update kernel
// works out of LDS copies of x, N, b (and H when it fits); HBM keeps H for the traces, sq only for
// particle counts whose P² keys do not fit the register budget.
struct Work {
  double *H, *b, *N, *x, *phi, *sq;
  __device__ Work(double* w, int P) {
    H = w; b = H + (size_t)P * 36; N = b + (size_t)P * 6; x = N + (size_t)P * 6; phi = x + (size_t)P * 6;
    sq = phi + (size_t)P * 6;
  }
};

// The 22 raw sums of particle p.  One rank (or particle sharding): the context's own record.  Source-row sharding
// (svnicp_set_row_shard): a.sums is the all-gathered [n_ranks][P][22] array of the ranks' partial records — rank r summed
// its own source rows — and every rank adds the same records in the same (rank) order, so the replicas stay bit-identical.
// Small registrations (api.hip: small chain): a.sums is the accumulate kernel's `partial` array itself — one record per
// workgroup, record stride sums_stride — added here in block order: no k_reduce_partials launch.
__device__ __forceinline__ void load_sums(const UpdateArgs& a, int p, double* s) {
  const double* rec = a.sums + (size_t)p * kNSums;
  const size_t stride = a.sums_stride ? (size_t)a.sums_stride : (size_t)a.P * kNSums;
#pragma unroll
  for (int i = 0; i < kNSums; ++i) s[i] = rec[i];
  for (int r = 1; r < a.n_ranks; ++r) {
    rec += stride;
#pragma unroll
    for (int i = 0; i < kNSums; ++i) s[i] += rec[i];
  }
}

// shared state of the exact-median selection
struct SelShared {
  unsigned int hist[256];
  unsigned long long prefix;
  unsigned int rank;
  int nan_flag;
  double h;
};
__device__ __forceinline__ void sel_init(SelShared* S, int P, int tid) {
  if (tid < 256) S->hist[tid] = 0;
  if (tid == 0) { S->nan_flag = 0; S->prefix = 0ull; S->rank = (unsigned int)(((size_t)P * P - 1) / 2); S->h = __builtin_nan(""); }
}

__device__ __forceinline__ double pair_sq(const double* lx, int i, int j) {  // SVNICP.cpp:257-260
  double s = 0.0;
#pragma unroll
  for (int d = 0; d < 6; ++d) { const double df = lx[i * 6 + d] - lx[j * 6 + d]; s += df * df; }
  return s;
}

// h = median(all P² pair distances) / log(P+1)  (SVNICP.cpp:254-262 / SVGDICP.cpp:464-471): exact lower
// median (torch::median) by an 8-pass radix select on the non-negative f64 bit patterns; keys stay in
// registers when P² <= KREG*UT, two barriers per pass, the 256-bin scan runs in wave 0.  Block-wide call.
template <int T>   // T: threads of the calling workgroup
__device__ __attribute__((noinline)) void rbf_bandwidth(const double* lx, int P, double* sq_global, SelShared* S, int tid, int lane, int wave) {
  const int n = P * P;
  const bool keys_in_regs = n <= KREG * T;
  const float invP = 1.0f / (float)P;
  unsigned long long key[KREG];
  if (keys_in_regs) {
#pragma unroll
    for (int i = 0; i < KREG; ++i) {
      const int e = i * T + tid;
      key[i] = ~0ull;
      if (e < n) {
        int r = (int)((float)e * invP);
        if (r * P > e) --r;
        if ((r + 1) * P <= e) ++r;
        const double s = pair_sq(lx, r, e - r * P);
        key[i] = (unsigned long long)__double_as_longlong(s);
        if (s != s) S->nan_flag = 1;
      }
    }
  } else {
    for (int e = tid; e < n; e += T) {
      const int r = e / P;
      const double s = pair_sq(lx, r, e - r * P);
      sq_global[e] = s;
      if (s != s) S->nan_flag = 1;
    }
    __syncthreads();
  }
  for (int pass = 7; pass >= 0; --pass) {
    const int shift = pass * 8;
    const unsigned long long pre = S->prefix;
    if (keys_in_regs) {
#pragma unroll
      for (int i = 0; i < KREG; ++i) {
        const unsigned long long k = key[i];
        if (i * T + tid < n && (pass == 7 || (k >> (shift + 8)) == pre)) atomicAdd(&S->hist[(k >> shift) & 255ull], 1u);
      }
    } else {
      for (int e = tid; e < n; e += T) {
        const unsigned long long k = (unsigned long long)__double_as_longlong(sq_global[e]);
        if (pass == 7 || (k >> (shift + 8)) == pre) atomicAdd(&S->hist[(k >> shift) & 255ull], 1u);
      }
    }
    __syncthreads();
    if (wave == 0) {
      unsigned int c[4], tot = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { c[i] = S->hist[4 * lane + i]; tot += c[i]; S->hist[4 * lane + i] = 0; }
      unsigned int incl = tot;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned int v = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += v;
      }
      unsigned int cum = incl - tot;  // elements in bins before mine
      const unsigned int rank = S->rank;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (rank >= cum && rank < cum + c[i]) {
          S->prefix = (pre << 8) | (unsigned long long)(4 * lane + i);
          S->rank = rank - cum;
        }
        cum += c[i];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double med = S->nan_flag ? __builtin_nan("") : __longlong_as_double((long long)S->prefix);
    S->h = med / log((double)(P + 1));
  }
  __syncthreads();
}

// threads cooperating on one particle in the Stein-direction phase (power of two, <= 64)
__device__ __forceinline__ int threads_per_particle(int P) {
  int tpp = 1;
  while (tpp < 64 && tpp * 2 * P <= UT) tpp <<= 1;
  return tpp;
}

__global__ __launch_bounds__(UT) void k_particle_update(UpdateArgs a) {
  if (a.ctl[0]) return;
  extern __shared__ __align__(16) double dyn[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const int P = a.P;
  Work w(a.work, P);
  double* lx = dyn;               // [P][6]  x = [t ; Log R]
  double* lN = lx + 6 * P;        // [P][6]  Newton step
  double* lb = lN + 6 * P;        // [P][6]  b
  double* lphi = lb + 6 * P;      // [P][6]  Stein direction
  const double* Hsrc = a.h_in_lds ? (lphi + 6 * P) : w.H;  // [P][36]
  double* lH = a.h_in_lds ? (lphi + 6 * P) : nullptr;
  __shared__ SelShared sel;
  __shared__ double sh_Hinv[36];
  __shared__ double sh_Hmean[36];
  __shared__ double sh_norm[UT / kWave];

  unsigned long long tdbg = a.dbg ? __builtin_readcyclecounter() : 0ull;
  auto stamp = [&](int i) {  // debug (SVNICP_DEBUG): cycles of thread 0 between phase boundaries
    if (!a.dbg || tid != 0) return;
    const unsigned long long now = __builtin_readcyclecounter();
    a.dbg[i] += now - tdbg;
    tdbg = now;
  };
  // ---- 1. per particle: H, b, Newton step, x = [t ; Log R] ----
  for (int p = tid; p < P; p += UT) {
    double Rc[9], H[36], b[6], LU[36], x6[6];
    int piv[6];
    mat3_mul(a.pose.R0, a.R + 9 * p, Rc);
    { double sm[kNSums]; load_sums(a, p, sm); finalize_Hb(sm, Rc, H, b); }
#pragma unroll
    for (int i = 0; i < 36; ++i) { LU[i] = H[i]; if (lH) lH[p * 36 + i] = H[i]; }
    if (!lH || a.trH) {
#pragma unroll
      for (int i = 0; i < 36; ++i) w.H[(size_t)p * 36 + i] = H[i];
    }
    const bool ok = lu6(LU, piv);
#pragma unroll
    for (int i = 0; i < 6; ++i) x6[i] = b[i];
    lu6_solve(LU, piv, x6);                                   // SVNICP.cpp:162
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      lb[p * 6 + i] = b[i];
      lN[p * 6 + i] = ok ? x6[i] : __builtin_nan("");
    }
    double lg[3];
    so3_log(a.R + 9 * p, lg);                                 // SVNICP.cpp:74-77
#pragma unroll
    for (int i = 0; i < 3; ++i) { lx[p * 6 + i] = a.t[3 * p + i]; lx[p * 6 + 3 + i] = lg[i]; }
  }
  sel_init(&sel, P, tid);
  __syncthreads();
  stamp(0);

  if (P > 1) {
    // ---- 2. mean Hessian (SVNICP.cpp:85) and its inverse, RBF bandwidth from the exact median ----
    if (!a.full_grad && tid < 36 * 8) {  // 8 lanes per entry, strided over particles, folded by shuffles
      const int e = tid >> 3, part = tid & 7;
      double s = 0.0;
      for (int p = part; p < P; p += 8) s += Hsrc[(size_t)p * 36 + e];
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) s += __shfl_xor(s, off, 8);
      if (part == 0) sh_Hmean[e] = s / P;
    }
    __syncthreads();
    if (!a.full_grad && wave == UT / kWave - 1 && lane < 6) {  // linalg::inv (SVNICP.cpp:225): one column per lane
      double LU[36], col[6];
      int piv[6];
#pragma unroll
      for (int i = 0; i < 36; ++i) LU[i] = sh_Hmean[i];
      const bool ok = lu6(LU, piv);
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = (r == lane) ? 1.0 : 0.0;
      lu6_solve(LU, piv, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) sh_Hinv[6 * r + lane] = ok ? col[r] : __builtin_nan("");
    }
    rbf_bandwidth<UT>(lx, P, w.sq, &sel, tid, lane, wave);
    stamp(1);
    const double h = sel.h;
    // ---- 4. Stein direction: TPP threads per particle split the sum over j, folded by shuffles;
    //         the pair distance is recomputed from LDS (bit-identical, cheaper than an HBM load) ----
    const int tpp = threads_per_particle(P);
    const int per_pass = UT / tpp;
    for (int base = 0; base < P; base += per_pass) {
      const int pi = base + tid / tpp, part = tid % tpp;
      const bool act = pi < P;
      double xi[6];
#pragma unroll
      for (int d = 0; d < 6; ++d) xi[d] = act ? lx[pi * 6 + d] : 0.0;
      if (!a.full_grad) {                                     // svgd_grad, SVNICP.cpp:218-227
        double g[6] = {0, 0, 0, 0, 0, 0}, kn[6] = {0, 0, 0, 0, 0, 0}, ks = 0.0;
        if (act)
          for (int j = part; j < P; j += tpp) {
            double df[6], sq = 0.0;
#pragma unroll
            for (int d = 0; d < 6; ++d) { df[d] = xi[d] - lx[j * 6 + d]; sq += df[d] * df[d]; }
            const double k = exp(-sq / h);
#pragma unroll
            for (int d = 0; d < 6; ++d) {
              g[d] += df[d] * k;
              kn[d] += k * (-lN[j * 6 + d]);
            }
            ks += k;
          }
        for (int off = tpp >> 1; off > 0; off >>= 1) {
#pragma unroll
          for (int d = 0; d < 6; ++d) { g[d] += __shfl_xor(g[d], off, kWave); kn[d] += __shfl_xor(kn[d], off, kWave); }
          ks += __shfl_xor(ks, off, kWave);
        }
        if (act && part == 0) {
#pragma unroll
          for (int d = 0; d < 6; ++d) g[d] = 2 / h * g[d];
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            double hg = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) hg += sh_Hinv[6 * r + c] * g[c];
            lphi[pi * 6 + r] = (kn[r] + hg) / ks;
          }
        }
      } else {                                                // svn_full_grad, SVNICP.cpp:229-252
        double Hm[36], u[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 36; ++e) Hm[e] = 0.0;
        if (act)
          for (int j = part; j < P; j += tpp) {
            double df[6], sq = 0.0;
#pragma unroll
            for (int d = 0; d < 6; ++d) { df[d] = xi[d] - lx[j * 6 + d]; sq += df[d] * df[d]; }
            const double k = exp(-sq / h);
            double g[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) g[d] = 2 / h * (df[d] * k);
            const double k2 = k * k;
            const double* Hj = Hsrc + (size_t)j * 36;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
              for (int c = 0; c < 6; ++c) Hm[6 * r + c] += k2 * Hj[6 * r + c] + g[r] * g[c];
              u[r] += k * (-lb[j * 6 + r]) + g[r];
            }
          }
        for (int off = tpp >> 1; off > 0; off >>= 1) {
#pragma unroll
          for (int e = 0; e < 36; ++e) Hm[e] += __shfl_xor(Hm[e], off, kWave);
#pragma unroll
          for (int r = 0; r < 6; ++r) u[r] += __shfl_xor(u[r], off, kWave);
        }
        if (act && part == 0) {
#pragma unroll
          for (int e = 0; e < 36; ++e) Hm[e] /= P;
#pragma unroll
          for (int r = 0; r < 6; ++r) u[r] /= P;
          int piv[6];
          const bool ok = lu6(Hm, piv);
          double out[6] = {0, 0, 0, 0, 0, 0};
          // inv(Hm)·u column by column (the reference forms the inverse, then multiplies)
          for (int c = 0; c < 6; ++c) {
            double col[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = (r == c) ? 1.0 : 0.0;
            lu6_solve(Hm, piv, col);
#pragma unroll
            for (int r = 0; r < 6; ++r) out[r] += col[r] * u[c];
          }
#pragma unroll
          for (int r = 0; r < 6; ++r) lphi[pi * 6 + r] = ok ? a.lr * out[r] : __builtin_nan("");
        }
      }
    }
  } else {
    if (tid == 0)
      for (int d = 0; d < 6; ++d) lphi[d] = -lN[d];           // SVNICP.cpp:89
  }
  __syncthreads();

  stamp(2);
  // ---- 5. traces (tests only) ----
  if (a.trH) {
    for (int e = tid; e < P * 36; e += UT) a.trH[e] = w.H[e];
    for (int e = tid; e < P * 6; e += UT) { a.trb[e] = lb[e]; a.trN[e] = lN[e]; a.trphi[e] = lphi[e]; }
    if (tid == 0) *a.trh = sel.h;
  }

  // ---- 6. pose update (SVNICP.cpp:268-279) + early stop statistic ----
  double my_norm = 0.0;
  for (int p = tid; p < P; p += UT) {
    double phi[6], dR[9], Jl[9], dt[3], Rn[9], Rdt[3], Ro[9];
#pragma unroll
    for (int d = 0; d < 6; ++d) phi[d] = lphi[p * 6 + d];
    so3_exp(phi + 3, dR, Jl);
    mat3_vec(Jl, phi, dt);
#pragma unroll
    for (int i = 0; i < 9; ++i) Ro[i] = a.R[9 * p + i];
    mat3_mul(Ro, dR, Rn);
    mat3_vec(Rn, dt, Rdt);                                    // uses the UPDATED R (:277-278)
    double tn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tn[i] = Rdt[i] + a.t[3 * p + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) a.R[9 * p + i] = Rn[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.t[3 * p + i] = tn[i];
    // next iteration's total pose (SVNICP.cpp:58-59)
    double Rt[9], tt[3];
    mat3_mul(a.pose.R0, Rn, Rt);
    mat3_vec(a.pose.R0, tn, tt);
#pragma unroll
    for (int i = 0; i < 9; ++i) a.Rtot[12 * p + i] = Rt[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.Rtot[12 * p + 9 + i] = a.pose.t0[i] + tt[i];
    double n2 = 0.0;
#pragma unroll
    for (int d = 0; d < 6; ++d) n2 += phi[d] * phi[d];
    my_norm += sqrt(n2);
    // pose_particles_ = [t ; Log R] (SVNICP.cpp:103-106)
    double lg[3];
    so3_log(Rn, lg);
#pragma unroll
    for (int i = 0; i < 3; ++i) { a.pose_out[i * P + p] = tn[i]; a.pose_out[(3 + i) * P + p] = lg[i]; }
  }
  stamp(3);
  bool stop = false;
  if (a.check_early_stop) {  // block-uniform
    for (int off = 32; off > 0; off >>= 1) my_norm += __shfl_xor(my_norm, off, kWave);
    if (lane == 0) sh_norm[wave] = my_norm;
    __syncthreads();
    double m = 0.0;
    for (int i = 0; i < UT / kWave; ++i) m += sh_norm[i];
    m /= P;
    // torch::lt(f64 0-dim, f32 1-dim) promotes to float32 (SVNICP.cpp:42,96-97)
    stop = (float)m < (float)a.conv_thr;
  }
  if (stop) {
    if (tid == 0) { a.ctl[0] = 1; a.ctl[1] = a.iteration + 1; }
    return;  // history row of the stopping epoch stays zero (break before :103-107)
  }
  __syncthreads();
  for (int e = tid; e < 6 * P; e += UT) a.history[(size_t)a.iteration * 6 * P + e] = (float)a.pose_out[e];
  stamp(4);
}

// ---------------------------------------------------------------------------------------------
// Large particle sets (P > 256: several GPUs' shards, or C4 on one GPU): the same per-iteration
// Stein step as k_particle_update, cut into workgroup-parallel kernels so that the O(P²) pair work
// runs on the whole chip instead of one CU.  Same arithmetic per particle.  The exact lower median of
// the P² pair distances comes from two parallel passes over the pairs: (1) a histogram of the f64 keys
// in logarithmic bins (48 octaves from 2^-40, 256 mantissa steps each; everything outside lands in the
// edge bins) locates the bin holding the median and the rank inside it; (2) the keys of that one bin
// (~0.4 % of the pairs) are collected and an exact radix select runs on them.  Deterministic, no
// sampling, exact for any input (a degenerate distribution only makes the last select longer).
// uctl doubles: [2] h  [3..38] Hinv ; as u64: [40] nan flag [42] median bin [43] rank inside
// the bin [44] collected count ; [64 .. 64+P) step norms ; then the global histogram (u32 x HB_NB).
// ---------------------------------------------------------------------------------------------
constexpr int UCTL_H = 2, UCTL_HINV = 3, UCTL_NAN = 40, UCTL_BIN = 42, UCTL_RANK = 43, UCTL_CNT = 44,
              UCTL_NORM = 64;
constexpr int HB_OCT = 48, HB_NB = HB_OCT * 256, HB_EXP0 = 1023 - 40;
constexpr int SEL_LDS_KEYS = 16384;
constexpr int COLL_CHUNK = 4096;   // pairs per collect chunk = capacity of its LDS staging buffer

__device__ __forceinline__ int key_bin(unsigned long long k) {
  const long long kb = (long long)(k >> 44) - ((long long)HB_EXP0 << 8);
  return kb < 0 ? 0 : (kb >= HB_NB ? HB_NB - 1 : (int)kb);
}
__device__ __forceinline__ unsigned int* upd_hist(double* uctl, int P) {
  return reinterpret_cast<unsigned int*>(uctl + UCTL_NORM + ((P + 7) & ~7));
}
__device__ __forceinline__ double* upd_hpart(double* uctl, int P) {  // [ceil(P/128)][36] partial Hessian sums
  return uctl + UCTL_NORM + ((P + 7) & ~7) + HB_NB / 2;
}

// SVGD-ICP pieces shared by the one-workgroup kernel and this chain (defined with k_particle_update_svgd below)
__device__ void svgd_gradient(const UpdateArgs& a, int p, double* g6);
__device__ double svgd_step_one(const UpdateArgs& a, int p, const double* phi6, const double* xold6);

// ---- the sums-dependent half of the Stein step: k_upd_prepare ----------------------------------------------------------
// Per particle: H (+1e-6·I), b and the Newton step N = H⁻¹b (SVNICP.cpp:146-162) — in SVGD-ICP mode the first-order
// gradient goes into the N slot (SVGDICP.cpp:398-455); for the default SVN branch also the mean Hessian (summed in particle
// order) and its inverse (SVNICP.cpp:85,225).  It needs the sums and nothing else; the other half of the step (the pair
// statistics: k_upd_median or the k_upd_hist chain) needs the poses and nothing else and runs on a second stream beside the
// search and accumulate kernels.  Workgroups 0 … ceil(P/64)−1: one particle per lane of wave 0 (a 6x6 LU per lane is a long
// serial chain: 64 per workgroup spreads it over the chip); the last workgroup: the mean Hessian from ITS OWN finalisation
// of every particle (no workgroup waits for another) and the inverse.
// Measured and dropped in round 3: running this as the tail of k_reduce_partials (its last workgroup, one ticket per
// workgroup) — as one workgroup for all particles 20 us, with one reduce workgroup per particle + a ticketed mean 28 us,
// against 6 + 8 us for the two launches: a serial tail on one CU costs more than the launch it saves.
constexpr int PREP_T = 256, PREP_CH = 128, PREP_PW = 64;
struct PrepShared { double H[PREP_CH][37]; double Hmean[36]; };   // 37.3 KB

// bx: workgroup index inside the prepare part of the launch (the block may have more than PREP_T threads: the others only
// pass the barriers)
__device__ __forceinline__ void prepare_body(const UpdateArgs& a, int bx) {
  if (a.ctl[0]) return;
  __shared__ PrepShared sh;
  const int tid = threadIdx.x, P = a.P;
  Work w(a.work, P);
  const int n_pw = (P + PREP_PW - 1) / PREP_PW;
  if (bx < n_pw) {
    const int p = bx * PREP_PW + tid;
    if (tid >= PREP_PW || p >= P) return;
    if (a.svgd) {
      double g6[6];
      svgd_gradient(a, p, g6);
#pragma unroll
      for (int d = 0; d < 6; ++d) w.N[p * 6 + d] = g6[d];
      return;
    }
    double Rc[9], H[36], b[6], LU[36], x6[6], sm[kNSums];
    int piv[6];
    mat3_mul(a.pose.R0, a.R + 9 * p, Rc);
    load_sums(a, p, sm);
    if (a.sums_out) {   // small chain: the reduced record, where k_reduce_partials would have left it (svnicp_sums_devptr)
#pragma unroll
      for (int i = 0; i < kNSums; ++i) a.sums_out[(size_t)p * kNSums + i] = sm[i];
    }
    finalize_Hb(sm, Rc, H, b);
#pragma unroll
    for (int i = 0; i < 36; ++i) { w.H[(size_t)p * 36 + i] = H[i]; LU[i] = H[i]; }
    const bool ok = lu6(LU, piv);
#pragma unroll
    for (int i = 0; i < 6; ++i) x6[i] = b[i];
    lu6_solve(LU, piv, x6);                                   // SVNICP.cpp:162
#pragma unroll
    for (int i = 0; i < 6; ++i) { w.b[p * 6 + i] = b[i]; w.N[p * 6 + i] = ok ? x6[i] : __builtin_nan(""); }
    return;
  }
  // last workgroup (launched only for the default SVN branch): mean Hessian and its inverse
  double hsum = 0.0;                   // thread e < 36: Σ_p H_p[e], particle order
  for (int c0 = 0; c0 < P; c0 += PREP_CH) {
    const int p = c0 + tid;
    if (tid < PREP_CH && p < P) {
      double Rc[9], H[36], b[6], sm[kNSums];
      mat3_mul(a.pose.R0, a.R + 9 * p, Rc);
      load_sums(a, p, sm);
      finalize_Hb(sm, Rc, H, b);
#pragma unroll
      for (int i = 0; i < 36; ++i) sh.H[tid][i] = H[i];
    }
    __syncthreads();
    const int cnt = P - c0 < PREP_CH ? P - c0 : PREP_CH;
    if (tid < 36)
      for (int q = 0; q < cnt; ++q) hsum += sh.H[q][tid];
    __syncthreads();
  }
  if (tid < 36) sh.Hmean[tid] = hsum / P;                     // mean over particles (SVNICP.cpp:85)
  __syncthreads();
  if (tid < 6) {                                              // linalg::inv (SVNICP.cpp:225): column tid of the inverse
    double LU[36], col[6];
    int piv[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) LU[i] = sh.Hmean[i];
    const bool ok = lu6(LU, piv);
#pragma unroll
    for (int r = 0; r < 6; ++r) col[r] = (r == tid) ? 1.0 : 0.0;
    lu6_solve(LU, piv, col);
#pragma unroll
    for (int r = 0; r < 6; ++r) a.uctl[UCTL_HINV + 6 * r + tid] = ok ? col[r] : __builtin_nan("");
  }
}
__global__ __launch_bounds__(PREP_T) void k_upd_prepare(UpdateArgs a) { prepare_body(a, (int)blockIdx.x); }

// sums[p_lo + i][s] = Σ_blk partial[blk][i][s], block order fixed.  Workgroup = 16 entries × 16 block lanes; each block
// lane walks blk = bl, bl+16, … with eight loads in flight and the 16 lanes are folded in order: deterministic, and
// independent of the launch geometry.
__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partial, int nblk, int Ppad, int p_lo,
                                                          int n_particles, double* __restrict__ sums, const int* __restrict__ ctl) {
  if (ctl[0]) return;
  __shared__ double red[16][17];
  const int el = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int entry = blockIdx.x * 16 + el;  // index into [n_particles][kNSums]
  const int n_entries = n_particles * kNSums;
  double a = 0.0;
  if (entry < n_entries) {
    const size_t stride = (size_t)Ppad * kNSums;
    const double* src = partial + entry;
    int blk = bl;
    for (; blk + 7 * 16 < nblk; blk += 8 * 16) {
      double v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(blk + 16 * i) * stride];
#pragma unroll
      for (int i = 0; i < 8; ++i) a += v[i];
    }
    for (; blk < nblk; blk += 16) a += src[(size_t)blk * stride];
  }
  red[bl][el] = a;
  __syncthreads();
  if (bl == 0 && entry < n_entries) {
    double s = red[0][el];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += red[i][el];
    sums[(size_t)p_lo * kNSums + entry] = s;
  }
}

// pass 1 over all pairs: log-binned histogram (LDS per workgroup, merged with global atomics); the bin of the
// lower median is found at the start of k_upd_collect (a last-workgroup scan here cost 30 us of serial tail)
__global__ __launch_bounds__(256) void k_upd_hist(UpdateArgs a) {
  if (a.ctl[0]) return;
  extern __shared__ __align__(16) double dyn[];
  const int tid = threadIdx.x;
  const int P = a.P;
  Work w(a.work, P);
  double* lx = dyn;
  unsigned int* lh = reinterpret_cast<unsigned int*>(dyn + 6 * P);
  // x = pose_particles_ = [t ; Log R] as the last pose update left it (SVNICP.cpp:74-77,103-106; SVGD-ICP: as it stands,
  // SVGDICP.cpp:106-110) — this chain runs beside the stage-B kernels and must not depend on anything they produce
  for (int e = tid; e < 6 * P; e += 256) {
    const int pp = e / 6, d = e - 6 * pp;
    const double v = a.pose_out[d * P + pp];
    lx[e] = v;
    if (blockIdx.x == 0) w.x[e] = v;   // k_upd_direction reads x from here
  }
  for (int e = tid; e < HB_NB; e += 256) lh[e] = 0u;
  __syncthreads();
  unsigned long long* u = reinterpret_cast<unsigned long long*>(a.uctl);
  unsigned int* gh = upd_hist(a.uctl, P);
  const int n = P * P;
  bool nan = false;
  {
    const int stride = gridDim.x * 256;              // pair index advance per step: (di, dj) without a division per pair
    const int di = stride / P, dj = stride - di * P;
    int e = blockIdx.x * 256 + tid;
    int i = e / P, j = e - i * P;
    for (; e < n; e += stride) {
      const double s = pair_sq(lx, i, j);
      if (s != s) nan = true;
      atomicAdd(&lh[key_bin((unsigned long long)__double_as_longlong(s))], 1u);
      j += dj; i += di;
      if (j >= P) { j -= P; ++i; }
    }
  }
  if (nan) u[UCTL_NAN] = 1ull;
  __syncthreads();
  for (int e = tid; e < HB_NB; e += 256) {
    const unsigned int c = lh[e];
    if (c) atomicAdd(&gh[e], c);
  }
}

// pass 2 over all pairs: the keys of the median's bin go to work.sq (LDS staging, one global atomic per workgroup)
__global__ __launch_bounds__(256) void k_upd_collect(UpdateArgs a) {
  if (a.ctl[0]) return;
  extern __shared__ __align__(16) double dyn[];
  __shared__ unsigned int sh_cnt, sh_wsum[4];
  __shared__ int sh_bin;
  __shared__ unsigned long long sh_base;
  const int tid = threadIdx.x;
  const int P = a.P;
  Work w(a.work, P);
  double* lx = dyn;
  double* lbuf = dyn + 6 * P;  // [COLL_CHUNK]: matches of one chunk of pairs
  for (int e = tid; e < 6 * P; e += 256) { const int pp = e / 6, d = e - 6 * pp; lx[e] = a.pose_out[d * P + pp]; }
  if (tid == 0) sh_cnt = 0u;
  unsigned long long* u = reinterpret_cast<unsigned long long*>(a.uctl);
  const int n = P * P;
  {  // bin of the lower median: every workgroup scans the finished global histogram itself (48 plain loads per
     // thread, L2 resident); workgroup 0 publishes bin and rank for k_upd_select
    const unsigned int* gh = upd_hist(a.uctl, P);
    constexpr int CH = HB_NB / 256;  // bins per thread, contiguous
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    unsigned int c[CH], tot = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) { c[i] = gh[tid * CH + i]; tot += c[i]; }
    unsigned int incl = tot;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned int v = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += v;
    }
    if (lane == kWave - 1) sh_wsum[wave] = incl;
    __syncthreads();
    unsigned int cum = incl - tot;
    for (int wv = 0; wv < wave; ++wv) cum += sh_wsum[wv];
    const unsigned int rank = (unsigned int)((n - 1) / 2);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (rank >= cum && rank < cum + c[i]) {
        sh_bin = tid * CH + i;
        if (blockIdx.x == 0) { u[UCTL_BIN] = (unsigned long long)(tid * CH + i); u[UCTL_RANK] = rank - cum; }
      }
      cum += c[i];
    }
  }
  __syncthreads();
  const int bstar = sh_bin;
  // the workgroup's pairs in chunks of COLL_CHUNK: a chunk cannot overflow the LDS buffer, and each chunk with
  // matches costs one global atomic
  for (int c0 = blockIdx.x * COLL_CHUNK; c0 < n; c0 += gridDim.x * COLL_CHUNK) {
    const int c1 = (c0 + COLL_CHUNK < n) ? c0 + COLL_CHUNK : n;
    {
      const int di = 256 / P, dj = 256 - di * P;
      int e = c0 + tid;
      int i = e / P, j = e - i * P;
      for (; e < c1; e += 256) {
        const double s = pair_sq(lx, i, j);
        if (key_bin((unsigned long long)__double_as_longlong(s)) == bstar) lbuf[atomicAdd(&sh_cnt, 1u)] = s;
        j += dj; i += di;
        if (j >= P) { j -= P; ++i; }
      }
    }
    __syncthreads();
    const unsigned int cnt = sh_cnt;
    if (cnt) {  // block-uniform
      if (tid == 0) sh_base = atomicAdd(&u[UCTL_CNT], (unsigned long long)cnt);
      __syncthreads();
      const unsigned long long base = sh_base;
      for (unsigned int e = tid; e < cnt; e += 256) w.sq[base + e] = lbuf[e];
      __syncthreads();
      if (tid == 0) sh_cnt = 0u;
      __syncthreads();
    }
  }
}

// generic block-wide exact rank selection over n non-negative f64 keys given by key_at(e); passes above
// first_pass are skipped with their digits taken from prefix0 (keys known to share those bits)
template <class F>
__device__ unsigned long long block_select(F key_at, int n, unsigned int rank, int first_pass, unsigned long long prefix0,
                                           SelShared* S, int tid, int lane, int wave) {
  if (tid < 256) S->hist[tid] = 0;
  if (tid == 0) { S->prefix = prefix0; S->rank = rank; }
  __syncthreads();
  for (int pass = first_pass; pass >= 0; --pass) {
    const int shift = pass * 8;
    const unsigned long long pre = S->prefix;
    for (int e = tid; e < n; e += UT) {
      const unsigned long long k = key_at(e);
      if (pass == 7 || (k >> (shift + 8)) == pre) atomicAdd(&S->hist[(k >> shift) & 255ull], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      unsigned int c[4], tot = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { c[i] = S->hist[4 * lane + i]; tot += c[i]; S->hist[4 * lane + i] = 0; }
      unsigned int incl = tot;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned int v = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += v;
      }
      unsigned int cum = incl - tot;
      const unsigned int r = S->rank;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (r >= cum && r < cum + c[i]) { S->prefix = (pre << 8) | (unsigned long long)(4 * lane + i); S->rank = r - cum; }
        cum += c[i];
      }
    }
    __syncthreads();
  }
  return S->prefix;
}

// exact median inside its bin -> h
__global__ __launch_bounds__(UT) void k_upd_select(UpdateArgs a) {
  if (a.ctl[0]) return;
  extern __shared__ __align__(16) double dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = a.P;
  Work w(a.work, P);
  __shared__ SelShared sel;
  const unsigned long long* u = reinterpret_cast<const unsigned long long*>(a.uctl);
  const int m = (int)u[UCTL_CNT];
  const unsigned int r = (unsigned int)u[UCTL_RANK];
  const int bstar = (int)u[UCTL_BIN];
  // interior bins share the top 20 key bits: passes 7 and 6 are known
  const bool interior = bstar > 0 && bstar < HB_NB - 1;
  const unsigned long long top20 = (unsigned long long)bstar + ((unsigned long long)HB_EXP0 << 8);
  const int first_pass = interior ? 5 : 7;
  const unsigned long long prefix0 = interior ? (top20 >> 4) : 0ull;
  unsigned long long kmed;
  if (m <= SEL_LDS_KEYS) {
    for (int e = tid; e < m; e += UT) dyn[e] = w.sq[e];
    __syncthreads();
    auto key_at = [&](int e) -> unsigned long long { return (unsigned long long)__double_as_longlong(dyn[e]); };
    kmed = block_select(key_at, m, r, first_pass, prefix0, &sel, tid, lane, wave);
  } else {  // degenerate distribution (most pairs in one bin): same select on the global buffer
    auto key_at = [&](int e) -> unsigned long long { return (unsigned long long)__double_as_longlong(w.sq[e]); };
    kmed = block_select(key_at, m, r, first_pass, prefix0, &sel, tid, lane, wave);
  }
  if (tid == 0) {
    const double med = u[UCTL_NAN] ? __builtin_nan("") : __longlong_as_double((long long)kmed);
    a.uctl[UCTL_H] = med / log((double)(P + 1));              // SVNICP.cpp:262
  }
  // leave the chain's global state as the next iteration's k_upd_hist expects it (svnicp_align_begin zeroes it once)
  __syncthreads();
  unsigned int* gh = upd_hist(a.uctl, P);
  for (int e = tid; e < HB_NB; e += UT) gh[e] = 0u;
  if (tid == 0) {
    unsigned long long* uw = reinterpret_cast<unsigned long long*>(a.uctl);
    uw[UCTL_NAN] = 0ull; uw[UCTL_CNT] = 0ull;
  }
}

// The pair statistics of the Stein step for 2 <= P <= 128, one workgroup: the exact lower median of the P² pair distances
// (torch::median over all entries incl. the diagonal's zeros, SVNICP.cpp:262) through an LDS copy of the log-binned
// histogram of the k_upd_* chain — bin the keys, find the median's bin, collect that bin (~0.4 % of the keys), rank its keys
// by counting — and with it the bandwidth h.  Needs the poses only (x = pose_particles_ = [t ; Log R], which the last pose
// update left in pose_out), so it is launched on the context's second stream at the START of an iteration and runs beside
// the search and accumulate kernels; k_upd_direction (one wavefront per particle, pose update fused) waits for it.
// Measured (debug stamps): the 8-pass LDS radix select took 60 % of the fused kernel's 63 us; this kernel takes 13.5 us.
constexpr int FRONT_BUF = 2048;  // keys of the median's bin held in LDS (+8 slack for the unrolled ranking); more (degenerate input) -> 8-pass select
template <int T>   // T: threads of the workgroup (T for the kernels of this file, 256 inside the persistent small-registration kernel)
__device__ __forceinline__ void median_body(const UpdateArgs& a) {
  if (a.ctl[0]) return;
  extern __shared__ __align__(16) double dyn[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int P = a.P;
  Work w(a.work, P);
  double* lx = dyn;                                                   // [P][6]
  double* lbuf = dyn + 6 * P;                                         // [FRONT_BUF]
  unsigned int* lh = reinterpret_cast<unsigned int*>(lbuf + FRONT_BUF + 8);  // [HB_NB]
  __shared__ SelShared sel;
  __shared__ unsigned int sh_scan[T];
  __shared__ unsigned int sh_cnt;
  __shared__ int sh_bin, sh_rank, sh_nan;

  unsigned long long tdbg = a.dbg ? __builtin_readcyclecounter() : 0ull;
  auto stamp = [&](int i) {   // debug option: thread-0 cycles per phase of the median workgroup
    if (!a.dbg || tid != 0) return;
    const unsigned long long now = __builtin_readcyclecounter();
    a.dbg[i] += now - tdbg;
    tdbg = now;
  };
  for (int e = tid; e < HB_NB; e += T) lh[e] = 0u;
  if (tid == 0) { sh_cnt = 0u; sh_nan = 0; sh_bin = 0; sh_rank = 0; }
  for (int p = tid; p < P; p += T) {   // x = pose_particles_ (SVNICP.cpp:74-77,103-106; SVGD-ICP: as it stands, SVGDICP.cpp:106-110)
#pragma unroll
    for (int d = 0; d < 6; ++d) { const double v = a.pose_out[d * P + p]; lx[p * 6 + d] = v; w.x[p * 6 + d] = v; }
  }
  __syncthreads();
  stamp(0);

  // pass 1 over the pairs: log-binned histogram.  The matrix of pair distances is symmetric bit for bit ((a-b)² == (b-a)²)
  // with zeros on the diagonal: only the pairs i < j are binned, each with weight 2, and the P diagonal zeros go in with
  // one update — half the distance evaluations and half the LDS atomics (which pile up on a few bins: 46 % of this
  // workgroup's time went into this pass)
  const int n = P * P;
  bool nan = false;
  // the pairs i < j as a rectangle of Pe/2 rows x (Pe - 1) columns (Pe = P rounded up to even): row a holds (a, c + 1) for
  // c >= a and (Pe - 1 - a, Pe - 1 - c) for c < a — every unordered pair exactly once, so all lanes work in every step
  constexpr int KH = ((KREG + 1) / 2 + 1) * (512 / T);   // steps per thread: KH * T >= (Pe / 2)(Pe - 1) for P <= 128
  const int Pe = P + (P & 1), W = Pe - 1, npair = (Pe / 2) * W;
  const int di = T / W, dj = T - di * W;   // pair index advance per step of T entries
  double keys[KH];                           // this thread's pair distances with i < j
  const int bin0 = key_bin(0ull);            // bin of +0.0
  if (tid == 0) atomicAdd(&lh[bin0], (unsigned int)P);
  for (int p = tid; p < P; p += T) { const double sq = pair_sq(lx, p, p); if (sq != sq) nan = true; }   // a non-finite particle: inf - inf on the diagonal
  {
    int ra = tid / W, c = tid - ra * W;
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int e = tid + k * T;
      keys[k] = __builtin_huge_val();
      const int i = c >= ra ? ra : Pe - 1 - ra, j = c >= ra ? c + 1 : Pe - 1 - c;
      if (e < npair && j < P) {              // (j < P also implies i < P; only an odd P has a virtual last index)
        const double sq = pair_sq(lx, i, j);
        if (sq != sq) nan = true;
        keys[k] = sq;
        atomicAdd(&lh[key_bin((unsigned long long)__double_as_longlong(sq))], 2u);
      }
      c += dj; ra += di;
      if (c >= W) { c -= W; ++ra; }
    }
  }
  if (nan) sh_nan = 1;
  __syncthreads();
  stamp(1);
  {  // bin of the lower median: contiguous chunk of bins per thread, block-wide exclusive scan of the chunk sums
    constexpr int CH = HB_NB / T;
    unsigned int c[CH], tot = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) { c[i] = lh[tid * CH + i]; tot += c[i]; }
    unsigned int incl = tot;  // inclusive scan inside the wavefront, then the eight wave totals
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned int v = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += v;
    }
    if (lane == kWave - 1) sh_scan[wave] = incl;
    __syncthreads();
    unsigned int wbase = 0;
    for (int wv = 0; wv < wave; ++wv) wbase += sh_scan[wv];
    unsigned int cum = wbase + incl - tot;
    const unsigned int rank = (unsigned int)((n - 1) / 2);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (rank >= cum && rank < cum + c[i]) { sh_bin = tid * CH + i; sh_rank = (int)(rank - cum); }
      cum += c[i];
    }
  }
  __syncthreads();
  stamp(2);
  // pass 2: the keys of that bin (one copy of each i < j pair)
  const int bstar = sh_bin;
#pragma unroll
  for (int k = 0; k < KH; ++k) {
    if (keys[k] < __builtin_huge_val() && key_bin((unsigned long long)__double_as_longlong(keys[k])) == bstar) {
      const unsigned int pos = atomicAdd(&sh_cnt, 1u);
      if (pos < FRONT_BUF) lbuf[pos] = keys[k];
    }
  }
  __syncthreads();
  stamp(3);
  const int m = (int)sh_cnt;
  double med;
  if (m <= FRONT_BUF) {
    // exact rank inside the bin by counting, every collected key standing for two matrix entries and the diagonal for P
    // zeros: the value with #less <= r < #less + #equal is the median
    const int r = sh_rank;
    const bool zin = bstar == bin0;          // the diagonal's zeros are in this bin
    for (int e = m + tid; e < ((m + 7) & ~7); e += T) lbuf[e] = __builtin_huge_val();  // pad to the unroll width
    __syncthreads();
    for (int e = tid; e < m; e += T) {
      const double v = lbuf[e];
      int lt = 0, eq = 0;
      for (int j0 = 0; j0 < m; j0 += 8) {  // eight broadcast reads in flight
        double u[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) u[t] = lbuf[j0 + t];
#pragma unroll
        for (int t = 0; t < 8; ++t) { lt += u[t] < v ? 1 : 0; eq += u[t] == v ? 1 : 0; }
      }
      const int LT = 2 * lt + ((zin && 0.0 < v) ? P : 0), EQ = 2 * eq + ((zin && v == 0.0) ? P : 0);
      if (LT <= r && r < LT + EQ) sel.h = v;  // every matching thread writes the same value
    }
    if (zin && tid == 0) {                    // the median may be one of the diagonal's zeros
      int eq0 = 0;
      for (int j = 0; j < m; ++j) eq0 += lbuf[j] == 0.0 ? 1 : 0;
      if (r < 2 * eq0 + P) sel.h = 0.0;
    }
    __syncthreads();
    med = sel.h;
  } else {  // degenerate distribution (most pairs in one bin): the general 8-pass select
    sel_init(&sel, P, tid);
    __syncthreads();
    rbf_bandwidth<T>(lx, P, w.sq, &sel, tid, lane, wave);
    med = sel.h * log((double)(P + 1));  // rbf_bandwidth returns h, undo its scaling
    __syncthreads();
  }
  stamp(4);
  if (tid == 0) a.uctl[UCTL_H] = (sh_nan ? __builtin_nan("") : med) / log((double)(P + 1));  // SVNICP.cpp:262
}
__global__ __launch_bounds__(UT) void k_upd_median(UpdateArgs a) { median_body<UT>(a); }

// Small registrations: both halves of the Stein step's front in ONE launch on the main stream — the last workgroup runs the
// pair statistics, the others the sums-dependent half.  At the scan-to-map loop's sizes every kernel of an iteration runs
// at its launch latency, and the second stream's fork and join cost 6-8 us each: k_reduce_partials (the prepare lanes add
// the workgroups' records themselves, load_sums), k_upd_median's own launch and both event waits fall away.
__global__ __launch_bounds__(UT) void k_upd_prepare_median(UpdateArgs a) {
  if (blockIdx.x + 1 == gridDim.x) median_body<UT>(a);
  else prepare_body(a, (int)blockIdx.x);
}

// pose update of one particle (SVNICP.cpp:268-279); its step norm goes to uctl[UCTL_NORM + p]
__device__ void upd_pose_one(const UpdateArgs& a, int p, const double* phi) {
  const int P = a.P;
  double dR[9], Jl[9], dt[3], Rn[9], Rdt[3], Ro[9];
  so3_exp(phi + 3, dR, Jl);
  mat3_vec(Jl, phi, dt);
#pragma unroll
  for (int i = 0; i < 9; ++i) Ro[i] = a.R[9 * p + i];
  mat3_mul(Ro, dR, Rn);
  mat3_vec(Rn, dt, Rdt);
  double tn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) tn[i] = Rdt[i] + a.t[3 * p + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) a.R[9 * p + i] = Rn[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) a.t[3 * p + i] = tn[i];
  double Rt[9], tt[3];
  mat3_mul(a.pose.R0, Rn, Rt);
  mat3_vec(a.pose.R0, tn, tt);
#pragma unroll
  for (int i = 0; i < 9; ++i) a.Rtot[12 * p + i] = Rt[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) a.Rtot[12 * p + 9 + i] = a.pose.t0[i] + tt[i];
  double n2 = 0.0;
#pragma unroll
  for (int d = 0; d < 6; ++d) n2 += phi[d] * phi[d];
  a.uctl[UCTL_NORM + p] = sqrt(n2);
  double lg[3];
  so3_log(Rn, lg);
#pragma unroll
  for (int i = 0; i < 3; ++i) { a.pose_out[i * P + p] = tn[i]; a.pose_out[(3 + i) * P + p] = lg[i]; }
  if (!a.check_early_stop) {  // no stop decision pending: the history row (SVNICP.cpp:103-107) can go out now
    float* hrow = a.history + (size_t)a.iteration * 6 * P;
#pragma unroll
    for (int i = 0; i < 3; ++i) { hrow[i * P + p] = (float)tn[i]; hrow[(3 + i) * P + p] = (float)lg[i]; }
  }
}

// Stein direction (SVNICP.cpp:218-252), one wavefront per particle, then that particle's pose update.
// x and the Newton steps of all particles are read from the prepare kernel's arrays (L2 resident); R/t
// of particle pi are only touched by its own wavefront.
__device__ __forceinline__ void direction_body(const UpdateArgs& a, int bx) {
  if (a.ctl[0]) return;
  constexpr int TPP = kWave;
  const int tid = threadIdx.x;
  const int P = a.P;
  Work w(a.work, P);
  const int pi = bx * (256 / TPP) + tid / TPP, part = tid % TPP;
  if (pi >= P) return;  // whole wavefront
  const double h = a.uctl[UCTL_H];
  double xi[6], phi[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) xi[d] = w.x[pi * 6 + d];
  if (a.svgd) {  // svgd_grad (SVGDICP.cpp:457-474) + optimizer step + pose refresh of this particle (:476-494, :118-121)
    double gr[6] = {0, 0, 0, 0, 0, 0}, kg[6] = {0, 0, 0, 0, 0, 0};
    for (int j = part; j < P; j += TPP) {
      double df[6], sq = 0.0;
#pragma unroll
      for (int d = 0; d < 6; ++d) { df[d] = xi[d] - w.x[j * 6 + d]; sq += df[d] * df[d]; }
      const double k = exp(-sq / h);
#pragma unroll
      for (int d = 0; d < 6; ++d) { gr[d] += df[d] * k; kg[d] += k * (-w.N[j * 6 + d]); }
    }
    for (int off = TPP >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int d = 0; d < 6; ++d) { gr[d] += __shfl_xor(gr[d], off, kWave); kg[d] += __shfl_xor(kg[d], off, kWave); }
    }
    if (part == 0) {
#pragma unroll
      for (int d = 0; d < 6; ++d) { phi[d] = (kg[d] + 2 / h * gr[d]) / P; w.phi[pi * 6 + d] = phi[d]; }
      a.uctl[UCTL_NORM + pi] = svgd_step_one(a, pi, phi, xi);
      if (!a.check_early_stop) {  // no stop decision pending: the history row (SVGDICP.cpp:133) can go out now
        float* hrow = a.history + (size_t)a.iteration * 6 * P;
#pragma unroll
        for (int d = 0; d < 6; ++d) hrow[d * P + pi] = (float)a.pose_out[d * P + pi];
      }
    }
    return;
  }
  if (!a.full_grad) {
    double g[6] = {0, 0, 0, 0, 0, 0}, kn[6] = {0, 0, 0, 0, 0, 0}, ks = 0.0;
    for (int j = part; j < P; j += TPP) {
      double df[6], sq = 0.0;
#pragma unroll
      for (int d = 0; d < 6; ++d) { df[d] = xi[d] - w.x[j * 6 + d]; sq += df[d] * df[d]; }
      const double k = exp(-sq / h);
#pragma unroll
      for (int d = 0; d < 6; ++d) { g[d] += df[d] * k; kn[d] += k * (-w.N[j * 6 + d]); }
      ks += k;
    }
    for (int off = TPP >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int d = 0; d < 6; ++d) { g[d] += __shfl_xor(g[d], off, kWave); kn[d] += __shfl_xor(kn[d], off, kWave); }
      ks += __shfl_xor(ks, off, kWave);
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) g[d] = 2 / h * g[d];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double hg = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) hg += a.uctl[UCTL_HINV + 6 * r + c] * g[c];
      phi[r] = (kn[r] + hg) / ks;
    }
  } else {
    double Hm[36], uu[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 36; ++e) Hm[e] = 0.0;
    for (int j = part; j < P; j += TPP) {
      double df[6], sq = 0.0;
#pragma unroll
      for (int d = 0; d < 6; ++d) { df[d] = xi[d] - w.x[j * 6 + d]; sq += df[d] * df[d]; }
      const double k = exp(-sq / h);
      double g[6];
#pragma unroll
      for (int d = 0; d < 6; ++d) g[d] = 2 / h * (df[d] * k);
      const double k2 = k * k;
      const double* Hj = w.H + (size_t)j * 36;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) Hm[6 * r + c] += k2 * Hj[6 * r + c] + g[r] * g[c];
        uu[r] += k * (-w.b[j * 6 + r]) + g[r];
      }
    }
    for (int off = TPP >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int e = 0; e < 36; ++e) Hm[e] += __shfl_xor(Hm[e], off, kWave);
#pragma unroll
      for (int r = 0; r < 6; ++r) uu[r] += __shfl_xor(uu[r], off, kWave);
    }
#pragma unroll
    for (int e = 0; e < 36; ++e) Hm[e] /= P;
#pragma unroll
    for (int r = 0; r < 6; ++r) uu[r] /= P;
    int piv[6];
    const bool ok = lu6(Hm, piv);
    double out[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < 6; ++c) {
      double col[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = (r == c) ? 1.0 : 0.0;
      lu6_solve(Hm, piv, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) out[r] += col[r] * uu[c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) phi[r] = ok ? a.lr * out[r] : __builtin_nan("");
  }
  if (part == 0) {
#pragma unroll
    for (int r = 0; r < 6; ++r) w.phi[pi * 6 + r] = phi[r];
    upd_pose_one(a, pi, phi);
  }
}

__global__ __launch_bounds__(256) void k_upd_direction(UpdateArgs a) { direction_body(a, (int)blockIdx.x); }

// early-stop decision on a fixed-order sum (SVNICP.cpp:95-101), traces, history (SVNICP.cpp:103-107)
__device__ __forceinline__ void finish_body(const UpdateArgs& a) {
  if (a.ctl[0]) return;
  const int tid = threadIdx.x;
  const int P = a.P;
  Work w(a.work, P);
  __shared__ double sh_part[256];
  __shared__ int sh_stop;
  if (a.trH) {
    if (!a.svgd) {
      for (int e = tid; e < P * 36; e += 256) a.trH[e] = w.H[e];
      for (int e = tid; e < P * 6; e += 256) a.trb[e] = w.b[e];
    }
    for (int e = tid; e < P * 6; e += 256) { a.trN[e] = w.N[e]; a.trphi[e] = w.phi[e]; }
    if (tid == 0) *a.trh = a.uctl[UCTL_H];
  }
  if (a.check_early_stop) {
    double s = 0.0;
    for (int p = tid; p < P; p += 256) s += a.uctl[UCTL_NORM + p];
    sh_part[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {  // fixed tree: every replica decides alike
      if (tid < off) sh_part[tid] += sh_part[tid + off];
      __syncthreads();
    }
    if (tid == 0) {
      const double m = sh_part[0] / P;
      const int stop = (float)m < (float)a.conv_thr;
      if (stop) { a.ctl[0] = 1; a.ctl[1] = a.iteration + 1; }
      sh_stop = stop;
    }
    __syncthreads();
    if (sh_stop) return;
  } else {
    return;  // history already written by k_upd_direction
  }
  for (int e = tid; e < 6 * P; e += 256) a.history[(size_t)a.iteration * 6 * P + e] = (float)a.pose_out[e];
}
__global__ __launch_bounds__(256) void k_upd_finish(UpdateArgs a) { finish_body(a); }

// ---------------------------------------------------------------------------------------------
// Small registrations, all iterations in ONE launch (svnicp_align of a context that qualifies for the small chain).
// At the scan-to-map loop's sizes an iteration is four dependent launches of 10-16 us each for a few microseconds of work
// (DESIGN.md §4.3).  k_small_registration keeps a few dozen workgroups resident for the whole registration (cooperative
// launch: the runtime refuses the grid unless every workgroup is resident at once) and runs the very same device bodies
// on virtual blocks, phase by phase, with a grid barrier between the phases:
//   A1 workgroups 0 … GS-1: search, each over its own slice of the source points;  workgroup GS: the pair statistics (it
//      only registers at the next barrier and works on through A2)
//   A2 workgroups 0 … GA-1: accumulate (the four-launch chain's partition: at most 32 records per particle)
//   B  per particle group: the workgroups' partial records added in block order, H, b, Newton step; mean Hessian + inverse
//   C  one wavefront per particle: Stein direction + pose update   [D  workgroup 0: early-stop decision, history, traces]
// The barrier is an arrival counter (four, used in turn) and a generation word in global memory: the last workgroup to
// arrive resets the counter and bumps the generation, the others poll it with s_sleep — and give up after a bounded number of polls (about a
// second), set the error word and leave, so that no wave can wait forever whatever happens to a sibling; the host then
// reports SVNICP_ERR_HIP instead of a result.  Every wave passes __threadfence() on both sides of a barrier (release of its
// own writes, invalidation of its L1 before it reads the others').  Same arithmetic, same block partition and the same
// order of additions as the four-launch small chain: bit-identical results (test_small_registration_persistent_kernel).
struct SmallArgs {
  int GS;                 // workgroups of the search phase; workgroup GS runs the pair statistics
  int GA;                 // workgroups of the accumulate phase (= records per particle in `partial`)
  int spts_per_block;     // source points per search workgroup
  int iterations;
  unsigned int* bar;      // [0] generation, [1] error word, [4 + k] arrivals of barrier k mod 4 (all zero at launch)
};

// arrive at barrier `k` (the k-th of this launch); wait = false: arrival only (the caller has nothing the others need before
// the NEXT barrier and goes on working — it still releases the barrier if it happens to be the last to arrive)
__device__ __forceinline__ bool grid_barrier(const SmallArgs& s, unsigned int& k, int nblocks, bool wait = true) {
  __shared__ int sh_ok;
  // The fences are agent-scope: on this part every XCD has its own L2, so a release writes the XCD's dirty lines back and
  // an acquire invalidates — once per WORKGROUP (thread 0, between two workgroup barriers that order the other waves'
  // accesses against it), not once per thread: 256 threads fencing on both sides cost 20 us per barrier.
  __syncthreads();
  if (threadIdx.x == 0) {
    int ok = 1;
    __threadfence();                     // release: the workgroup's writes are device-visible before the arrival
    volatile unsigned int* vb = s.bar;
    unsigned int* cnt = s.bar + 4 + (k & 3u);
    if (atomicAdd(cnt, 1u) == (unsigned int)nblocks - 1u) {
      *cnt = 0u;                         // (barrier k + 4 cannot begin before barrier k + 3 has ended, i.e. long after this)
      __threadfence();
      atomicAdd(&s.bar[0], 1u);
    } else if (wait) {
      unsigned int polls = 0u;
      while ((int)(vb[0] - (k + 1u)) < 0) {   // generation k + 1 = barrier k released
        __builtin_amdgcn_s_sleep(4);
        if (++polls > (1u << 22) || vb[1] != 0u) { atomicExch(&s.bar[1], 1u); ok = 0; break; }   // bounded: nobody waits forever
      }
    }
    __threadfence();                     // acquire: no stale line is read after the barrier
    sh_ok = ok;
  }
  __syncthreads();
  ++k;
  return sh_ok != 0;
}

template <int PW, int WP, int NRB, bool TAIL, bool SVGD>
__global__ __launch_bounds__(256) void k_small_registration(AccumArgs a, UpdateArgs u, SmallArgs s) {
  extern __shared__ __align__(16) double dyn[];
  const int bx = (int)blockIdx.x, nblocks = (int)gridDim.x;
  const int P = u.P;
  const int n_prep = (P + PREP_PW - 1) / PREP_PW + ((!u.svgd && !u.full_grad) ? 1 : 0);
  const int n_dir = (P + 3) / 4;
  const bool want_finish = u.check_early_stop != 0;   // (no traces here: a context that records traces runs the four-launch chain)
  UpdateArgs ub = u;                     // phase B reads the accumulate workgroups' records themselves
  ub.sums = a.partial; ub.n_ranks = s.GA; ub.sums_stride = a.Ppad * kNSums;
  AccumArgs as = a;                      // the search phase has its own, finer slices of the source points
  as.spts_per_block = s.spts_per_block;
  unsigned int k = 0u;
  unsigned long long tdbg = u.dbg ? __builtin_readcyclecounter() : 0ull;
  auto stamp = [&](int i) {              // option debug: workgroup 0's cycles per phase (barrier included), summed over the iterations
    if (!u.dbg || bx != 0 || threadIdx.x != 0) return;
    const unsigned long long now = __builtin_readcyclecounter();
    u.dbg[i] += now - tdbg;
    tdbg = now;
  };
  for (int it = 0; it < s.iterations; ++it) {
    if (a.ctl[0]) break;                 // early stop (uniform: read behind the last barrier's acquire)
    u.iteration = it; ub.iteration = it;
    // ---- phase A1: search on GS workgroups; workgroup GS starts the pair statistics and only REGISTERS at the barrier
    if (bx < s.GS) search_body<PW, WP, NRB, TAIL>(as, bx, 0);
    stamp(0);
    if (bx == s.GS) {
      if (!grid_barrier(s, k, nblocks, false)) return;
      median_body<256>(u);
    } else {
      if (!grid_barrier(s, k, nblocks)) return;
      stamp(1);
      // ---- phase A2: accumulate on GA workgroups
      if (bx < s.GA) accumulate_body<PW, WP, true, SVGD>(a, bx, 0, dyn);
      stamp(2);
    }
    if (!grid_barrier(s, k, nblocks)) return;
    stamp(3);
    // ---- phase B
    if (bx < n_prep) prepare_body(ub, bx);
    stamp(4);
    if (!grid_barrier(s, k, nblocks)) return;
    stamp(5);
    // ---- phase C
    if (bx < n_dir) direction_body(u, bx);
    stamp(6);
    if (!grid_barrier(s, k, nblocks)) return;
    stamp(7);
    if (want_finish) {
      if (bx == 0) finish_body(u);
      if (!grid_barrier(s, k, nblocks)) return;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// SVGD-ICP mode (first-order sibling): replaces the tail of SVGDICP::stein_align per iteration
// (src/core/SVGDICP.cpp:106-133): sgd_grad's finalisation (:398-455, Euler partials :335-396),
// svgd_grad + rbf_kernel (:457-474), pose_update through torch::optim (:476-494, options :142-170),
// the displacement early stop (:123-131) and the particle history (:133).
// Reference quirk kept: the RBF kernel is evaluated on pose_particles_ as it stood BEFORE this
// epoch's parameters were read, i.e. at epoch 0 on the previous registration's final particles.
// ---------------------------------------------------------------------------------------------
__device__ void euler_partials(const double* R0, double roll, double pitch, double yaw, double dR[3][9]) {
  const double A = cos(yaw), Bs = sin(yaw), C = cos(pitch), D = sin(pitch), E = cos(roll), F = sin(roll);
  const double DE = D * E, DF = D * F, AC = A * C, AF = A * F, AE = A * E;
  const double ADE = A * DE, ADF = A * DF, BC = Bs * C, BE = Bs * E, BF = Bs * F, BDE = Bs * DE;
  const double pr[9] = {0, ADE + BF, BE - ADF, 0, -AF + BDE, Bs * (-DF) - AE, 0, C * E, C * (-F)};
  const double pp[9] = {A * -D, AC * F, AC * E, Bs * -D, BC * F, BC * E, -C, -DF, -DE};
  const double py[9] = {-BC, -Bs * DF - AE, AF - BDE, AC, -BE + ADF, ADE + BF, 0, 0, 0};
  mat3_mul(R0, pr, dR[0]);
  mat3_mul(R0, pp, dR[1]);
  mat3_mul(R0, py, dR[2]);
}

// sgd_grad of one particle from the raw sums (SVGDICP.cpp:398-455): Euler-angle partials, (count + 1) normalisation,
// scaled by the source size
__device__ void svgd_gradient(const UpdateArgs& a, int p, double* g6) {
  double s[kNSums];
  load_sums(a, p, s);
  const double* eu = a.eul + 6 * p;
  double dR[3][9];
  euler_partials(a.pose.R0, eu[3], eu[4], eu[5], dR);
  const double cnt1 = s[4] + 1.0;  // nonzero_count + 1
  const double* R0 = a.pose.R0;
#pragma unroll
  for (int j = 0; j < 3; ++j)      // error.sum(1).matmul(R0) / (count + 1)
    g6[j] = ((s[10] * R0[j] + s[11] * R0[3 + j] + s[12] * R0[6 + j]) / cnt1) * a.n_src;
#pragma unroll
  for (int k = 0; k < 3; ++k) {    // Σ_b e·(dR_k s) = Σ_ij dR_k[i][j]·(Σ_b e_i s_j)
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) v += dR[k][3 * i + j] * s[13 + 3 * i + j];
    g6[3 + k] = (v / cnt1) * a.n_src;
  }
}

// optimizer step of one particle (param.grad = -stein_grad, SVGDICP.cpp:476-494 with torch's defaults, :142-170), next
// epoch's R_, t_, total pose (:88-91) and pose_particles_ (:118-121); returns |new pose - xold| for the early stop (:123-131)
__device__ double svgd_step_one(const UpdateArgs& a, int p, const double* phi6, const double* xold6) {
  const int P = a.P;
  const int step = a.iteration + 1;
  double n2 = 0.0, e6[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int i = p * 6 + d;
    double g = -phi6[d];
    double v = a.eul[i];
    double* m1 = a.opt + i; double* m2 = a.opt + (size_t)6 * P + i; double* m3 = a.opt + (size_t)12 * P + i;
    switch (a.optimizer) {
      case 0: {  // Adam: betas (0.9, 0.999), eps 1e-8
        const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
        const double e1 = b1 * (*m1) + (1 - b1) * g;
        const double e2 = b2 * (*m2) + (1 - b2) * g * g;
        *m1 = e1; *m2 = e2;
        const double bc1 = 1 - pow(b1, (double)step), bc2 = 1 - pow(b2, (double)step);
        v -= (a.lr / bc1) * (e1 / (sqrt(e2) / sqrt(bc2) + eps));
      } break;
      case 1: {  // RMSprop: alpha .99, eps 1e-8, weight_decay 1e-8, momentum .9
        const double alpha = 0.99, eps = 1e-8, wd = 1e-8, mom = 0.9;
        g = g + wd * v;
        const double sq = alpha * (*m1) + (1 - alpha) * g * g;
        const double buf = mom * (*m2) + g / (sqrt(sq) + eps);
        *m1 = sq; *m2 = buf;
        v -= a.lr * buf;
      } break;
      case 2: v -= a.lr * g; break;  // SGD
      default: {  // Adagrad: eps 1e-10
        const double ss = (*m3) + g * g;
        *m3 = ss;
        v -= a.lr * (g / (sqrt(ss) + 1e-10));
      } break;
    }
    a.eul[i] = v;
    e6[d] = v;
    const double df = v - xold6[d];
    n2 += df * df;
  }
  // next epoch: R_ = Euler(rx,ry,rz), t_ = (x,y,z) (SVGDICP.cpp:88-91)
  double Rm[9], Rt[9], tt[3];
  euler_to_R(e6[3], e6[4], e6[5], Rm);
  mat3_mul(a.pose.R0, Rm, Rt);
  mat3_vec(a.pose.R0, e6, tt);
#pragma unroll
  for (int i = 0; i < 9; ++i) { a.R[9 * p + i] = Rm[i]; a.Rtot[12 * p + i] = Rt[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { a.t[3 * p + i] = e6[i]; a.Rtot[12 * p + 9 + i] = a.pose.t0[i] + tt[i]; }
#pragma unroll
  for (int d = 0; d < 6; ++d) a.pose_out[d * P + p] = e6[d];   // pose_particles_ (SVGDICP.cpp:118-121)
  return sqrt(n2);
}

__global__ __launch_bounds__(UT) void k_particle_update_svgd(UpdateArgs a) {
  if (a.ctl[0]) return;
  extern __shared__ __align__(16) double dyn[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const int P = a.P;
  Work w(a.work, P);
  double* lx = dyn;             // [P][6] pose_particles_ before this epoch's step
  double* lg = lx + 6 * P;      // [P][6] sgd gradient
  double* lphi = lg + 6 * P;    // [P][6] stein gradient
  __shared__ SelShared sel;
  __shared__ double sh_norm[UT / kWave];

  // ---- 1. sgd_grad from the raw sums (SVGDICP.cpp:398-455) ----
  for (int p = tid; p < P; p += UT) {
    svgd_gradient(a, p, lg + p * 6);
#pragma unroll
    for (int d = 0; d < 6; ++d) lx[p * 6 + d] = a.pose_out[d * P + p];
  }
  sel_init(&sel, P, tid);
  __syncthreads();

  // ---- 2. svgd_grad (SVGDICP.cpp:457-474) ----
  if (P > 1) {
    rbf_bandwidth<UT>(lx, P, w.sq, &sel, tid, lane, wave);
    const double h = sel.h;
    const int tpp = threads_per_particle(P);
    const int per_pass = UT / tpp;
    for (int base = 0; base < P; base += per_pass) {
      const int pi = base + tid / tpp, part = tid % tpp;
      const bool act = pi < P;
      double xi[6], gr[6] = {0, 0, 0, 0, 0, 0}, kg[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int d = 0; d < 6; ++d) xi[d] = act ? lx[pi * 6 + d] : 0.0;
      if (act)
        for (int j = part; j < P; j += tpp) {
          double df[6], sq = 0.0;
#pragma unroll
          for (int d = 0; d < 6; ++d) { df[d] = xi[d] - lx[j * 6 + d]; sq += df[d] * df[d]; }
          const double k = exp(-sq / h);
#pragma unroll
          for (int d = 0; d < 6; ++d) { gr[d] += df[d] * k; kg[d] += k * (-lg[j * 6 + d]); }
        }
      for (int off = tpp >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 6; ++d) { gr[d] += __shfl_xor(gr[d], off, kWave); kg[d] += __shfl_xor(kg[d], off, kWave); }
      }
      if (act && part == 0) {
#pragma unroll
        for (int d = 0; d < 6; ++d) lphi[pi * 6 + d] = (kg[d] + 2 / h * gr[d]) / P;
      }
    }
  } else {
    if (tid == 0)
      for (int d = 0; d < 6; ++d) lphi[d] = -lg[d];          // SVGDICP.cpp:112
  }
  __syncthreads();

  if (a.trN) {  // traces (tests only): newton slot carries the sgd gradient
    for (int e = tid; e < P * 6; e += UT) { a.trN[e] = lg[e]; a.trphi[e] = lphi[e]; }
    if (tid == 0) *a.trh = sel.h;
  }

  // ---- 3. optimizer step (param.grad = -stein_grad, SVGDICP.cpp:476-494), pose refresh, early stop ----
  double my_norm = 0.0;
  for (int p = tid; p < P; p += UT) my_norm += svgd_step_one(a, p, lphi + p * 6, lx + p * 6);
  bool stop = false;
  if (a.check_early_stop) {
    for (int off = 32; off > 0; off >>= 1) my_norm += __shfl_xor(my_norm, off, kWave);
    if (lane == 0) sh_norm[wave] = my_norm;
    __syncthreads();
    double m = 0.0;
    for (int i = 0; i < UT / kWave; ++i) m += sh_norm[i];
    m /= P;
    stop = (float)m < (float)a.conv_thr;                          // float32 compare (type promotion)
  }
  if (stop) {
    if (tid == 0) { a.ctl[0] = 1; a.ctl[1] = a.iteration + 1; }   // finish_iter_ = epoch + 1 (:128)
    return;
  }
  __syncthreads();
  for (int e = tid; e < 6 * P; e += UT) a.history[(size_t)a.iteration * 6 * P + e] = (float)a.pose_out[e];
}

// constructor / add_cloud: R = Exp(r), t, total pose (SVNICP.cpp:20-38, SVGDICP.cpp:46-62)
__global__ void k_init_particles(const double* __restrict__ init, int P, Pose0 pose, int mode, double* R, double* t,
                                 double* Rtot, double* pose_out, int refresh_pose, double* eul, BeginZero z) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  // start of a registration (svnicp_align_begin): the control words and the small areas that must start at zero, in this
  // launch instead of five fill / copy launches of their own (each one costs a small registration 5-8 us)
  for (int a = 0; a < z.n; ++a)
    for (unsigned int e = (unsigned int)p; e < z.dwords[a]; e += gridDim.x * blockDim.x) z.ptr[a][e] = 0u;
  if (z.ctl && p == 0) { z.ctl[0] = 0; z.ctl[1] = z.iterations; z.ctl[2] = 0; z.ctl[3] = 0; }   // stop flag, finish_iter (SVGDICP.cpp:42)
  if (p >= P) return;
  double r[3] = {0, 0, 0}, tv[3], Rm[9];
  if (mode == 2) {  // keep the current R_, t_: only the total pose is recomputed
    for (int i = 0; i < 9; ++i) Rm[i] = R[9 * p + i];
    for (int i = 0; i < 3; ++i) tv[i] = t[3 * p + i];
  } else {
    r[0] = init[3 * P + p]; r[1] = init[4 * P + p]; r[2] = init[5 * P + p];
    tv[0] = init[p]; tv[1] = init[P + p]; tv[2] = init[2 * P + p];
    if (mode == 0) so3_exp(r, Rm, nullptr); else euler_to_R(r[0], r[1], r[2], Rm);
    if (mode == 1 && eul) {  // SVGD: the optimizer parameters are the pose entries themselves (SVGDICP.cpp:46-53)
      for (int i = 0; i < 3; ++i) { eul[6 * p + i] = tv[i]; eul[6 * p + 3 + i] = r[i]; }
    }
  }
  double Rt[9], tt[3];
  mat3_mul(pose.R0, Rm, Rt);
  mat3_vec(pose.R0, tv, tt);
  for (int i = 0; i < 9; ++i) { R[9 * p + i] = Rm[i]; Rtot[12 * p + i] = Rt[i]; }
  for (int i = 0; i < 3; ++i) { t[3 * p + i] = tv[i]; Rtot[12 * p + 9 + i] = pose.t0[i] + tt[i]; }
  if (refresh_pose) {
    double lg[3];
    if (mode == 0) so3_log(Rm, lg); else { lg[0] = r[0]; lg[1] = r[1]; lg[2] = r[2]; }
    for (int i = 0; i < 3; ++i) { pose_out[i * P + p] = tv[i]; pose_out[(3 + i) * P + p] = lg[i]; }
  }
}

// get_transformation / get_distribution / get_cov_matrix / get_particle_weight
// (SVNICP.cpp:281-308; SVGDICP.cpp:497-524).  out = mean[6] var[6] cov[36] weights[P]
__global__ void k_stats(StatsArgs a) {
  const int tid = threadIdx.x;
  const int P = a.P;
  __shared__ double mean[6];
  // the sums below run over the particles in order, one thread per output: from LDS (a coalesced copy first) instead of
  // 3 x P dependent global loads (48 -> 9 us at 128 particles); same order of additions, same bits
  constexpr int kStage = 1024;
  __shared__ double sp[6 * kStage];
  const bool staged = P <= kStage;
  if (staged) for (int e = tid; e < 6 * P; e += blockDim.x) sp[e] = a.pose[e];
  __syncthreads();
  const double* pose = staged ? sp : a.pose;
  // SVNICP.cpp:46: torch::ones({P,1}) / P is float32, promoted to f64 in the products
  const double wsvn = (double)(1.0f / (float)P);
  if (tid < 6) {
    double s = 0.0;
    if (a.mode == 0) { for (int p = 0; p < P; ++p) s += pose[tid * P + p] * wsvn; }
    else { for (int p = 0; p < P; ++p) s += pose[tid * P + p]; s /= P; }
    mean[tid] = s;
    a.out[tid] = s;
  }
  __syncthreads();
  if (tid < 6) {
    double s = 0.0;
    if (a.mode == 0) { for (int p = 0; p < P; ++p) { const double d = pose[tid * P + p] - mean[tid]; s += d * d * wsvn; } }
    else { for (int p = 0; p < P; ++p) { const double d = pose[tid * P + p] - mean[tid]; s += d * d; } s /= (P - 1); }
    a.out[6 + tid] = s;
  }
  if (tid < 36) {
    const int r = tid / 6, c = tid % 6;
    double s = 0.0;
    const double wgt = a.mode == 0 ? wsvn : 1.0;
    for (int p = 0; p < P; ++p) s += wgt * ((pose[r * P + p] - mean[r]) * (pose[c * P + p] - mean[c]));
    a.out[12 + tid] = a.mode == 0 ? s : s / P;
  }
  for (int p = tid; p < P; p += blockDim.x) a.out[48 + p] = a.mode == 0 ? wsvn : 1.0;
}

}  // namespace

size_t update_workspace_doubles(int P) { return (size_t)P * (36 + 6 * 4) + (size_t)P * P + 64; }
size_t update_uctl_doubles(int P) { return (size_t)UCTL_NORM + (size_t)((P + 7) & ~7) + (size_t)HB_NB / 2 + (size_t)36 * ((P + 127) / 128) + 8; }

hipError_t launch_init_particles(const double* init6xP, int P, const Pose0& pose, int mode, double* R, double* t,
                                 double* Rtot, double* pose_out, int refresh_pose, double* eul, hipStream_t st, const BeginZero* zero) {
  BeginZero z{};
  if (zero) z = *zero;
  unsigned int most = 0;
  for (int a = 0; a < z.n; ++a) most = z.dwords[a] > most ? z.dwords[a] : most;
  int blocks = (P + 127) / 128;
  const int for_zero = (int)((most + 128u * 32u - 1u) / (128u * 32u));   // about 32 words per thread
  if (for_zero > blocks) blocks = for_zero > 256 ? 256 : for_zero;
  hipLaunchKernelGGL(k_init_particles, dim3(blocks), dim3(128), 0, st, init6xP, P, pose, mode, R, t, Rtot,
                     pose_out, refresh_pose, eul, z);
  return hipGetLastError();
}

// ---- the Stein step of 2 <= P particles as three pieces (api.hip sequences them) --------------------------------------
//   launch_update_median     pair statistics -> bandwidth h: needs the poses only; on the context's SECOND stream, beside the
//                            stage-B kernels of the same iteration
//   launch_update_prepare    H, b, Newton step, mean-Hessian inverse: needs the sums only
//   launch_update_direction  Stein direction + pose update per particle [+ k_upd_finish: early stop, traces, history]
hipError_t launch_update_median(const UpdateArgs& a, int num_cus, int max_p_one_workgroup, hipStream_t st) {
  const int P = a.P;
  if (P <= max_p_one_workgroup && P <= 128) {   // one workgroup, keys in registers (KREG)
    const size_t smem = (size_t)P * 6 * sizeof(double) + (size_t)(FRONT_BUF + 8) * sizeof(double) + (size_t)HB_NB * sizeof(unsigned int);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_upd_median), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_upd_median, dim3(1), dim3(UT), smem, st, a);
    return hipGetLastError();
  }
  const size_t n = (size_t)P * P;
  const size_t xs = (size_t)P * 6 * sizeof(double);
  // few, fat workgroups: each one merges its LDS histogram into the global one with atomics, and those contend
  int nb = (int)((n + 1023) / 1024);
  if (nb > num_cus / 2) nb = num_cus / 2;
  if (nb < 1) nb = 1;
  const size_t lds_hist = xs + (size_t)HB_NB * sizeof(unsigned int);
  const size_t lds_coll = xs + (size_t)COLL_CHUNK * sizeof(double);
  const size_t lds_sel = (size_t)SEL_LDS_KEYS * sizeof(double);
  if (lds_hist > 150 * 1024 || lds_coll > 150 * 1024) return hipErrorInvalidValue;  // P > ~2000 on a 256-CU part
  {  // raise the dynamic-LDS cap (per device; a cheap host-side call)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_upd_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_upd_collect), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_upd_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sel);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_upd_hist, dim3(nb), dim3(256), lds_hist, st, a);
  hipLaunchKernelGGL(k_upd_collect, dim3(nb), dim3(256), lds_coll, st, a);
  hipLaunchKernelGGL(k_upd_select, dim3(1), dim3(UT), lds_sel, st, a);
  return hipGetLastError();
}

hipError_t launch_update_prepare(const UpdateArgs& a, hipStream_t st) {
  const int need_mean = (!a.svgd && !a.full_grad) ? 1 : 0;   // only the default SVN branch preconditions with the mean Hessian
  hipLaunchKernelGGL(k_upd_prepare, dim3((a.P + PREP_PW - 1) / PREP_PW + need_mean), dim3(PREP_T), 0, st, a);
  return hipGetLastError();
}

// 2 <= P <= 128 only (the one-workgroup pair statistics)
hipError_t launch_update_prepare_median(const UpdateArgs& a, hipStream_t st) {
  const int P = a.P;
  const int need_mean = (!a.svgd && !a.full_grad) ? 1 : 0;
  const size_t smem = (size_t)P * 6 * sizeof(double) + (size_t)(FRONT_BUF + 8) * sizeof(double) + (size_t)HB_NB * sizeof(unsigned int);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_upd_prepare_median), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_upd_prepare_median, dim3((P + PREP_PW - 1) / PREP_PW + need_mean + 1), dim3(UT), smem, st, a);
  return hipGetLastError();
}

// ---- the persistent small-registration kernel: host side ----
namespace {
template <int PW, int WP, int NRB, bool TAIL>
hipError_t launch_small_t(const AccumArgs& a, const UpdateArgs& u, const SmallArgs& s, int grid, size_t smem, hipStream_t st) {
  const void* fn = u.svgd ? reinterpret_cast<const void*>(k_small_registration<PW, WP, NRB, TAIL, true>)
                          : reinterpret_cast<const void*>(k_small_registration<PW, WP, NRB, TAIL, false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, smem);
  if (e != hipSuccess) return e;
  if (per_cu < 1) return hipErrorCooperativeLaunchTooLarge;
  AccumArgs aa = a; UpdateArgs uu = u; SmallArgs ss = s;
  void* args[3] = {&aa, &uu, &ss};
  return hipLaunchCooperativeKernel(fn, dim3((unsigned)grid), dim3(256), args, (unsigned int)smem, st);
}
}  // namespace

// (PW, WP) of the plan and knn_count for which the persistent kernel is instantiated (the others run the four-launch chain)
bool small_registration_supported(int PW, int WP, int K) {
  return K >= 97 && K <= 100 && ((WP == 1 && (PW == 16 || PW == 32 || PW == 64)) || (PW == 64 && WP == 2));
}

// all iterations of a small registration in one cooperative launch; `bar`: three zeroed words (arrivals, generation, error)
hipError_t launch_small_registration(const AccumPlan& plan, AccumArgs a, const UpdateArgs& u, int iterations, unsigned int* bar,
                                     int num_cus, hipStream_t st) {
  const int P = u.P;
  a.Ppad = plan.Ppad; a.pts_per_block = plan.pts_per_block; a.spts_per_block = plan.pts_per_block;   // one slice of points per workgroup, both bodies
  SmallArgs s{};
  s.GA = plan.grid_x; s.iterations = iterations; s.bar = bar;
  {  // search slices: one pass of the four waves per workgroup at least, at most num_cus - 1 workgroups
    const int pass = 4 * (64 / plan.PW);
    int64_t spb = (a.B + (num_cus - 2)) / (num_cus - 1);
    spb = (spb + pass - 1) / pass * pass;
    s.spts_per_block = (int)spb;
    s.GS = (int)((a.B + spb - 1) / spb);
  }
  const int n_prep = (P + PREP_PW - 1) / PREP_PW + ((!u.svgd && !u.full_grad) ? 1 : 0);
  const int n_dir = (P + 3) / 4;
  int grid = s.GS + 1;
  if (s.GA > grid) grid = s.GA;
  if (n_prep > grid) grid = n_prep;
  if (n_dir > grid) grid = n_dir;
  if (grid > num_cus) return hipErrorCooperativeLaunchTooLarge;
  const size_t smem_median = (size_t)P * 6 * sizeof(double) + (size_t)(FRONT_BUF + 8) * sizeof(double) + (size_t)HB_NB * sizeof(unsigned int);
  const size_t smem = smem_median > plan.smem ? smem_median : plan.smem;
  if (!small_registration_supported(plan.PW, plan.WP, plan.K)) return hipErrorInvalidValue;
  if (plan.PW == 16) return launch_small_t<16, 1, 6, true>(a, u, s, grid, smem, st);
  if (plan.PW == 32) return launch_small_t<32, 1, 6, true>(a, u, s, grid, smem, st);
  if (plan.WP == 1) return launch_small_t<64, 1, 6, true>(a, u, s, grid, smem, st);
  return launch_small_t<64, 2, 6, true>(a, u, s, grid, smem, st);
}

hipError_t launch_update_direction(const UpdateArgs& a, hipStream_t st, bool finish) {
  hipLaunchKernelGGL(k_upd_direction, dim3((a.P + 3) / 4), dim3(256), 0, st, a);
  if (finish && (a.check_early_stop || a.trH)) hipLaunchKernelGGL(k_upd_finish, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}
const double* update_step_norms(const UpdateArgs& a) { return a.uctl + UCTL_NORM; }

hipError_t launch_reduce_partials(const double* partial, int nblk, int Ppad, int p_lo, int n_particles, double* sums, const int* ctl,
                                  hipStream_t st) {
  const int n_entries = n_particles * kNSums;
  if (n_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_reduce_partials, dim3((n_entries + 15) / 16), dim3(256), 0, st, partial, nblk, Ppad, p_lo, n_particles, sums, ctl);
  return hipGetLastError();
}

hipError_t launch_update(const UpdateArgs& a_in, hipStream_t st) {
  UpdateArgs a = a_in;
  const size_t base = (size_t)a.P * 24 * sizeof(double);           // x, N, b, phi
  const size_t with_h = base + (size_t)a.P * 36 * sizeof(double);  // + H
  a.h_in_lds = with_h <= 120 * 1024 ? 1 : 0;
  const size_t smem = a.h_in_lds ? with_h : base;
  if (smem > 150 * 1024) return hipErrorInvalidValue;  // P > 800: not supported by the one-workgroup update
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_particle_update),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_particle_update, dim3(1), dim3(UT), smem, st, a);
  return hipGetLastError();
}

hipError_t launch_update_svgd(const UpdateArgs& a, hipStream_t st) {
  const size_t smem = (size_t)a.P * 18 * sizeof(double);
  if (smem > 150 * 1024) return hipErrorInvalidValue;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_particle_update_svgd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_particle_update_svgd, dim3(1), dim3(UT), smem, st, a);
  return hipGetLastError();
}

hipError_t launch_stats(const StatsArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_stats, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace svnicp
