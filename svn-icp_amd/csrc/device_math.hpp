// device_math.hpp — small f64 device helpers shared by the kernels (gfx950, wave64).
// The whole library is compiled with -ffp-contract=off: a*b+c below is a rounded multiply
// followed by a rounded add, exactly like the reference CPU arithmetic; fused operations are
// written explicitly as fma() only where bit-parity with the reference is not required.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svnicp {

constexpr int kWave = 64;
constexpr int kNSums = 22;  // per-particle raw sums exchanged between GPUs (see stein_iter.hip)

struct Mat3 { double m[9]; };
struct Vec3 { double v[3]; };
struct Pose0 { double R0[9]; double t0[3]; };  // initial mean, SVGDICP.h:102-110

// C = A*B, evaluation order of torch.matmul restated in oracle/svnicp_oracle.c mat3_mul
__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3_vec(const double* A, const double* v, double* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
__device__ __forceinline__ void mat3T_vec(const double* A, const double* v, double* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}

// Rodrigues Exp + left Jacobian — reference SVNICP.cpp:166-194 (to_rotation_tensor, J_l_)
__device__ inline void so3_exp(const double* r, double* R, double* Jl) {
  const double angle = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  double a[3];
  if (angle < 1e-12) { a[0] = a[1] = a[2] = 0.0; }
  else { a[0] = r[0] / angle; a[1] = r[1] / angle; a[2] = r[2] / angle; }
  const double c = cos(angle), s = sin(angle);
  const double ah[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
  const double soa = s / angle;          // NaN for angle == 0, as in the reference (:188)
  const double omc_a = (1 - c) / angle;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double I = (i == j) ? 1.0 : 0.0;
      const double aa = a[i] * a[j];
      R[3 * i + j] = (c * I + (1 - c) * aa) + s * ah[3 * i + j];
      if (Jl) Jl[3 * i + j] = (soa * I + (1 - soa) * aa) + omc_a * ah[3 * i + j];
    }
}

// SO(3) Log — reference SVNICP.cpp:196-215 (rotm_to_ypr_tensor)
__device__ inline void so3_log(const double* R, double* w) {
  double c = 0.5 * (R[0] + R[4] + R[8] - 1);
  c = (c < -1) ? -1 : c;
  c = (c > 1) ? 1 : c;
  const double angle = acos(c);
  const double sa = sin(angle);
  const bool nonzero = fabs(sa) > 1e-12;
  const double f = 0.5 / (nonzero ? sa : 1.0) * angle;
  w[0] = f * (R[7] - R[5]);
  w[1] = f * (R[2] - R[6]);
  w[2] = f * (R[3] - R[1]);
  if (!nonzero) { w[0] = w[1] = w[2] = 0.0; }
}

// Rz(yaw) Ry(pitch) Rx(roll) — reference SVGDICP.cpp:226-260
__device__ inline void euler_to_R(double roll, double pitch, double yaw, double* R) {
  const double A = cos(yaw), Bs = sin(yaw), C = cos(pitch), D = sin(pitch), E = cos(roll), F = sin(roll);
  R[0] = C * A;  R[1] = F * D * A - E * Bs;  R[2] = F * Bs + E * D * A;
  R[3] = C * Bs; R[4] = E * A + F * D * Bs;  R[5] = E * D * Bs - F * A;
  R[6] = -D;     R[7] = F * C;               R[8] = E * C;
}

// 6x6 LU with partial pivoting (what at::linalg_solve / linalg_inv do through LAPACK gesv);
// reference call sites SVNICP.cpp:162,225,250.  A is overwritten.  Every loop is fully
// unrolled and the row exchange is a predicated swap, so A/piv/x stay in registers (a
// runtime-indexed local array would be demoted to scratch memory).
__device__ __forceinline__ bool lu6(double* A, int* piv) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double mx = fabs(A[6 * k + k]);
#pragma unroll
    for (int r = k + 1; r < 6; ++r) {
      const double v = fabs(A[6 * r + k]);
      if (v > mx) { mx = v; p = r; }
    }
    piv[k] = p;
#pragma unroll
    for (int r = k + 1; r < 6; ++r) {
      const bool sw = (p == r);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double a = A[6 * k + c], b = A[6 * r + c];
        A[6 * k + c] = sw ? b : a;
        A[6 * r + c] = sw ? a : b;
      }
    }
    if (A[6 * k + k] == 0.0) ok = false;
    const double inv = 1.0 / A[6 * k + k];
#pragma unroll
    for (int r = k + 1; r < 6; ++r) A[6 * r + k] *= inv;
#pragma unroll
    for (int r = k + 1; r < 6; ++r)
#pragma unroll
      for (int c = k + 1; c < 6; ++c) A[6 * r + c] -= A[6 * r + k] * A[6 * k + c];
  }
  return ok;
}
__device__ __forceinline__ void lu6_solve(const double* LU, const int* piv, double* x) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int r = k + 1; r < 6; ++r) {
      const bool sw = (piv[k] == r);
      const double a = x[k], b = x[r];
      x[k] = sw ? b : a;
      x[r] = sw ? a : b;
    }
  }
#pragma unroll
  for (int r = 1; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < r; ++c) x[r] -= LU[6 * r + c] * x[c];
#pragma unroll
  for (int r = 5; r >= 0; --r) {
#pragma unroll
    for (int c = r + 1; c < 6; ++c) x[r] -= LU[6 * r + c] * x[c];
    x[r] /= LU[6 * r + r];
  }
}

}  // namespace svnicp
