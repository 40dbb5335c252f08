// update_single.hpp — device code shared by the Stein-step kernels (particle_update.hip) and the fused stage-B kernels
// (stein_iter.hip): the finalisation of H, b from the 22 raw sums, and the complete Stein step of a ONE-particle
// registration (plain ICP through the solver, BASELINE configuration C1), which the accumulate kernel's last workgroup
// performs itself.
#pragma once
#include "kernels.hpp"

namespace svnicp {
namespace {

// H (6x6) and b (6) of one particle from its 22 raw sums and Rc = R0·R  (see stein_iter.hip)
__device__ inline void finalize_Hb(const double* s, const double* Rc, double* H, double* b) {
  const double sw = s[0];
  const double a0 = s[1], a1 = s[2], a2 = s[3];
  const double xx = s[4], xy = s[5], xz = s[6], yy = s[7], yz = s[8], zz = s[9];
  const double tr = xx + yy + zz;
#pragma unroll
  for (int i = 0; i < 36; ++i) H[i] = 0.0;
  H[0] = H[7] = H[14] = sw;                      // Σ w·I
  // top-right −Σw·ŝ, bottom-left +Σw·ŝ with ŝ = [[0,−s2,s1],[s2,0,−s0],[−s1,s0,0]]
  H[0 * 6 + 4] = a2;  H[0 * 6 + 5] = -a1;
  H[1 * 6 + 3] = -a2; H[1 * 6 + 5] = a0;
  H[2 * 6 + 3] = a1;  H[2 * 6 + 4] = -a0;
  H[3 * 6 + 1] = -a2; H[3 * 6 + 2] = a1;
  H[4 * 6 + 0] = a2;  H[4 * 6 + 2] = -a0;
  H[5 * 6 + 0] = -a1; H[5 * 6 + 1] = a0;
  // bottom-right Σw(‖s‖²I − ssᵀ)
  H[3 * 6 + 3] = tr - xx; H[3 * 6 + 4] = -xy;     H[3 * 6 + 5] = -xz;
  H[4 * 6 + 3] = -xy;     H[4 * 6 + 4] = tr - yy; H[4 * 6 + 5] = -yz;
  H[5 * 6 + 3] = -xz;     H[5 * 6 + 4] = -yz;     H[5 * 6 + 5] = tr - zz;
#pragma unroll
  for (int i = 0; i < 6; ++i) H[7 * i] += 1e-6;  // SVNICP.cpp:153
  // b_t = Rcᵀ Σwe ; b_r = vee-part of G = Rcᵀ·C, C[i][j] = Σ (we)_i s_j
  mat3T_vec(Rc, s + 10, b);
  double G[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) G[3 * i + j] = Rc[i] * s[13 + j] + Rc[3 + i] * s[16 + j] + Rc[6 + i] * s[19 + j];
  b[3] = G[7] - G[5];  // s_y u_z − s_z u_y  with u_i s_j = G[i][j]
  b[4] = G[2] - G[6];
  b[5] = G[3] - G[1];
}


// P = 1 (SVNICP.cpp:81-89: no kernel, no repulsion — the Stein direction is the Newton step itself, phi = −H⁻¹b): finalise
// H, b, solve, update the pose (SVNICP.cpp:268-279), early-stop test, history and traces — the body of k_particle_update
// for its only particle, run by ONE thread.  `s` = the particle's 22 reduced sums.
__device__ inline void update_single_particle(const UpdateArgs& a, const double* s) {
  double Rc[9], H[36], b[6], LU[36], x6[6], phi[6];
  int piv[6];
  mat3_mul(a.pose.R0, a.R, Rc);
  finalize_Hb(s, Rc, H, b);
#pragma unroll
  for (int i = 0; i < 36; ++i) LU[i] = H[i];
  const bool ok = lu6(LU, piv);
#pragma unroll
  for (int i = 0; i < 6; ++i) x6[i] = b[i];
  lu6_solve(LU, piv, x6);                                   // SVNICP.cpp:162
#pragma unroll
  for (int i = 0; i < 6; ++i) { x6[i] = ok ? x6[i] : __builtin_nan(""); phi[i] = -x6[i]; }   // SVNICP.cpp:89
  if (a.trH) {   // traces (tests only)
#pragma unroll
    for (int i = 0; i < 36; ++i) a.trH[i] = H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { a.trb[i] = b[i]; a.trN[i] = x6[i]; a.trphi[i] = phi[i]; }
    *a.trh = __builtin_nan("");                             // no bandwidth with one particle
  }
  double dR[9], Jl[9], dt[3], Rn[9], Rdt[3], Ro[9], tn[3];
  so3_exp(phi + 3, dR, Jl);
  mat3_vec(Jl, phi, dt);
#pragma unroll
  for (int i = 0; i < 9; ++i) Ro[i] = a.R[i];
  mat3_mul(Ro, dR, Rn);
  mat3_vec(Rn, dt, Rdt);                                    // uses the UPDATED R (:277-278)
#pragma unroll
  for (int i = 0; i < 3; ++i) tn[i] = Rdt[i] + a.t[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) a.R[i] = Rn[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) a.t[i] = tn[i];
  double Rt[9], tt[3];                                      // next iteration's total pose (SVNICP.cpp:58-59)
  mat3_mul(a.pose.R0, Rn, Rt);
  mat3_vec(a.pose.R0, tn, tt);
#pragma unroll
  for (int i = 0; i < 9; ++i) a.Rtot[i] = Rt[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) a.Rtot[9 + i] = a.pose.t0[i] + tt[i];
  double n2 = 0.0, lg[3];
#pragma unroll
  for (int d = 0; d < 6; ++d) n2 += phi[d] * phi[d];
  so3_log(Rn, lg);                                          // pose_particles_ = [t ; Log R] (SVNICP.cpp:103-106)
#pragma unroll
  for (int i = 0; i < 3; ++i) { a.pose_out[i] = tn[i]; a.pose_out[3 + i] = lg[i]; }
  if (a.check_early_stop && (float)sqrt(n2) < (float)a.conv_thr) {   // float32 compare (SVNICP.cpp:42,96-97)
    a.ctl[0] = 1; a.ctl[1] = a.iteration + 1;
    return;                                                 // the stopping epoch's history row stays zero
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { a.history[(size_t)a.iteration * 6 + i] = (float)tn[i]; a.history[(size_t)a.iteration * 6 + 3 + i] = (float)lg[i]; }
}

}  // namespace
}  // namespace svnicp
