// example_register.cpp — the call sequence of OdometryPipeline::ICP_processing
// (/root/reference/svn-icp/src/core/OdometryPipeline.cpp:573-607) on the C++ shim.
//   g++ -std=c++17 -I include -I svn-icp_amd/host example_register.cpp -L svn-icp_amd -lsvnicp_hip
// Registers a noisy, displaced copy of a synthetic surface and prints mean pose + covariance diagonal.
#include <cmath>
#include <cstdio>
#include <random>

#include "svnicp_hip_shim.hpp"

int main() {
  const int P = 16, B = 4000, M = 12000;
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-5.0, 5.0);
  std::normal_distribution<double> N(0.0, 0.005);
  std::vector<double> tgt(3 * (size_t)M), src(3 * (size_t)B), init(6 * (size_t)P, 0.0);
  for (int i = 0; i < M; ++i) { const double x = U(rng), y = U(rng); tgt[3 * i] = x; tgt[3 * i + 1] = y; tgt[3 * i + 2] = 0.5 * std::sin(x) * std::cos(0.7 * y); }
  const double off[3] = {0.05, -0.03, 0.02};
  for (int i = 0; i < B; ++i) for (int d = 0; d < 3; ++d) src[3 * i + d] = tgt[3 * (size_t)(i * 3) + d] - off[d] + N(rng);
  std::uniform_real_distribution<double> Upos(-0.05, 0.05);
  for (int p = 1; p < P; ++p) for (int d = 0; d < 3; ++d) init[d * P + p] = Upos(rng);
  try {
    svnicp::SteinICPParam prm; prm.iterations = 15; prm.lr = 1.0; prm.KNN_count = 20; prm.SVN_full_grad = false;
    svnicp::SVNICP solver(prm, init, svnicp::ParticleWeightOpt{});
    solver.add_cloud(src.data(), B, tgt.data(), M, init.data(), P);
    const double R0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t0[3] = {0, 0, 0};
    solver.set_initial_mean(R0, t0);
    if (solver.stein_align() != svnicp::ALIGN_SUCCESS) return 2;
    const auto m = solver.get_transformation();
    const auto c = solver.get_cov_matrix();
    std::printf("mean %.5f %.5f %.5f  %.5f %.5f %.5f\n", m[0], m[1], m[2], m[3], m[4], m[5]);
    std::printf("var  %.3e %.3e %.3e\n", c[0], c[7], c[14]);
    const bool ok = std::fabs(m[0] - off[0]) < 5e-3 && std::fabs(m[1] - off[1]) < 5e-3 && std::fabs(m[2] - off[2]) < 5e-3;
    std::printf("%s\n", ok ? "OK" : "MISMATCH");
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "svnicp: %s\n", e.what());
    return 3;  // e.g. no gfx950 device: the library has no CPU path
  }
}
