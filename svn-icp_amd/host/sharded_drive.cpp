// sharded_drive.cpp — one rank of a multi-GPU Stein-ICP registration in C++: RCCL communicator, svnicp::Sharded<>
// (sharded_registration.hpp) over the split-phase C ABI, result dump.  The C++ counterpart of bench.py --gpus N / of
// svn-icp_amd/sharded.py; tests/test_gpu_parity.py::test_cpp_rccl_host_world1 runs it as a single rank and holds the
// result to the bits of svnicp_align.
//   hipcc -std=c++17 -Wall -Werror -I include -I svn-icp_amd/host svn-icp_amd/host/sharded_drive.cpp \
//         -L svn-icp_amd -lsvnicp_hip -lrccl -Wl,-rpath,'$ORIGIN/..' -o svn-icp_amd/host/sharded_drive
//   RANK=r WORLD_SIZE=W LOCAL_RANK=d sharded_drive case.bin out.bin [rows|particles] [svn|svgd] [id_file] [repeat]
// case.bin : int64 B, M; int32 P, iterations, knn, full_grad, early_stop; f64 lr, max_dist, threshold;
//            f64 src[3B], tgt[3M], init[6P]
// out.bin  : int32 state, iterations_run; f64 mean[6], cov[36], particles[6P]; int64 local_rows; int32 cand[local_rows][knn]
// World size > 1: rank 0 writes the ncclUniqueId to id_file, the other ranks wait for it (no MPI / torch needed).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sharded_registration.hpp"

template <typename T> static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
template <typename T> static void wr(FILE* f, const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 4; } } while (0)
#define NCCLOK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 5; } } while (0)

template <class SOLVER>
static int run(const svnicp::SteinICPParam& prm, svnicp::Split split, int device, ncclComm_t comm, int rank, int world, hipStream_t stream,
               const std::vector<double>& src, int64_t B, const std::vector<double>& tgt, int64_t M, const std::vector<double>& init, int P,
               int repeat, FILE* fo) {
  svnicp::Sharded<SOLVER> s(prm, init, device, comm, rank, world, stream, split);
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t0[3] = {0, 0, 0};
  int state = 0;
  double best_ms = 1e30;
  for (int rep = 0; rep < repeat; ++rep) {
    const auto t_start = std::chrono::steady_clock::now();
    s.add_cloud(src.data(), B, tgt.data(), M, init.data(), P);
    s.set_initial_mean(I3, t0);
    state = (int)s.stein_align();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    best_ms = ms < best_ms ? ms : best_ms;
  }
  int iters = 0;
  if (svnicp_get_iterations_run(s.solver().handle(), &iters) != 0) { fprintf(stderr, "svnicp_get_iterations_run failed\n"); return 6; }
  const auto mean = s.solver().get_transformation();
  const auto cov = s.solver().get_cov_matrix();
  const auto part = s.solver().get_particles();
  const int64_t rows = s.local_rows();
  std::vector<int32_t> cand((size_t)rows * prm.KNN_count);
  if (svnicp_get_candidates(s.solver().handle(), cand.data()) != 0) { fprintf(stderr, "svnicp_get_candidates failed\n"); return 6; }
  wr(fo, &state, 1); wr(fo, &iters, 1);
  wr(fo, mean.data(), 6); wr(fo, cov.data(), 36); wr(fo, part.data(), part.size());
  wr(fo, &rows, 1); wr(fo, cand.data(), cand.size());
  printf("rank %d/%d: state %d, %d iterations, %lld local rows, best of %d: %.3f ms per registration (host wall, upload included); mean t = %.6f %.6f %.6f\n",
         rank, world, state, iters, (long long)rows, repeat, best_ms, mean[0], mean[1], mean[2]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: RANK=r WORLD_SIZE=W LOCAL_RANK=d %s case.bin out.bin [rows|particles] [svn|svgd] [id_file] [repeat]\n", argv[0]); return 64; }
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", 0);
  const svnicp::Split split = (argc > 3 && std::string(argv[3]) == "particles") ? svnicp::Split::Particles : svnicp::Split::Rows;
  const bool svgd = argc > 4 && std::string(argv[4]) == "svgd";
  const char* id_file = argc > 5 ? argv[5] : nullptr;
  const int repeat = argc > 6 ? atoi(argv[6]) : 1;
  FILE* fi = fopen(argv[1], "rb");
  if (!fi) { perror(argv[1]); return 65; }
  int64_t B = 0, M = 0;
  int32_t P = 0, iterations = 0, knn = 0, full = 0, es = 0;
  double lr = 0, max_dist = 0, thr = 0;
  if (!rd(fi, &B, 1) || !rd(fi, &M, 1) || !rd(fi, &P, 1) || !rd(fi, &iterations, 1) || !rd(fi, &knn, 1) || !rd(fi, &full, 1) || !rd(fi, &es, 1) ||
      !rd(fi, &lr, 1) || !rd(fi, &max_dist, 1) || !rd(fi, &thr, 1)) return 66;
  std::vector<double> src((size_t)3 * B), tgt((size_t)3 * M), init((size_t)6 * P);
  if (!rd(fi, src.data(), src.size()) || !rd(fi, tgt.data(), tgt.size()) || !rd(fi, init.data(), init.size())) return 66;
  fclose(fi);

  HIPOK(hipSetDevice(device));
  hipStream_t stream;
  HIPOK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  // communicator: also at world size 1 (one rank, ncclCommInitRank) so that the single-GPU test runs the same code
  ncclUniqueId id;
  if (rank == 0) {
    NCCLOK(ncclGetUniqueId(&id));
    if (world > 1) {
      if (!id_file) { fprintf(stderr, "world size > 1 needs an id_file to hand the ncclUniqueId to the other ranks\n"); return 64; }
      const std::string tmp = std::string(id_file) + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f) { perror(tmp.c_str()); return 65; }
      wr(f, &id, 1); fclose(f);
      if (rename(tmp.c_str(), id_file) != 0) { perror("rename"); return 65; }
    }
  } else {
    if (!id_file) { fprintf(stderr, "world size > 1 needs an id_file\n"); return 64; }
    bool got = false;
    for (int tries = 0; tries < 6000 && !got; ++tries) {   // up to a minute
      FILE* f = fopen(id_file, "rb");
      if (f) { got = rd(f, &id, 1); fclose(f); }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    if (!got) { fprintf(stderr, "rank %d: no ncclUniqueId in %s\n", rank, id_file); return 67; }
  }
  ncclComm_t comm;
  NCCLOK(ncclCommInitRank(&comm, world, id, rank));
  int nccl_world = 0;
  NCCLOK(ncclCommCount(comm, &nccl_world));
  if (nccl_world != world) { fprintf(stderr, "communicator has %d ranks, expected %d\n", nccl_world, world); return 5; }

  svnicp::SteinICPParam prm;
  prm.iterations = iterations; prm.KNN_count = knn; prm.lr = lr; prm.max_dist = max_dist; prm.SVN_full_grad = full != 0;
  prm.check_early_stop = es != 0; prm.convergence_threshold = thr; prm.optimizer = "Adam";
  FILE* fo = fopen(argv[2], "wb");
  if (!fo) { perror(argv[2]); return 65; }
  int rc = 0;
  try {
    rc = svgd ? run<svnicp::SVGDICP>(prm, split, device, comm, rank, world, stream, src, B, tgt, M, init, P, repeat, fo)
              : run<svnicp::SVNICP>(prm, split, device, comm, rank, world, stream, src, B, tgt, M, init, P, repeat, fo);
  } catch (const std::exception& e) {
    fprintf(stderr, "svnicp: %s\n", e.what());
    rc = 3;   // e.g. no gfx950 device: the library has no CPU path
  }
  fclose(fo);
  NCCLOK(ncclCommDestroy(comm));
  HIPOK(hipStreamDestroy(stream));
  return rc;
}
