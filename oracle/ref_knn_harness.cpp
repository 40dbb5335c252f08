// ref_knn_harness.cpp — extern "C" door onto the reference's unmodified
// KNearestNeighborIdxCpu (/root/reference/svn-icp/src/core/knn/knn_cpu.cpp:13-69).
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The reference function is float32-only
// (accessor<float,3>, knn_cpu.cpp:28-33), so this pins the selection / tie / ordering
// semantics of oracle's orc_knn_topk on float32-exact inputs.
#include <torch/torch.h>
#include <tuple>
#include <cstdint>
#include <cstring>

std::tuple<at::Tensor, at::Tensor> KNearestNeighborIdxCpu(
    const at::Tensor& p1, const at::Tensor& p2, const at::Tensor& lengths1,
    const at::Tensor& lengths2, const int norm, const int K);

extern "C" int ref_knn_cpu_f32(const float* p1, int64_t P1, const float* p2, int64_t P2, int K,
                               int64_t* idx_out, float* dist_out) {
  try {
    auto a = torch::from_blob(const_cast<float*>(p1), {1, P1, 3}, torch::kFloat32).clone();
    auto b = torch::from_blob(const_cast<float*>(p2), {1, P2, 3}, torch::kFloat32).clone();
    auto l1 = torch::full({1}, P1, torch::kInt64);
    auto l2 = torch::full({1}, P2, torch::kInt64);
    auto r = KNearestNeighborIdxCpu(a, b, l1, l2, 2, K);
    auto idx = std::get<0>(r).contiguous();
    auto dst = std::get<1>(r).contiguous();
    std::memcpy(idx_out, idx.data_ptr<int64_t>(), sizeof(int64_t) * P1 * K);
    std::memcpy(dist_out, dst.data_ptr<float>(), sizeof(float) * P1 * K);
    return 0;
  } catch (...) {
    return -1;
  }
}
