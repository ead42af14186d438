// Binding TU for the REFERENCE's own CPU KNN op (test infrastructure; build container only).
// Compiles /root/reference/nerf_loc/models/ops/knn/src/knn_cpu.cpp where it lies (never copied) and
// exposes KNearestNeighborIdxCpu (knn_cpu.cpp:13) through a C entry point working on raw buffers.
// knn_api.cpp itself cannot be used: it pulls knn.h -> CUDA headers.
#include <torch/torch.h>
#include <tuple>

std::tuple<at::Tensor, at::Tensor> KNearestNeighborIdxCpu(const at::Tensor& p1, const at::Tensor& p2,
                                                          const at::Tensor& lengths1, const at::Tensor& lengths2, int K);

extern "C" int ref_knn_cpu(const float* q, int64_t n, const float* p, int64_t m, int K, float* out_d2, int64_t* out_idx) {
  auto fopt = torch::TensorOptions().dtype(torch::kFloat32);
  at::Tensor p1 = torch::from_blob(const_cast<float*>(q), {1, n, 3}, fopt);
  at::Tensor p2 = torch::from_blob(const_cast<float*>(p), {1, m, 3}, fopt);
  at::Tensor l1 = torch::full({1}, n, torch::kInt64), l2 = torch::full({1}, m, torch::kInt64);
  auto r = KNearestNeighborIdxCpu(p1, p2, l1, l2, K);
  at::Tensor idx = std::get<0>(r).contiguous(), d = std::get<1>(r).contiguous();
  std::memcpy(out_idx, idx.data_ptr<int64_t>(), sizeof(int64_t) * n * K);
  std::memcpy(out_d2, d.data_ptr<float>(), sizeof(float) * n * K);
  return 0;
}
