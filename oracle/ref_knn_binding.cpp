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

// KNearestNeighborBackwardCpu (knn_cpu.cpp:68-117): gradients of the squared distances w.r.t. both point sets
std::tuple<at::Tensor, at::Tensor> KNearestNeighborBackwardCpu(const at::Tensor& p1, const at::Tensor& p2, const at::Tensor& lengths1,
                                                               const at::Tensor& lengths2, const at::Tensor& idxs, const at::Tensor& grad_dists);

extern "C" int ref_knn_backward_cpu(const float* q, int64_t n, const float* p, int64_t m, int K, const int64_t* idx, const float* grad_d2,
                                    float* grad_q, float* grad_p) {
  auto fopt = torch::TensorOptions().dtype(torch::kFloat32);
  at::Tensor p1 = torch::from_blob(const_cast<float*>(q), {1, n, 3}, fopt);
  at::Tensor p2 = torch::from_blob(const_cast<float*>(p), {1, m, 3}, fopt);
  at::Tensor ix = torch::from_blob(const_cast<int64_t*>(idx), {1, n, K}, torch::TensorOptions().dtype(torch::kInt64));
  at::Tensor gd = torch::from_blob(const_cast<float*>(grad_d2), {1, n, K}, fopt);
  at::Tensor l1 = torch::full({1}, n, torch::kInt64), l2 = torch::full({1}, m, torch::kInt64);
  auto r = KNearestNeighborBackwardCpu(p1, p2, l1, l2, ix, gd);
  at::Tensor g1 = std::get<0>(r).contiguous(), g2 = std::get<1>(r).contiguous();
  std::memcpy(grad_q, g1.data_ptr<float>(), sizeof(float) * n * 3);
  std::memcpy(grad_p, g2.data_ptr<float>(), sizeof(float) * m * 3);
  return 0;
}
