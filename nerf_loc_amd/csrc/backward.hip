// Backward kernels, first slice: the two reductions of the gradient path that are not GEMM-shaped.
//   composite_backward_kernel  gradient of front-to-back alpha compositing (conditional_nerf/model.py:544-560,597: weights, rgb, depth,
//                              depth uncertainty, composited feature) w.r.t. the density, the per-sample colours and the per-sample feature
//                              rows — one wave per ray, the forward's transmittance is recomputed (nothing is saved by the forward pass)
//   knn_backward_kernel        KNearestNeighborBackwardKernel (ops/knn/src/knn.cu:449-490): d(squared distance)/d(query, support point)
// Both are called from nerf_loc_amd/diff_render.py (autograd.Function) when the gradient path runs on the GPU, and are checked against
// PyTorch autograd of the same expressions and, end to end, against the reference's autograd goldens (tests/test_backward_kernels.py,
// tests/test_diff_render.py).
#include "common.h"

namespace {

// w_s = a_s T_s, a_s = 1 - exp(-delta_s sigma_s), T_s = prod_{j<s} (1 - a_j).  With q_s = dL/dw_s:
//   dL/dsigma_s = delta_s ( q_s T_{s+1} - B_s ),  B_s = sum_{j>s} q_j w_j          (no division: exp(-delta sigma) may underflow to 0)
// q_s = g_rgb . rgb_s - [white] sum_c g_rgb_c + g_depth z_s + g_unc ((z_s - D)^2 - 2 D (1 - W) z_s) + g_feat . ft_s + g_w_s
//   (D = sum w z, W = sum w; the uncertainty sum_s w_s (z_s - D)^2 depends on w also through D)
template <int CH>
__global__ __launch_bounds__(256) void composite_backward_kernel(const float* __restrict__ z_vals, const float* __restrict__ sigma,
                                                                 const float* __restrict__ rgb_s, const float* __restrict__ ft, int R, int S, int C,
                                                                 int white_bkgd, const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                                 const float* __restrict__ g_unc, const float* __restrict__ g_feat,
                                                                 const float* __restrict__ g_w, float* __restrict__ g_sigma,
                                                                 float* __restrict__ g_rgb_s, float* __restrict__ g_ft) {
  __shared__ float wsh[4][256];   // weights, then q
  __shared__ float qsh[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wv;
  if (r >= R) return;
  const float* z = z_vals + (size_t)r * S;
  float zs[CH], al[CH], dl[CH], w[CH], Tn[CH];   // Tn = T_{s+1}
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) {
      zs[j] = z[s];
      dl[j] = (s + 1 < S) ? z[s + 1] - zs[j] : 1e2f;
      al[j] = 1.f - expf(-dl[j] * sigma[(size_t)r * S + s]);
      prod *= (1.f - al[j]);
    } else { zs[j] = 0.f; al[j] = 0.f; dl[j] = 0.f; }
  }
  float inc = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(inc, o, 64);
    if (lane >= o) inc *= t;
  }
  float T = __shfl_up(inc, 1, 64);
  if (lane == 0) T = 1.f;
  float wsum = 0.f, dsum = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) {
      w[j] = al[j] * T;
      T *= (1.f - al[j]);
      Tn[j] = T;
      wsum += w[j];
      dsum += w[j] * zs[j];
      wsh[wv][s] = w[j];
    } else { w[j] = 0.f; Tn[j] = 0.f; }
  }
  wsum = wave_sum(wsum);
  dsum = wave_sum(dsum);
  __builtin_amdgcn_wave_barrier();
  // ---- q_s: the feature term needs a C-long dot product per sample (lanes over channels, one wave reduction per sample); the
  // per-sample gradient rows g_ft = w_s g_feat and g_rgb_s = w_s g_rgb are written on the way
  const float gr = g_rgb ? g_rgb[3 * (size_t)r] : 0.f, gg = g_rgb ? g_rgb[3 * (size_t)r + 1] : 0.f, gb = g_rgb ? g_rgb[3 * (size_t)r + 2] : 0.f;
  const float gd = g_depth ? g_depth[r] : 0.f, gu = g_unc ? g_unc[r] : 0.f;
  if (ft && g_feat) {
    const float* gf = g_feat + (size_t)r * C;
    for (int s = 0; s < S; ++s) {
      const float* row = ft + ((size_t)r * S + s) * C;
      const float ws = wsh[wv][s];
      float acc = 0.f;
      for (int c = lane; c < C; c += 64) {
        const float g = gf[c];
        acc += g * row[c];
        if (g_ft) g_ft[((size_t)r * S + s) * C + c] = ws * g;
      }
      acc = wave_sum(acc);
      if (lane == 0) qsh[wv][s] = acc;
    }
  } else {
    for (int s = lane; s < S; s += 64) qsh[wv][s] = 0.f;
    if (g_ft) for (size_t i = lane; i < (size_t)S * C; i += 64) g_ft[(size_t)r * S * C + i] = 0.f;
  }
  __builtin_amdgcn_wave_barrier();
  float q[CH], suf = 0.f;   // suf: this lane's sum of q w over its own samples
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    q[j] = 0.f;
    if (s < S) {
      const float* c = rgb_s + 3 * ((size_t)r * S + s);
      const float dz = zs[j] - dsum;
      float v = qsh[wv][s] + gr * c[0] + gg * c[1] + gb * c[2] + gd * zs[j] + gu * (dz * dz - 2.f * dsum * (1.f - wsum) * zs[j]);
      if (white_bkgd) v -= gr + gg + gb;
      if (g_w) v += g_w[(size_t)r * S + s];
      q[j] = v;
      suf += v * w[j];
      if (g_rgb_s) { float* o = g_rgb_s + 3 * ((size_t)r * S + s); o[0] = w[j] * gr; o[1] = w[j] * gg; o[2] = w[j] * gb; }
    }
  }
  // exclusive suffix sum over lanes of the lane totals: B(lane) = sum over lanes > lane
  float run = suf;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(run, o, 64);
    if (lane + o < 64) run += t;
  }
  float B = run - suf;   // samples of the lanes behind this one
#pragma unroll
  for (int j = CH - 1; j >= 0; --j) {
    const int s = lane * CH + j;
    if (s < S) {
      g_sigma[(size_t)r * S + s] = dl[j] * (q[j] * Tn[j] - B);
      B += q[j] * w[j];
    }
  }
}

// one thread per (query, neighbour): grad_p1[n] += 2 g (p1 - p2), grad_p2[idx] -= the same (atomics, like the reference); neighbours
// k >= M (fewer support points than K: padded slots) and negative indices are ignored
__global__ void knn_backward_kernel(const float* __restrict__ xyz, const float* __restrict__ sp, const int* __restrict__ idx,
                                    const float* __restrict__ g_d2, int N, int K, int M, float* __restrict__ g_xyz, float* __restrict__ g_sp) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float x = xyz[3 * (size_t)n], y = xyz[3 * (size_t)n + 1], zc = xyz[3 * (size_t)n + 2];
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int k = 0; k < K && k < M; ++k) {
    const int i = idx[(size_t)n * K + k];
    if (i < 0 || i >= M) continue;
    const float g = 2.f * g_d2[(size_t)n * K + k];
    const float dx = g * (x - sp[3 * (size_t)i]), dy = g * (y - sp[3 * (size_t)i + 1]), dz = g * (zc - sp[3 * (size_t)i + 2]);
    ax += dx; ay += dy; az += dz;
    if (g_sp) { atomicAdd(g_sp + 3 * (size_t)i, -dx); atomicAdd(g_sp + 3 * (size_t)i + 1, -dy); atomicAdd(g_sp + 3 * (size_t)i + 2, -dz); }
  }
  if (g_xyz) { g_xyz[3 * (size_t)n] = ax; g_xyz[3 * (size_t)n + 1] = ay; g_xyz[3 * (size_t)n + 2] = az; }
}

}  // namespace

extern "C" {

int nl_composite_backward(const float* z_vals, const float* sigma, const float* rgb_s, const float* ft, int64_t R, int S, int C, int white_bkgd,
                          const float* g_rgb, const float* g_depth, const float* g_unc, const float* g_feat, const float* g_weights, float* g_sigma,
                          float* g_rgb_s, float* g_ft, void* stream) {
  if (R == 0) return NL_OK;
  if (!z_vals || !sigma || !rgb_s || !g_sigma || R < 0 || S < 1 || S > 256 || C < 0 || ((g_feat || g_ft) && !ft) || R > 0x7fffffffll / S) return NL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)nl_cdiv(R, 4));
#define NL_CB(CH) hipLaunchKernelGGL(composite_backward_kernel<CH>, grid, dim3(256), 0, st, z_vals, sigma, rgb_s, ft, (int)R, S, C, white_bkgd, g_rgb, \
                                     g_depth, g_unc, g_feat, g_weights, g_sigma, g_rgb_s, g_ft)
  if (S <= 64) NL_CB(1); else if (S <= 128) NL_CB(2); else if (S <= 192) NL_CB(3); else NL_CB(4);
#undef NL_CB
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_knn_backward(const float* xyz, const float* sp_xyz, const int32_t* idx, const float* g_d2, int64_t N, int K, int64_t M, float* g_xyz,
                    float* g_sp_xyz, void* stream) {
  if (N == 0) return NL_OK;
  if (!xyz || !idx || !g_d2 || N < 0 || K < 1 || K > NL_KNN_MAX_K || M < 0 || (M > 0 && !sp_xyz) || N > 0x7fffffffll / K || M > 0x7fffffffll) return NL_ERR_BAD_ARG;
  hipLaunchKernelGGL(knn_backward_kernel, dim3((unsigned)nl_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, xyz, sp_xyz, idx, g_d2, (int)N, K, (int)M,
                     g_xyz, g_sp_xyz);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

}  // extern "C"
