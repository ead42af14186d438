// Backward kernels, first slice: the two reductions of the gradient path that are not GEMM-shaped.
//   composite_backward_kernel  gradient of front-to-back alpha compositing (conditional_nerf/model.py:544-560,597: weights, rgb, depth,
//                              depth uncertainty, composited feature) w.r.t. the density, the per-sample colours and the per-sample feature
//                              rows — one wave per ray, the forward's transmittance is recomputed (nothing is saved by the forward pass)
//   knn_backward_kernel        KNearestNeighborBackwardKernel (ops/knn/src/knn.cu:449-490): d(squared distance)/d(query, support point)
// Both are called from nerf_loc_amd/diff_render.py (autograd.Function) when the gradient path runs on the GPU, and are checked against
// PyTorch autograd of the same expressions and, end to end, against the reference's autograd goldens (tests/test_backward_kernels.py,
// tests/test_diff_render.py).
#include <stdlib.h>
#include "common.h"

namespace {

__device__ __forceinline__ float bk_rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int bk_rli(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

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

// ---- glue of the whole-path backward (abi.hip: do_render_backward) -----------------------------------------------------------------------
// hc[r][c] = sum_s w_s ft[r][s][c], wsum4[r] = (sum_s w_s, 0, 0, 0): the composited hidden rows feat_mlp.2's weight gradient multiplies
template <int CH>
__global__ __launch_bounds__(256) void ray_feat_sum_kernel(const float* __restrict__ z_vals, const float* __restrict__ sigma, const float* __restrict__ ft, int R,
                                                           int S, int C, float* __restrict__ hc, float* __restrict__ wsum4) {
  __shared__ float wsh[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wv;
  if (r >= R) return;
  const float* z = z_vals + (size_t)r * S;
  float al[CH];
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) {
      const float dl = (s + 1 < S) ? z[s + 1] - z[s] : 1e2f;
      al[j] = 1.f - expf(-dl * sigma[(size_t)r * S + s]);
      prod *= (1.f - al[j]);
    } else al[j] = 0.f;
  }
  float inc = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(inc, o, 64); if (lane >= o) inc *= t; }
  float T = __shfl_up(inc, 1, 64);
  if (lane == 0) T = 1.f;
  float ws = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) { const float w = al[j] * T; T *= (1.f - al[j]); ws += w; wsh[wv][s] = w; }
  }
  ws = wave_sum(ws);
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < C; c += 64) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a = fmaf(wsh[wv][s], ft[((size_t)r * S + s) * C + c], a);
    hc[(size_t)r * C + c] = a;
  }
  if (lane == 0) *(float4*)(wsum4 + 4 * (size_t)r) = make_float4(ws, 0.f, 0.f, 0.f);
}
// sigma = softplus(geo . w + b) backwards: g_geo (N, W) = g_pre w, gpre4 (N, 4) = (g_pre, 0, 0, 0), g_pre = g_sigma * sigmoid(pre); one wave per sample
__global__ __launch_bounds__(256) void sigma_backward_kernel(const float* __restrict__ geo, int N, int W, const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ g_sigma, float* __restrict__ g_geo, float* __restrict__ gpre4) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float a = 0.f;
  for (int c = lane; c < W; c += 64) a = fmaf(geo[(size_t)n * W + c], w[c], a);
  const float pre = wave_sum(a) + b[0];
  const float gp = g_sigma[n] * (pre > 20.f ? 1.f : nl_sigmoid(pre));
  for (int c = lane; c < W; c += 64) g_geo[(size_t)n * W + c] = gp * w[c];
  if (lane == 0) *(float4*)(gpre4 + 4 * (size_t)n) = make_float4(gp, 0.f, 0.f, 0.f);
}
// the weights' total cotangent: gw[r][s] = g_wts[r][s] + g_feat[r] . b2   (feat = W2 . sum_s w_s hidden_s + b2 sum_s w_s); one wave per ray
__global__ __launch_bounds__(256) void gw_total_kernel(const float* __restrict__ g_wts, const float* __restrict__ g_feat, const float* __restrict__ b2, int R, int S,
                                                       int C, float* __restrict__ gw, const float* __restrict__ g_beta, const float* __restrict__ bv) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float a = 0.f;
  if (g_feat) for (int c = lane; c < C; c += 64) a = fmaf(g_feat[(size_t)r * C + c], b2[c], a);
  a = wave_sum(a);
  const float gb = g_beta ? g_beta[r] : 0.f;   // + the uncertainty head: beta = sum_s w_s b_s + beta_min
  for (int s = lane; s < S; s += 64) gw[(size_t)r * S + s] = a + (g_wts ? g_wts[(size_t)r * S + s] : 0.f) + (g_beta ? gb * bv[(size_t)r * S + s] : 0.f);
}
// training-mode uncertainty head (model.py:587-592): beta[r] = sum_s w_s softplus(geo_s . wb + bb) + beta_min.  Forward from the kept weights / head values:
__global__ __launch_bounds__(256) void beta_forward_kernel(const float* __restrict__ wts, const float* __restrict__ bv, int R, int S, float beta_min, float* __restrict__ beta) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float a = 0.f;
  for (int s = lane; s < S; s += 64) a = fmaf(wts[(size_t)r * S + s], bv[(size_t)r * S + s], a);
  a = wave_sum(a);
  if (lane == 0) beta[r] = a + beta_min;
}
// ... and backwards (the weights' cotangent g_beta b_s is added by gw_total_kernel, before compositing is differentiated): the head's pre-activation
// gradient g_beta w_s sigmoid(pre) goes to g_geo (+=) and to gpre4
__global__ __launch_bounds__(256) void beta_backward_kernel(const float* __restrict__ geo, int N, int S, int W, const float* __restrict__ wb, const float* __restrict__ bb,
                                                            const float* __restrict__ wts, const float* __restrict__ g_beta,
                                                            float* __restrict__ g_geo, float* __restrict__ gpre4) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float gb = g_beta[n / S];
  float a = 0.f;
  for (int c = lane; c < W; c += 64) a = fmaf(geo[(size_t)n * W + c], wb[c], a);
  const float pre = wave_sum(a) + bb[0];
  const float gp = gb * wts[n] * (pre > 20.f ? 1.f : nl_sigmoid(pre));
  for (int c = lane; c < W; c += 64) g_geo[(size_t)n * W + c] += gp * wb[c];
  if (lane == 0) *(float4*)(gpre4 + 4 * (size_t)n) = make_float4(gp, 0.f, 0.f, 0.f);
}

// xyz = o + z d: g_o[r] = sum_s g_xyz, g_d[r] = sum_s (z_s g_xyz + g_dir), g_qc[r] = sum_s g_qcN; g_xyz = ga + gb (+ gc); one wave per ray
__global__ __launch_bounds__(256) void ray_reduce_kernel(const float* __restrict__ ga, const float* __restrict__ gb, const float* __restrict__ gc,
                                                         const float* __restrict__ g_dir, const float* __restrict__ g_qcN, const float* __restrict__ z, int R, int S,
                                                         float* __restrict__ g_o, float* __restrict__ g_d, float* __restrict__ g_qc) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f}, q[3] = {0.f, 0.f, 0.f};
  for (int s = lane; s < S; s += 64) {
    const size_t n = (size_t)r * S + s;
    const float zz = z[n];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float g = ga[3 * n + k] + (gb ? gb[3 * n + k] : 0.f) + (gc ? gc[3 * n + k] : 0.f);
      o[k] += g;
      d[k] += zz * g + (g_dir ? g_dir[3 * n + k] : 0.f);
      if (g_qcN) q[k] += g_qcN[3 * n + k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = wave_sum(o[k]); d[k] = wave_sum(d[k]); q[k] = wave_sum(q[k]); }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { g_o[3 * (size_t)r + k] = o[k]; g_d[3 * (size_t)r + k] = d[k]; if (g_qc) g_qc[3 * (size_t)r + k] = q[k]; }
  }
}

}  // namespace

int nl_launch_ray_feat_sum(const float* z, const float* sigma, const float* ft, int64_t R, int S, int C, float* hc, float* wsum4, hipStream_t st) {
  if (R <= 0) return NL_OK;
  dim3 grid((unsigned)nl_cdiv(R, 4));
#define NL_RF(CH) hipLaunchKernelGGL(ray_feat_sum_kernel<CH>, grid, dim3(256), 0, st, z, sigma, ft, (int)R, S, C, hc, wsum4)
  if (S <= 64) NL_RF(1); else if (S <= 128) NL_RF(2); else if (S <= 192) NL_RF(3); else NL_RF(4);
#undef NL_RF
  NL_LAUNCH_CHECK();
  return NL_OK;
}
int nl_launch_sigma_backward(const float* geo, int64_t N, int W, const float* w, const float* b, const float* g_sigma, float* g_geo, float* gpre4, hipStream_t st) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(sigma_backward_kernel, dim3((unsigned)nl_cdiv(N, 4)), dim3(256), 0, st, geo, (int)N, W, w, b, g_sigma, g_geo, gpre4);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
int nl_launch_beta_forward(const float* wts, const float* bv, int64_t R, int S, float beta_min, float* beta, hipStream_t st) {
  if (R <= 0) return NL_OK;
  hipLaunchKernelGGL(beta_forward_kernel, dim3((unsigned)nl_cdiv(R, 4)), dim3(256), 0, st, wts, bv, (int)R, S, beta_min, beta);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
int nl_launch_beta_backward(const float* geo, int64_t N, int S, int W, const float* wb, const float* bb, const float* wts, const float* g_beta, float* g_geo, float* gpre4,
                            hipStream_t st) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(beta_backward_kernel, dim3((unsigned)nl_cdiv(N, 4)), dim3(256), 0, st, geo, (int)N, S, W, wb, bb, wts, g_beta, g_geo, gpre4);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
int nl_launch_gw_total(const float* g_wts, const float* g_feat, const float* b2, int64_t R, int S, int C, float* gw, const float* g_beta, const float* bv,
                       hipStream_t st) {
  if (R <= 0) return NL_OK;
  hipLaunchKernelGGL(gw_total_kernel, dim3((unsigned)nl_cdiv(R, 4)), dim3(256), 0, st, g_wts, g_feat, b2, (int)R, S, C, gw, g_beta, bv);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
int nl_launch_ray_reduce(const float* ga, const float* gb, const float* gc, const float* g_dir, const float* g_qcN, const float* z, int64_t R, int S, float* g_o,
                         float* g_d, float* g_qc, hipStream_t st) {
  if (R <= 0) return NL_OK;
  hipLaunchKernelGGL(ray_reduce_kernel, dim3((unsigned)nl_cdiv(R, 4)), dim3(256), 0, st, ga, gb, gc, g_dir, g_qcN, z, (int)R, S, g_o, g_d, g_qc);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

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

// =====================================================================================================================
// Neural-point branch (rows a9-a12), gradient w.r.t. its INPUTS with frozen weights (what PoseOptimizer needs, pose_optimizer.py:131-168,
// and the input-gradient half of a training step): the glue kernels between the transposed-weight GEMMs (abi.hip: do_point_backward).
// A Linear layer's input gradient needs no activations, LeakyReLU's needs the sign of its output, the attention needs q / k / v — so the
// backward pass re-runs the staged forward (point_encode -> GEMMs -> attention, the kernels of point.hip) into its workspace and walks back.
namespace {

// d/dx of y = (LayerNorm(x; eps) * gamma + beta) * sc with x = FC + G (ibrnet.py:117 + the aggregation scale): one wave per sample
template <int WPL>
__global__ __launch_bounds__(256) void ln_agg_backward_kernel(const float* __restrict__ FC, const float* __restrict__ G, const float* __restrict__ gy, int N, int W,
                                                              const float* __restrict__ gamma, float eps, const float* __restrict__ wscale,
                                                              float* __restrict__ gx, float* __restrict__ aff /* (N, 2W) [gy sc xhat | gy sc] or null */) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float x[WPL], g[WPL];
  float s = 0.f;
  const float sc = wscale ? wscale[n] : 1.f;
#pragma unroll
  for (int j = 0; j < WPL; ++j) {
    const int c = lane + 64 * j;
    x[j] = c < W ? FC[(size_t)n * W + c] + G[(size_t)n * W + c] : 0.f;
    g[j] = c < W ? gy[(size_t)n * W + c] * gamma[c] * sc : 0.f;
    s += x[j];
  }
  const float mean = wave_sum(s) / (float)W;
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < WPL; ++j) { const int c = lane + 64 * j; const float d = c < W ? x[j] - mean : 0.f; v += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(v) / (float)W + eps);
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int j = 0; j < WPL; ++j) { const int c = lane + 64 * j; if (c < W) { x[j] = (x[j] - mean) * rstd; a += g[j]; b += g[j] * x[j]; } }
  const float m1 = wave_sum(a) / (float)W, m2 = wave_sum(b) / (float)W;
#pragma unroll
  for (int j = 0; j < WPL; ++j) {
    const int c = lane + 64 * j;
    if (c < W) gx[(size_t)n * W + c] = rstd * (g[j] - m1 - x[j] * m2);
  }
  if (aff) {   // training: the rows whose column sums are d/d gamma and d/d beta
#pragma unroll
    for (int j = 0; j < WPL; ++j) {
      const int c = lane + 64 * j;
      if (c < W) { const float gs = gy[(size_t)n * W + c] * sc; aff[(size_t)n * 2 * W + c] = gs * x[j]; aff[(size_t)n * 2 * W + W + c] = gs; }
    }
  }
}

// backward of attn_kernel (point.hip; ibrnet.py:89-108): one wave per sample, lane owns dims {2 lane, 2 lane + 1}, head = lane / 16
__global__ __launch_bounds__(256) void attn_backward_kernel(const float* __restrict__ Q, const float* __restrict__ KV, const float* __restrict__ gO, int N, int K,
                                                            float* __restrict__ gQ, float* __restrict__ gKV) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float temp = 5.656854249492381f;
  const float2 q = *(const float2*)(Q + (size_t)n * 128 + 2 * lane);
  const float q0 = q.x / temp, q1 = q.y / temp;
  const float2 go = *(const float2*)(gO + (size_t)n * 128 + 2 * lane);
  auto head_sum = [](float p) { p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 8, 64); return p; };
  float sc[NL_KNN_MAX_K], gp[NL_KNN_MAX_K];
  float2 kk[NL_KNN_MAX_K];
  float mx = -3.4e38f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) {
    sc[k] = -3.4e38f; gp[k] = 0.f; kk[k] = make_float2(0.f, 0.f);
    if (k < K) {
      const float* r = KV + ((size_t)n * K + k) * 256;
      kk[k] = *(const float2*)(r + 2 * lane);
      const float2 vv = *(const float2*)(r + 128 + 2 * lane);
      sc[k] = head_sum(q0 * kk[k].x + q1 * kk[k].y);
      gp[k] = head_sum(go.x * vv.x + go.y * vv.y);     // dL/d(attention weight k) of this head
      mx = fmaxf(mx, sc[k]);
    }
  }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) { sc[k] = k < K ? expf(sc[k] - mx) : 0.f; den += sc[k]; }
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) { sc[k] /= den; dot += sc[k] * gp[k]; }
  float gq0 = 0.f, gq1 = 0.f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) {
    if (k < K) {
      const float gs = sc[k] * (gp[k] - dot);          // softmax backward -> dL/d(score k)
      float* r = gKV + ((size_t)n * K + k) * 256;
      *(float2*)(r + 2 * lane) = make_float2(gs * q0, gs * q1);                 // d score / d k = q / temp
      *(float2*)(r + 128 + 2 * lane) = make_float2(sc[k] * go.x, sc[k] * go.y);  // d out / d v = attention weight
      gq0 += gs * kk[k].x; gq1 += gs * kk[k].y;
    }
  }
  *(float2*)(gQ + (size_t)n * 128 + 2 * lane) = make_float2(gq0 / temp, gq1 / temp);
}

// g *= LeakyReLU'(pre-activation): the sign of the layer's OUTPUT h is the sign of its input (slope 0.01 > 0)
__global__ void lrelu_mask_kernel(float4* __restrict__ g, const float4* __restrict__ h, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = g[i];
  const float4 b = h[i];
  a.x *= b.x > 0.f ? 1.f : 0.01f; a.y *= b.y > 0.f ? 1.f : 0.01f; a.z *= b.z > 0.f ? 1.f : 0.01f; a.w *= b.w > 0.f ? 1.f : 0.01f;
  g[i] = a;
}

// out (N, W) = a + b
__global__ void add_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 x = a[i], y = b[i];
  o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

// backward of point_encode_kernel's positional encoding + ray_diff_fc columns (point.hip; utils.py:5-35, model.py:396-399, 36-39):
// gX (N*K, ldg): [0, 63) posenc columns, [63, 90) ray_diff_fc columns  ->  g_xyz (N, 3), g_dir (N, 3) (per SAMPLE; null when the
// direction was the nearest neighbour's)
__global__ __launch_bounds__(256) void point_encode_backward_kernel(const float* __restrict__ xyz, const float* __restrict__ dir, int dir_stride, int dir_div,
                                                                    int N, int K, int M, const int* __restrict__ idx, const float* __restrict__ sp_xyz,
                                                                    const float* __restrict__ sp_dir, const float* __restrict__ rd_w, float inv_span,
                                                                    const float* __restrict__ gX, int ldg, float* __restrict__ g_xyz,
                                                                    float* __restrict__ g_dir, float* __restrict__ tr /* training: (N*K, 68) or null */) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float qx = xyz[3 * (size_t)n], qy = xyz[3 * (size_t)n + 1], qz = xyz[3 * (size_t)n + 2];
  float dx = 0.f, dy = 0.f, dz = 0.f;
  if (dir) {
    const size_t dr = (size_t)(n / dir_div) * dir_stride;
    dx = dir[dr]; dy = dir[dr + 1]; dz = dir[dr + 2];
  } else if (M > 0) {
    const int i0 = idx[(size_t)n * K];
    dx = sp_dir[4 * (size_t)i0]; dy = sp_dir[4 * (size_t)i0 + 1]; dz = sp_dir[4 * (size_t)i0 + 2];
  }
  float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gd0 = 0.f, gd1 = 0.f, gd2 = 0.f;   // per-lane partial sums over the neighbours; reduced once at the end
  const float* w2 = rd_w + 80;
  for (int k = 0; k < K; ++k) {
    const bool have = k < M;
    const int i = idx[(size_t)n * K + k];
    const float* grow = gX + ((size_t)n * K + k) * ldg;
    const float nx = have ? sp_xyz[3 * (size_t)i] : 0.f, ny = have ? sp_xyz[3 * (size_t)i + 1] : 0.f, nz = have ? sp_xyz[3 * (size_t)i + 2] : 0.f;
    const float off[3] = {(qx - nx) * inv_span, (qy - ny) * inv_span, (qz - nz) * inv_span};
    // ---- positional encoding: column j of lane j < 63
    if (lane < 63) {
      const float g = grow[lane];
      int ax;
      float dv;
      if (lane < 3) { ax = lane; dv = 1.f; }
      else {
        const int j = lane - 3, f = j / 6, r = j - 6 * f;
        ax = r >= 3 ? r - 3 : r;
        const float o = ax == 0 ? off[0] : (ax == 1 ? off[1] : off[2]);
        const float sc2 = (float)(1 << f), arg = o * sc2;
        dv = r < 3 ? cosf(arg) * sc2 : -sinf(arg) * sc2;
      }
      const float t = g * dv;
      gx0 += ax == 0 ? t : 0.f; gx1 += ax == 1 ? t : 0.f; gx2 += ax == 2 ? t : 0.f;
    }
    // ---- ray_diff_fc (4 -> 16 -> 27, LeakyReLU after both): lane j < 16 owns hidden unit j, lane l < 27 output l; values cross lanes by readlane
    if (dir || tr) {   // (wave-uniform)
      const float ndx = have ? sp_dir[4 * (size_t)i] : 0.f, ndy = have ? sp_dir[4 * (size_t)i + 1] : 0.f, ndz = have ? sp_dir[4 * (size_t)i + 2] : 0.f;
      const float rr0 = dx - ndx, rr1 = dy - ndy, rr2 = dz - ndz;
      const float nrm = sqrtf(rr0 * rr0 + rr1 * rr1 + rr2 * rr2), nr = nrm + 1e-8f;
      const float r0 = rr0 / nr, r1 = rr1 / nr, r2 = rr2 / nr, r3 = dx * ndx + dy * ndy + dz * ndz;
      const int ju = lane < 16 ? lane : 0, lu = lane < 27 ? lane : 0;
      float a1 = rd_w[64 + ju];
      a1 = fmaf(rd_w[ju * 4 + 0], r0, a1); a1 = fmaf(rd_w[ju * 4 + 1], r1, a1); a1 = fmaf(rd_w[ju * 4 + 2], r2, a1); a1 = fmaf(rd_w[ju * 4 + 3], r3, a1);
      const float hj = nl_lrelu(a1);
      float a2 = w2[27 * 16 + lu];
#pragma unroll
      for (int j = 0; j < 16; ++j) a2 = fmaf(w2[lu * 16 + j], bk_rl(hj, j), a2);
      const float ga2 = lane < 27 ? grow[63 + lane] * (a2 > 0.f ? 1.f : 0.01f) : 0.f;
      float gh = 0.f;
#pragma unroll
      for (int l = 0; l < 27; ++l) gh = fmaf(w2[l * 16 + ju], bk_rl(ga2, l), gh);
      const float ga1 = lane < 16 ? gh * (a1 > 0.f ? 1.f : 0.01f) : 0.f;
      const float gr0 = wave_sum(rd_w[ju * 4 + 0] * ga1), gr1 = wave_sum(rd_w[ju * 4 + 1] * ga1), gr2 = wave_sum(rd_w[ju * 4 + 2] * ga1),
                  gr3 = wave_sum(rd_w[ju * 4 + 3] * ga1);
      // u = rr / (|rr| + 1e-8): du_i/drr_j = delta_ij / nr - rr_i rr_j / (|rr| nr^2)  (0 at rr = 0, like torch.norm's subgradient)
      const float gdot = gr0 * rr0 + gr1 * rr1 + gr2 * rr2;
      const float cc = nrm > 0.f ? gdot / (nrm * nr * nr) : 0.f;
      gd0 += gr0 / nr - rr0 * cc + gr3 * ndx;   // (the same value in every lane: not reduced)
      gd1 += gr1 / nr - rr1 * cc + gr3 * ndy;
      gd2 += gr2 / nr - rr2 * cc + gr3 * ndz;
      if (tr) {   // rows for ray_diff_fc's weight gradients: [input 4 | hidden 16 | d hidden 16 | d output 32 (27 used)]
        float* t = tr + ((size_t)n * K + k) * 68;
        if (lane < 4) t[lane] = lane == 0 ? r0 : lane == 1 ? r1 : lane == 2 ? r2 : r3;
        if (lane < 16) { t[4 + lane] = hj; t[20 + lane] = ga1; }
        if (lane < 32) t[36 + lane] = ga2;
      }
    }
  }
  gx0 = wave_sum(gx0); gx1 = wave_sum(gx1); gx2 = wave_sum(gx2);
  if (lane == 0) {
    g_xyz[3 * (size_t)n] = gx0 * inv_span; g_xyz[3 * (size_t)n + 1] = gx1 * inv_span; g_xyz[3 * (size_t)n + 2] = gx2 * inv_span;
    if (g_dir) { g_dir[3 * (size_t)n] = gd0; g_dir[3 * (size_t)n + 1] = gd1; g_dir[3 * (size_t)n + 2] = gd2; }
  }
}

// The same for K = 8 with LANE = one (sample, neighbour) row: the row's 90 gradient columns, its positional-encoding derivative (one fp64 sin / cos per
// axis + the double-angle recurrence, like the forward kernel) and the whole ray_diff_fc 4 -> 16 -> 27 forward + backward (~1000 FMAs against wave-uniform weights, which
// the compiler keeps in SGPRs) stay in the lane; the 8 rows of a sample are summed with three DPP steps.  The wave-per-sample kernel above walks the neighbours one after
// the other with five wave reductions each: 0.44 ms per 512-ray pose step against 0.06 ms here.
__device__ __forceinline__ void bk_sincos_d(double x, double& s, double& c) {   // Cody-Waite to [-pi/4, pi/4] + Taylor (error < 1e-11), branch-free
  const double kd = rint(x * 0.63661977236758134308);
  const int k = (int)kd;
  double r = fma(-kd, 1.5707963267948966, x);
  r = fma(-kd, 6.123233995736766e-17, r);
  const double r2 = r * r;
  const double ps = r + r * r2 * (-1.0 / 6 + r2 * (1.0 / 120 + r2 * (-1.0 / 5040 + r2 * (1.0 / 362880 + r2 * (-1.0 / 39916800)))));
  const double pc = 1.0 + r2 * (-0.5 + r2 * (1.0 / 24 + r2 * (-1.0 / 720 + r2 * (1.0 / 40320 + r2 * (-1.0 / 3628800 + r2 * (1.0 / 479001600))))));
  const bool sw = k & 1;
  const double ss = sw ? pc : ps, cc = sw ? ps : pc;
  s = (k & 2) ? -ss : ss;
  c = ((k + 1) & 2) ? -cc : cc;
}
__global__ __launch_bounds__(256) void point_encode_backward_rows_kernel(const float* __restrict__ xyz, const float* __restrict__ dir, int dir_stride, int dir_div,
                                                                         int N, int M, const int* __restrict__ idx, const float* __restrict__ sp_xyz,
                                                                         const float* __restrict__ sp_dir, const float* __restrict__ rd_w, float inv_span,
                                                                         const float* __restrict__ gX, int ldg, float* __restrict__ g_xyz, float* __restrict__ g_dir,
                                                                         float* __restrict__ tr /* training: (N*8, 68) rows for ray_diff_fc's weight gradients, or null */) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k = (int)(row & 7);
  const long long nl = row >> 3;
  const bool live = nl < N;
  const int n = (int)(live ? nl : N - 1);
  const bool have = k < M;
  const int i = idx[(size_t)n * 8 + k];
  const float* grow = gX + ((size_t)n * 8 + k) * ldg;
  float g[92];
#pragma unroll
  for (int c = 0; c < 23; ++c) { const float4 v = *(const float4*)(grow + 4 * c); g[4 * c] = v.x; g[4 * c + 1] = v.y; g[4 * c + 2] = v.z; g[4 * c + 3] = v.w; }
  const float q[3] = {xyz[3 * (size_t)n], xyz[3 * (size_t)n + 1], xyz[3 * (size_t)n + 2]};
  float go[3];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float off = (q[ax] - (have ? sp_xyz[3 * (size_t)i + ax] : 0.f)) * inv_span;
    double sn, cs;
    bk_sincos_d((double)off, sn, cs);
    double a = (double)g[ax];
    double sc2 = 1.0;
#pragma unroll
    for (int f = 0; f < 10; ++f) {   // columns 3 + 6 f + ax = sin(2^f off), 3 + 6 f + 3 + ax = cos(2^f off)  (utils.py:5-35)
      a += sc2 * ((double)g[3 + 6 * f + ax] * cs - (double)g[6 + 6 * f + ax] * sn);
      const double s2 = 2.0 * sn * cs;
      cs = fma(-2.0 * sn, sn, 1.0);
      sn = s2;
      sc2 *= 2.0;
    }
    go[ax] = (float)a;
  }
  float gd[3] = {0.f, 0.f, 0.f};
  if (dir || tr) {   // (kernel-uniform) ray_diff_fc (model.py:36-39, 396-399): 4 -> 16 -> 27, LeakyReLU after both
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (dir) { const size_t dr = (size_t)(n / dir_div) * dir_stride; dx = dir[dr]; dy = dir[dr + 1]; dz = dir[dr + 2]; }
    else if (M > 0) { const int i0 = idx[(size_t)n * 8]; dx = sp_dir[4 * (size_t)i0]; dy = sp_dir[4 * (size_t)i0 + 1]; dz = sp_dir[4 * (size_t)i0 + 2]; }   // model.py:391-392
    const float ndx = have ? sp_dir[4 * (size_t)i] : 0.f, ndy = have ? sp_dir[4 * (size_t)i + 1] : 0.f, ndz = have ? sp_dir[4 * (size_t)i + 2] : 0.f;
    const float rr0 = dx - ndx, rr1 = dy - ndy, rr2 = dz - ndz;
    const float nrm = sqrtf(rr0 * rr0 + rr1 * rr1 + rr2 * rr2), nr = nrm + 1e-8f;
    const float r0 = rr0 / nr, r1 = rr1 / nr, r2 = rr2 / nr, r3 = dx * ndx + dy * ndy + dz * ndz;
    const float* w2 = rd_w + 80;
    float a1[16], h[16], gh[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float t = rd_w[64 + j];
      t = fmaf(rd_w[j * 4 + 0], r0, t); t = fmaf(rd_w[j * 4 + 1], r1, t); t = fmaf(rd_w[j * 4 + 2], r2, t); t = fmaf(rd_w[j * 4 + 3], r3, t);
      a1[j] = t; h[j] = nl_lrelu(t); gh[j] = 0.f;
    }
#pragma unroll
    for (int l = 0; l < 27; ++l) {
      float a2 = w2[27 * 16 + l];
#pragma unroll
      for (int j = 0; j < 16; ++j) a2 = fmaf(w2[l * 16 + j], h[j], a2);
      const float ga2 = g[63 + l] * (a2 > 0.f ? 1.f : 0.01f);
      g[63 + l] = ga2;   // (kept for the training rows)
#pragma unroll
      for (int j = 0; j < 16; ++j) gh[j] = fmaf(w2[l * 16 + j], ga2, gh[j]);
    }
    float gr0 = 0.f, gr1 = 0.f, gr2 = 0.f, gr3 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float ga1 = gh[j] * (a1[j] > 0.f ? 1.f : 0.01f);
      gh[j] = ga1;
      gr0 = fmaf(rd_w[j * 4 + 0], ga1, gr0); gr1 = fmaf(rd_w[j * 4 + 1], ga1, gr1); gr2 = fmaf(rd_w[j * 4 + 2], ga1, gr2); gr3 = fmaf(rd_w[j * 4 + 3], ga1, gr3);
    }
    // u = rr / (|rr| + 1e-8): du_i/drr_j = delta_ij / nr - rr_i rr_j / (|rr| nr^2)  (0 at rr = 0, like torch.norm's subgradient)
    const float gdot = gr0 * rr0 + gr1 * rr1 + gr2 * rr2;
    const float cc = nrm > 0.f ? gdot / (nrm * nr * nr) : 0.f;
    gd[0] = gr0 / nr - rr0 * cc + gr3 * ndx;
    gd[1] = gr1 / nr - rr1 * cc + gr3 * ndy;
    gd[2] = gr2 / nr - rr2 * cc + gr3 * ndz;
    if (tr && live) {   // rows for ray_diff_fc's weight gradients: [input 4 | hidden 16 | d hidden 16 | d output 32 (27 used)]
      float4* t = (float4*)(tr + ((size_t)n * 8 + k) * 68);
      t[0] = make_float4(r0, r1, r2, r3);
#pragma unroll
      for (int c = 0; c < 4; ++c) { t[1 + c] = make_float4(h[4 * c], h[4 * c + 1], h[4 * c + 2], h[4 * c + 3]); t[5 + c] = make_float4(gh[4 * c], gh[4 * c + 1], gh[4 * c + 2], gh[4 * c + 3]); }
#pragma unroll
      for (int c = 0; c < 6; ++c) t[9 + c] = make_float4(g[63 + 4 * c], g[64 + 4 * c], g[65 + 4 * c], g[66 + 4 * c]);
      t[15] = make_float4(g[87], g[88], g[89], 0.f);
      t[16] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // the sample's 8 rows are 8 neighbouring lanes
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    float a = go[ax], b = gd[ax];
    a = nl_sum8(a); b = nl_sum8(b);
    go[ax] = a; gd[ax] = b;
  }
  if (live && k == 0) {
    g_xyz[3 * (size_t)n] = go[0] * inv_span; g_xyz[3 * (size_t)n + 1] = go[1] * inv_span; g_xyz[3 * (size_t)n + 2] = go[2] * inv_span;
    if (g_dir) { g_dir[3 * (size_t)n] = gd[0]; g_dir[3 * (size_t)n + 1] = gd[1]; g_dir[3 * (size_t)n + 2] = gd[2]; }
  }
}

// training: gradient of the gathered support features (knn_gather's backward, knn_utils.py:211-233 = index_add).  A wave takes 64 consecutive (sample,
// neighbour) rows = 8 consecutive samples of a ray, whose neighbour sets overlap almost completely: the rows are grouped by support point with ballots
// (wave-uniform), every group's rows are summed with lanes = channels, and ONE atomic per (point, channel) goes out — ~14 points per 64 rows instead of 64
// rows, and no two waves of a ray hammering the same line back to back (0.81 -> 0.35 ms per training step).
__global__ __launch_bounds__(256) void sp_feat_scatter_kernel(const float* __restrict__ gXF, int ld, int F, const int* __restrict__ idx, long long NK, int K, int M,
                                                              float* __restrict__ g_sp_feat) {
  const int lane = threadIdx.x & 63;
  const long long base = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (base >= NK) return;
  const long long row = base + lane;
  const int key = (row < NK && (int)(row % K) < M) ? idx[row] : -1;   // (columns >= M are zero-filled neighbours)
  unsigned long long todo = __ballot(key >= 0);
  while (todo) {
    const int k0 = __builtin_amdgcn_readlane(key, __ffsll((long long)todo) - 1);
    const unsigned long long grp = __ballot(key == k0);
    todo &= ~grp;
    for (int c0 = 0; c0 < F; c0 += 64) {
      const int c = c0 + lane;
      const bool on = c < F;
      const float* src = gXF + (size_t)base * ld + (on ? c : 0);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      unsigned long long g = grp;
      while (g) {   // four rows in flight
        const int r0 = __ffsll((long long)g) - 1; g &= g - 1;
        const int r1 = g ? __ffsll((long long)g) - 1 : -1; g &= g - (g != 0);
        const int r2 = g ? __ffsll((long long)g) - 1 : -1; g &= g - (g != 0);
        const int r3 = g ? __ffsll((long long)g) - 1 : -1; g &= g - (g != 0);
        const float v0 = on ? src[(size_t)r0 * ld] : 0.f;
        const float v1 = on && r1 >= 0 ? src[(size_t)r1 * ld] : 0.f;
        const float v2 = on && r2 >= 0 ? src[(size_t)r2 * ld] : 0.f;
        const float v3 = on && r3 >= 0 ? src[(size_t)r3 * ld] : 0.f;
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
      }
      if (on) atomicAdd(g_sp_feat + (size_t)k0 * F + c, (a0 + a1) + (a2 + a3));
    }
  }
}

// column sums of a (rows, M) matrix, coalesced: block = 64 columns x 4 row phases over one slab of rows; part[slab][M]
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ Y, int ldy, long long rows, int M, long long slab, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * slab, r1 = min(rows, r0 + slab);
  float s = 0.f;
  if (c < M)
    for (long long r = r0 + ph; r < r1; r += 4) s += Y[(size_t)r * ldy + c];
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && c < M) part[(size_t)blockIdx.y * M + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nslab, int M, float* __restrict__ out) {
  // 32 columns per block, 8 threads each over interleaved slabs, combined in a fixed order (one thread walking 256 slabs took 30 us)
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, zz = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + el;
  float s = 0.f;
  if (c < M)
    for (int z = zz; z < nslab; z += 8) s += part[(size_t)z * M + c];
  red[zz][el] = s;
  __syncthreads();
  if (zz == 0 && c < M) out[c] += ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
}

// the same final pass for the U-Net's LayerNorm tables: columns [0, n) / [n, 2 n) are the (L, Cc) position-major sums of d gamma / d beta; the state_dict's
// tables are channel-major: gw[c][l] += column l Cc + c
__global__ __launch_bounds__(256) void colsum_final_tables_kernel(const float* __restrict__ part, int nslab, int L, int Cc, float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, zz = threadIdx.x >> 5;
  const int n = L * Cc, c = blockIdx.x * 32 + el;
  float s = 0.f;
  if (c < 2 * n)
    for (int z = zz; z < nslab; z += 8) s += part[(size_t)z * 2 * n + c];
  red[zz][el] = s;
  __syncthreads();
  if (zz != 0 || c >= 2 * n) return;
  const float v = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
  const int e = c < n ? c : c - n;
  float* g = c < n ? gw : gb;
  if (g) { const int l = e / Cc, ch = e - l * Cc; g[(size_t)ch * L + l] += v; }
}

}  // namespace

// gw (Cc, L) / gb (Cc, L) += the column sums of Y (rows, 2 L Cc) [d gamma | d beta], position-major; scratch: 256 * 2 L Cc floats.  Either table may be null.
int nl_launch_colsum_tables(const float* Y, int64_t rows, int L, int Cc, float* gw, float* gb, float* scratch, hipStream_t st) {
  if (rows <= 0 || L <= 0 || Cc <= 0) return NL_OK;
  const int M = 2 * L * Cc;
  int nslab = (int)(nl_cdiv(rows, 64) < 256 ? nl_cdiv(rows, 64) : 256);
  const long long slab = nl_cdiv(rows, nslab);
  nslab = (int)nl_cdiv(rows, slab);
  hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)nl_cdiv(M, 64), (unsigned)nslab), dim3(256), 0, st, Y, M, (long long)rows, M, slab, scratch);
  NL_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_tables_kernel, dim3((unsigned)nl_cdiv(M, 32)), dim3(256), 0, st, scratch, nslab, L, Cc, gw, gb);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

// out (M) += column sums of Y (rows, M; ld ldy); scratch: 256 * M floats
int nl_launch_colsum(const float* Y, int ldy, int64_t rows, int M, float* out, float* scratch, hipStream_t st) {
  if (rows <= 0 || M <= 0) return NL_OK;
  int nslab = (int)(nl_cdiv(rows, 64) < 256 ? nl_cdiv(rows, 64) : 256);
  const long long slab = nl_cdiv(rows, nslab);
  nslab = (int)nl_cdiv(rows, slab);
  hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)nl_cdiv(M, 64), (unsigned)nslab), dim3(256), 0, st, Y, ldy, (long long)rows, M, slab, scratch);
  NL_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)nl_cdiv(M, 32)), dim3(256), 0, st, scratch, nslab, M, out);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

// dst[r][c] (+)= src[r][c] for c < cols: rows whose leading dimensions are not multiples of 4 floats
namespace {
__global__ void copy_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, long long rows, int cols, int add) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long long r = i / cols; const int c = (int)(i - r * cols);
  const float v = src[(size_t)r * lds + c];
  float* d = dst + (size_t)r * ldd + c;
  *d = add ? *d + v : v;
}
}  // namespace
int nl_launch_copy_rows(const float* src, int lds, float* dst, int ldd, int64_t rows, int cols, bool add, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return NL_OK;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)nl_cdiv(rows * cols, 256)), dim3(256), 0, st, src, lds, dst, ldd, (long long)rows, cols, add ? 1 : 0);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_sp_feat_scatter(const float* gXF, int ld, int F, const int* idx, int64_t N, int K, int64_t M, float* g_sp_feat, hipStream_t st) {
  if (N <= 0 || M <= 0) return NL_OK;
  hipLaunchKernelGGL(sp_feat_scatter_kernel, dim3((unsigned)nl_cdiv(N * K, 256)), dim3(256), 0, st, gXF, ld, F, idx, (long long)(N * K), K,
                     (int)(M > 0x7fffffff ? 0x7fffffff : M), g_sp_feat);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_ln_agg_backward(const float* FC, const float* G, const float* gy, int64_t N, int W, const float* gamma, float eps, const float* wscale, float* gx,
                              float* aff, hipStream_t st) {
  if (N <= 0) return NL_OK;
  dim3 grid((unsigned)nl_cdiv(N, 4));
  if (W <= 64) hipLaunchKernelGGL(ln_agg_backward_kernel<1>, grid, dim3(256), 0, st, FC, G, gy, (int)N, W, gamma, eps, wscale, gx, aff);
  else if (W <= 128) hipLaunchKernelGGL(ln_agg_backward_kernel<2>, grid, dim3(256), 0, st, FC, G, gy, (int)N, W, gamma, eps, wscale, gx, aff);
  else if (W <= 256) hipLaunchKernelGGL(ln_agg_backward_kernel<4>, grid, dim3(256), 0, st, FC, G, gy, (int)N, W, gamma, eps, wscale, gx, aff);
  else return NL_ERR_UNSUPPORTED;
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_attn_backward(const float* Q, const float* KV, const float* gO, int64_t N, int K, float* gQ, float* gKV, hipStream_t st) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(attn_backward_kernel, dim3((unsigned)nl_cdiv(N, 4)), dim3(256), 0, st, Q, KV, gO, (int)N, K, gQ, gKV);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_lrelu_mask(float* g, const float* h, size_t n, hipStream_t st) {   // n % 4 == 0, 16-B aligned
  if (n == 0) return NL_OK;
  hipLaunchKernelGGL(lrelu_mask_kernel, dim3((unsigned)nl_cdiv((int64_t)(n / 4), 256)), dim3(256), 0, st, (float4*)g, (const float4*)h, n / 4);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_add(const float* a, const float* b, float* o, size_t n, hipStream_t st) {
  if (n == 0) return NL_OK;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)nl_cdiv((int64_t)(n / 4), 256)), dim3(256), 0, st, (const float4*)a, (const float4*)b, (float4*)o, n / 4);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_point_encode_backward(const float* xyz, const float* dir, int dir_stride, int dir_div, int64_t N, int K, int64_t M, const int* idx,
                                    const float* sp_xyz, const float* sp_dir, const float* rd_w, float inv_span, const float* gX, int ldg, float* g_xyz,
                                    float* g_dir, float* tr, hipStream_t st) {
  if (N <= 0) return NL_OK;
  if (K == 8 && (ldg & 3) == 0 && ldg >= 92 && ((((size_t)gX) | ((size_t)tr)) & 15) == 0) {   // lane = row
    hipLaunchKernelGGL(point_encode_backward_rows_kernel, dim3((unsigned)nl_cdiv(N * 8, 256)), dim3(256), 0, st, xyz, dir, dir_stride, dir_div > 0 ? dir_div : 1, (int)N,
                       (int)(M > 0x7fffffff ? 0x7fffffff : M), idx, sp_xyz, sp_dir, rd_w, inv_span, gX, ldg, g_xyz, dir ? g_dir : nullptr, tr);
    NL_LAUNCH_CHECK();
    return NL_OK;
  }
  hipLaunchKernelGGL(point_encode_backward_kernel, dim3((unsigned)nl_cdiv(N, 4)), dim3(256), 0, st, xyz, dir, dir_stride, dir_div > 0 ? dir_div : 1, (int)N, K,
                     (int)(M > 0x7fffffff ? 0x7fffffff : M), idx, sp_xyz, sp_dir, rd_w, inv_span, gX, ldg, g_xyz, dir ? g_dir : nullptr, tr);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

// =====================================================================================================================
// Multi-view aggregation (rows a4-a7) and colour blend (a15): gradient w.r.t. the sample positions (and, for the blend, feature_agg and
// the query camera centre) with frozen weights / frozen support maps.  Three kernels share the work:
//   blend_backward_kernel    (lane = sample)  d rgb_s -> gradient of the per-(sample, view) inputs of rgb_blending_mlp: the 32 blend-projected
//                            feature taps, the tapped colour, the visibility, the 4 view-angle features; and of the per-sample projection
//   mv_geom_backward_kernel  (wave = sample)  gradients w.r.t. tapped values (from the statistics and / or from the blend) -> re-gathers the
//                            bilinear taps, contracts with their spatial derivative -> pixel -> camera -> world; view angles; the visibility
//                            weights' normalisation; emits d/d visibility and d/d depth-difference per (view, sample)
//   dec_backward_kernel      (lane = (view, sample))  the NeuRay decoders + visibility formula (visibility_decoder.py:64-148) backwards, through the
//                            border-mode tap of the visibility map and the NeuRay projection
#include "mvdec.h"

namespace {
using namespace nlmv;


struct TapD { int o[4]; float m[4]; float e, w, s, n; };   // clamped texel offsets, validity (0/1), fractional weights: value = s e t0 + s w t1 + n e t2 + n w t3
__device__ __forceinline__ TapD make_tapd(const Taps& t, int Wm, int Hm) {
  TapD d;
  const int x0 = min(max(t.x0, 0), Wm - 1), x1 = min(max(t.x0 + 1, 0), Wm - 1), y0 = min(max(t.y0, 0), Hm - 1), y1 = min(max(t.y0 + 1, 0), Hm - 1);
  d.o[0] = y0 * Wm + x0; d.o[1] = y0 * Wm + x1; d.o[2] = y1 * Wm + x0; d.o[3] = y1 * Wm + x1;
  d.m[0] = (t.mn && t.mw) ? 1.f : 0.f; d.m[1] = (t.mn && t.me) ? 1.f : 0.f; d.m[2] = (t.ms && t.mw) ? 1.f : 0.f; d.m[3] = (t.ms && t.me) ? 1.f : 0.f;
  // make_taps: nw = s e, ne = s w, sw = n e, se = n w with w = frac(ix), e = 1 - w, n = frac(iy), s = 1 - n
  d.s = t.nw + t.ne; d.n = t.sw + t.se; d.e = t.nw + t.sw; d.w = t.ne + t.se;
  return d;
}

// One wave per sample.  Inputs (each may be null = zero): g393 (N, ldg) gradient of the statistics row [mean F | var F | mean_dd, var_dd, mean_w];
// g_pf (N*V, 32) gradient of the blend-projected feature taps; g_rgbv (N*V, 4) gradient of [tapped r, g, b | visibility]; g_ang (N*V, 4).
// Outputs: g_xyz (N,3) (written), g_qc (N,3) or null, g_vis / g_dd (V,N) (written).
// SC (training, the maps' scatter-adds): a workgroup is EIGHT consecutive samples of a ray, whose taps in a view mostly fall on the same few texels.  For every
// view a wave leaves its gradient vector, its four texel keys and tap weights in LDS (no barrier in the view loops: a barrier per view made the eight waves wait
// for each other's tap loads and cost more than the atomics it saved); after ONE barrier a wave sums, for each texel it is the FIRST (wave, tap) of a view to name,
// all the contributions to it (ballot over the view's 32 keys) and issues that texel's atomics: ~10 texels per view instead of 32 tap rows — the atomics were 1.2
// of this kernel's 1.9 ms in a training step.  Waves past N run along with zero gradients and write nothing.
template <int VT, bool SC>
__global__ __launch_bounds__(SC ? 512 : 256) void mv_geom_backward_kernel(const NlViews vw, const float* __restrict__ viewsdev, const float* __restrict__ images,
                                                               const float* __restrict__ feat, int C, const float* __restrict__ pfeat,
                                                               const float* __restrict__ xyz, int N, const float* __restrict__ vis_in,
                                                               const float* __restrict__ dd_in, const float* __restrict__ g393, int ldg,
                                                               const float* __restrict__ g_pf, const float* __restrict__ g_rgbv,
                                                               const float* __restrict__ g_ang, float* __restrict__ g_xyz, float* __restrict__ g_qc,
                                                               float* __restrict__ g_vis, float* __restrict__ g_dd,
                                                               float* __restrict__ sc_feat /* training: (V,h,w,C) += , or null */,
                                                               float* __restrict__ sc_pfeat /* training: (V,h,w,32) +=, or null */,
                                                               const float* __restrict__ stats /* the forward's statistics rows (N, ldg) or null */) {
  constexpr int NW = SC ? 8 : 4;
  __shared__ float sc_vec[SC ? VT : 1][SC ? NW : 1][SC ? 64 : 1][3];   // lane's channels lane, lane + 64, lane + 128 side by side
  __shared__ float sc_pvec[SC ? VT : 1][SC ? NW : 1][SC ? 32 : 1];
  __shared__ int sc_key[SC ? VT : 1][32];
  __shared__ float sc_wt[SC ? VT : 1][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_raw = blockIdx.x * NW + wave;
  if (!SC && n_raw >= N) return;
  const bool act = n_raw < N;
  const int n = act ? n_raw : N - 1;
  const int V = vw.V, F = C + 3;
  // a view's texel keys / tap weights (wave-uniform values) into slot b
  auto publish = [&](int b, const int (&o)[4], const float (&m)[4], float s_, float e_, float w_, float n_) __attribute__((always_inline)) {
    if (lane < 4) {
      const int kq = lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3];
      const float mq = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
      const float wq = lane == 0 ? s_ * e_ : lane == 1 ? s_ * w_ : lane == 2 ? n_ * e_ : n_ * w_;
      sc_key[b][wave * 4 + lane] = (mq != 0.f && act) ? kq : -1;
      sc_wt[b][wave * 4 + lane] = wq;
    }
  };
  // the block's merge for one view (after the barrier): vec = sc_vec or sc_pvec
  auto merge_scatter = [&](int b, const int (&o)[4], const float (&m)[4], float* __restrict__ dst, int Cn, int v, size_t fmapsz, bool pf) __attribute__((always_inline)) {
    const int kk = sc_key[b][lane & 31];
    const float ww = sc_wt[b][lane & 31];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int mykey = (m[k] != 0.f && act) ? o[k] : -1;
      if (mykey < 0) continue;
      unsigned mask = (unsigned)__ballot(kk == mykey);
      // the texel's owner: one of the (wave, tap) pairs that name it, picked by the key (the first one would leave most of a ray's texels to wave 0)
      int nth = (int)((unsigned)mykey % (unsigned)__popc(mask));
      unsigned mm = mask;
      while (nth-- > 0) mm &= mm - 1;
      if (__ffs((int)mm) - 1 != wave * 4 + k) continue;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      while (mask) {   // two contributions in flight
        const int b0 = __ffs((int)mask) - 1;
        mask &= mask - 1;
        const int b1 = mask ? __ffs((int)mask) - 1 : b0;
        const float w0 = bk_rl(ww, b0), w1 = mask ? bk_rl(ww, b1) : 0.f;
        mask &= mask - (mask != 0u);
        if (pf) {
          if (lane < 32) { const float p0_ = sc_pvec[b][b0 >> 2][lane], p1_ = sc_pvec[b][b1 >> 2][lane]; a0 = fmaf(w0, p0_, a0); a0 = fmaf(w1, p1_, a0); }
        } else {
          const float* q0 = sc_vec[b][b0 >> 2][lane];
          const float* q1 = sc_vec[b][b1 >> 2][lane];
          const float x0 = q0[0], y0 = q0[1], z0 = q0[2], x1 = q1[0], y1 = q1[1], z1 = q1[2];
          a0 = fmaf(w0, x0, a0); a1 = fmaf(w0, y0, a1); a2 = fmaf(w0, z0, a2);
          a0 = fmaf(w1, x1, a0); a1 = fmaf(w1, y1, a1); a2 = fmaf(w1, z1, a2);
        }
      }
      float* sb = dst + (size_t)v * fmapsz * Cn + (size_t)mykey * Cn;
      if (pf) { if (lane < 32) atomicAdd(sb + lane, a0); }
      else {
        if (lane < Cn) atomicAdd(sb + lane, a0);
        if (lane + 64 < Cn) atomicAdd(sb + lane + 64, a1);
        if (lane + 128 < Cn) atomicAdd(sb + lane + 128, a2);
      }
    }
  };
  const float X = xyz[3 * (size_t)n], Y = xyz[3 * (size_t)n + 1], Z = xyz[3 * (size_t)n + 2];
  // ---------------------------------------------------------------- per-view scalars (lane = view)
  const int vl = lane < V ? lane : 0;
  const bool vact = lane < V;
  const float4 p0 = *(const float4*)(viewsdev + 12 * vl), p1 = *(const float4*)(viewsdev + 12 * vl + 4), p2 = *(const float4*)(viewsdev + 12 * vl + 8);
  const float cx = fmaf(p0.z, Z, fmaf(p0.y, Y, p0.x * X)) + p0.w, cy = fmaf(p1.z, Z, fmaf(p1.y, Y, p1.x * X)) + p1.w;
  const float cz = fmaf(p2.z, Z, fmaf(p2.y, Y, p2.x * X)) + p2.w;
  const float zc = fmaxf(cz, 1e-8f);
  const float pxr = cx / zc, pyr = cy / zc;
  const float px = fminf(fmaxf(pxr, -1e6f), 1e6f), py = fminf(fmaxf(pyr, -1e6f), 1e6f);
  const bool clx = !(pxr > -1e6f && pxr < 1e6f), cly = !(pyr > -1e6f && pyr < 1e6f);
  const float xn = 2.f * px / (float)(vw.Wimg - 1) - 1.f, yn = 2.f * py / (float)(vw.H - 1) - 1.f;
  const TapD tf = make_tapd(make_taps<true, false>(xn, yn, vw.w, vw.h), vw.w, vw.h);
  const TapD ti = make_tapd(make_taps<true, false>(xn, yn, vw.Wimg, vw.H), vw.Wimg, vw.H);
  const float a_vis = vact ? vis_in[(size_t)vl * N + n] : 0.f, a_dd = vact ? dd_in[(size_t)vl * N + n] : 0.f;
  float vsum = 0.f;
#pragma unroll
  for (int v = 0; v < VT; ++v) vsum += v < V ? bk_rl(a_vis, v) : 0.f;
  const float a_wgt = a_vis / (vsum + 1e-8f);
  float Wt = 0.f;   // sum of the weights (< 1 by the 1e-8)
#pragma unroll
  for (int v = 0; v < VT; ++v) Wt += v < V ? bk_rl(a_wgt, v) : 0.f;
  const float omW = 1.f - Wt;

  // ---------------------------------------------------------------- statistics part: pass 1 = weighted means, pass 2 = gradients
  // channel slots of this lane: feature channels lane, lane + 64, lane + 128 (slots 0..2) and colour plane `lane` (slot 3, lanes 0..2)
  auto tapval = [&](const float* base, size_t stride, const int (&o)[4], const float (&m)[4], float (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = base[(size_t)o[k] * stride] * m[k];
  };
  float gix = 0.f, giy = 0.f, gixI = 0.f, giyI = 0.f, gw = 0.f;   // lane v accumulates view v's totals at the end (wave-reduced per view below)
  float acc_ix[VT], acc_iy[VT], acc_ixI[VT], acc_iyI[VT], acc_w[VT];
#pragma unroll
  for (int v = 0; v < VT; ++v) acc_ix[v] = acc_iy[v] = acc_ixI[v] = acc_iyI[v] = acc_w[v] = 0.f;
  const size_t fmap = (size_t)vw.h * vw.w, imap = (size_t)vw.H * vw.Wimg;
  if (g393) {
    const float* g = g393 + (size_t)n * ldg;
    float mean[4] = {0.f, 0.f, 0.f, 0.f};
    if (stats) {   // the weighted means are columns [0, F) of the forward's statistics row: no first pass over the taps
      const float* sr = stats + (size_t)n * ldg;
#pragma unroll
      for (int j = 0; j < 3; ++j) { const int ch = lane + 64 * j; mean[j] = ch < C ? sr[3 + ch] : 0.f; }
      mean[3] = lane < 3 ? sr[lane] : 0.f;
    }
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (v < V && !stats) {
        int o[4]; float m[4]; float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = bk_rli(tf.o[k], v); m[k] = bk_rl(tf.m[k], v); }
        const float e = bk_rl(tf.e, v), w = bk_rl(tf.w, v), s = bk_rl(tf.s, v), nn = bk_rl(tf.n, v), wg = bk_rl(a_wgt, v);
        const float* fb = feat + (size_t)v * fmap * C;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int ch = lane + 64 * j;
          if (ch < C) { tapval(fb + ch, (size_t)C, o, m, t); mean[j] = fmaf((s * e * t[0] + s * w * t[1]) + (nn * e * t[2] + nn * w * t[3]), wg, mean[j]); }
        }
        if (lane < 3) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { o[k] = bk_rli(ti.o[k], v); m[k] = bk_rl(ti.m[k], v); }
          const float ei = bk_rl(ti.e, v), wi = bk_rl(ti.w, v), si = bk_rl(ti.s, v), ni = bk_rl(ti.n, v);
          tapval(images + (size_t)v * 3 * imap + (size_t)lane * imap, 1, o, m, t);
          mean[3] = fmaf((si * ei * t[0] + si * wi * t[1]) + (ni * ei * t[2] + ni * wi * t[3]), wg, mean[3]);
        }
      }
    }
    float gm[4], gv[4];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int ch = lane + 64 * j; gm[j] = ch < C ? g[3 + ch] : 0.f; gv[j] = ch < C ? g[F + 3 + ch] : 0.f; }
    gm[3] = lane < 3 ? g[lane] : 0.f; gv[3] = lane < 3 ? g[F + lane] : 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (v < V) {
        int o[4]; float m[4]; float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = bk_rli(tf.o[k], v); m[k] = bk_rl(tf.m[k], v); }
        const float e = bk_rl(tf.e, v), w = bk_rl(tf.w, v), s = bk_rl(tf.s, v), nn = bk_rl(tf.n, v), wg = bk_rl(a_wgt, v);
        const float* fb = feat + (size_t)v * fmap * C;
        float six = 0.f, siy = 0.f, sw = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int ch = lane + 64 * j;
          if (ch < C) {
            tapval(fb + ch, (size_t)C, o, m, t);
            const float x = (s * e * t[0] + s * w * t[1]) + (nn * e * t[2] + nn * w * t[3]);
            const float d = x - mean[j];
            const float gx = wg * (gm[j] + 2.f * gv[j] * (d - mean[j] * omW));
            six = fmaf(gx, s * (t[1] - t[0]) + nn * (t[3] - t[2]), six);
            siy = fmaf(gx, e * (t[2] - t[0]) + w * (t[3] - t[1]), siy);
            sw += gm[j] * x + gv[j] * (d * d - 2.f * x * mean[j] * omW);
            if constexpr (SC) { if (sc_feat) sc_vec[v][wave][lane][j] = gx; }   // grid_sample's backward towards the map (zeros padding: masked taps receive nothing)
          }
        }
        if constexpr (SC) { if (sc_feat) publish(v, o, m, s, e, w, nn); }
        float sixI = 0.f, siyI = 0.f;
        if (lane < 3) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { o[k] = bk_rli(ti.o[k], v); m[k] = bk_rl(ti.m[k], v); }
          const float ei = bk_rl(ti.e, v), wi = bk_rl(ti.w, v), si = bk_rl(ti.s, v), ni = bk_rl(ti.n, v);
          tapval(images + (size_t)v * 3 * imap + (size_t)lane * imap, 1, o, m, t);
          const float x = (si * ei * t[0] + si * wi * t[1]) + (ni * ei * t[2] + ni * wi * t[3]);
          const float d = x - mean[3];
          const float gx = wg * (gm[3] + 2.f * gv[3] * (d - mean[3] * omW));
          sixI = gx * (si * (t[1] - t[0]) + ni * (t[3] - t[2]));
          siyI = gx * (ei * (t[2] - t[0]) + wi * (t[3] - t[1]));
          sw += gm[3] * x + gv[3] * (d * d - 2.f * x * mean[3] * omW);
        }
        acc_ix[v] = wave_sum(six); acc_iy[v] = wave_sum(siy); acc_ixI[v] = wave_sum(sixI); acc_iyI[v] = wave_sum(siyI); acc_w[v] = wave_sum(sw);
      }
    }
  }
  // ---------------------------------------------------------------- blend part: blend-projected feature taps (32 channels) and tapped colours
  if (g_pf) {
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (v < V) {
        int o[4]; float m[4]; float t[4];
        float six = 0.f, siy = 0.f, sixI = 0.f, siyI = 0.f;
        if (lane < 32) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { o[k] = bk_rli(tf.o[k], v); m[k] = bk_rl(tf.m[k], v); }
          const float e = bk_rl(tf.e, v), w = bk_rl(tf.w, v), s = bk_rl(tf.s, v), nn = bk_rl(tf.n, v);
          tapval(pfeat + (size_t)v * fmap * 32 + lane, 32, o, m, t);
          const float gx = g_pf[((size_t)n * V + v) * 32 + lane];
          six = gx * (s * (t[1] - t[0]) + nn * (t[3] - t[2]));
          siy = gx * (e * (t[2] - t[0]) + w * (t[3] - t[1]));
          if constexpr (SC) { if (sc_pfeat) sc_pvec[v][wave][lane] = gx; }
        }
        if constexpr (SC) {   // (the statistics part published this view's keys unless it did not run)
          if (sc_pfeat && !(g393 && sc_feat)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[k] = bk_rli(tf.o[k], v); m[k] = bk_rl(tf.m[k], v); }
            publish(v, o, m, bk_rl(tf.s, v), bk_rl(tf.e, v), bk_rl(tf.w, v), bk_rl(tf.n, v));
          }
        }
        if (lane < 3 && g_rgbv) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { o[k] = bk_rli(ti.o[k], v); m[k] = bk_rl(ti.m[k], v); }
          const float ei = bk_rl(ti.e, v), wi = bk_rl(ti.w, v), si = bk_rl(ti.s, v), ni = bk_rl(ti.n, v);
          tapval(images + (size_t)v * 3 * imap + (size_t)lane * imap, 1, o, m, t);
          const float gx = g_rgbv[((size_t)n * V + v) * 4 + lane];
          sixI = gx * (si * (t[1] - t[0]) + ni * (t[3] - t[2]));
          siyI = gx * (ei * (t[2] - t[0]) + wi * (t[3] - t[1]));
        }
        acc_ix[v] += wave_sum(six); acc_iy[v] += wave_sum(siy); acc_ixI[v] += wave_sum(sixI); acc_iyI[v] += wave_sum(siyI);
      }
    }
  }
  if constexpr (SC) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (v < V) {
        int o[4]; float m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = bk_rli(tf.o[k], v); m[k] = bk_rl(tf.m[k], v); }
        if (g393 && sc_feat) merge_scatter(v, o, m, sc_feat, C, v, fmap, false);
        if (g_pf && sc_pfeat) merge_scatter(v, o, m, sc_pfeat, 32, v, fmap, true);
      }
    }
  }
  // hand view v's totals to lane v
#pragma unroll
  for (int v = 0; v < VT; ++v)
    if (lane == v) { gix = acc_ix[v]; giy = acc_iy[v]; gixI = acc_ixI[v]; giyI = acc_iyI[v]; gw = acc_w[v]; }

  float mdd = 0.f;   // weighted mean of the depth differences (all lanes: cross-lane reads stay outside divergent code)
#pragma unroll
  for (int v = 0; v < VT; ++v) mdd += v < V ? bk_rl(a_dd, v) * bk_rl(a_wgt, v) : 0.f;
  // ---------------------------------------------------------------- per-view scalar backward (lane = view)
  float gX = 0.f, gY = 0.f, gZ = 0.f, gq0 = 0.f, gq1 = 0.f, gq2 = 0.f, gvis_out = 0.f, gdd_out = 0.f;
  if (vact) {
    // pixel -> camera -> world (ibrnet.py:183-188).  ix = px (w - 1) / (Wimg - 1) for the feature map, = px for the image.
    float gpx = gix * ((float)(vw.w - 1) / (float)(vw.Wimg - 1)) + gixI, gpy = giy * ((float)(vw.h - 1) / (float)(vw.H - 1)) + giyI;
    if (clx) gpx = 0.f;
    if (cly) gpy = 0.f;
    const float gcx = gpx / zc, gcy = gpy / zc, gcz = cz > 1e-8f ? -(gpx * pxr + gpy * pyr) / zc : 0.f;
    gX = p0.x * gcx + p1.x * gcy + p2.x * gcz; gY = p0.y * gcx + p1.y * gcy + p2.y * gcz; gZ = p0.z * gcx + p1.z * gcy + p2.z * gcz;
    // view angles (ibrnet.py:144-167)
    if (g_ang) {
      const float4 ga = *(const float4*)(g_ang + ((size_t)n * V + vl) * 4);
      float qc0 = vw.qcam[0], qc1 = vw.qcam[1], qc2 = vw.qcam[2];
      if (vw.qrows) { const float* qr = vw.qrows + 3 * (size_t)(n / vw.qS); qc0 = qr[0]; qc1 = qr[1]; qc2 = qr[2]; }
      const float rq[3] = {qc0 - X, qc1 - Y, qc2 - Z};
      const float nq = sqrtf(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2]), dq = nq + 1e-6f;
      const float tq[3] = {rq[0] / dq, rq[1] / dq, rq[2] / dq};
      const float rt[3] = {viewsdev[192 + 3 * vl] - X, viewsdev[192 + 3 * vl + 1] - Y, viewsdev[192 + 3 * vl + 2] - Z};
      const float nt = sqrtf(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]), dt = nt + 1e-6f;
      const float tt[3] = {rt[0] / dt, rt[1] / dt, rt[2] / dt};
      const float df[3] = {tq[0] - tt[0], tq[1] - tt[1], tq[2] - tt[2]};
      const float ndf = sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]), nd = fmaxf(ndf, 1e-6f);
      float gdf[3] = {ga.x / nd, ga.y / nd, ga.z / nd};
      if (ndf > 1e-6f) {   // u = df / |df|
        const float dotg = (ga.x * df[0] + ga.y * df[1] + ga.z * df[2]) / (nd * nd * nd);
        gdf[0] -= df[0] * dotg; gdf[1] -= df[1] * dotg; gdf[2] -= df[2] * dotg;
      }
      const float gtq[3] = {gdf[0] + ga.w * tt[0], gdf[1] + ga.w * tt[1], gdf[2] + ga.w * tt[2]};
      const float gtt[3] = {-gdf[0] + ga.w * tq[0], -gdf[1] + ga.w * tq[1], -gdf[2] + ga.w * tq[2]};
      // t = r / (|r| + 1e-6): dt_i/dr_j = delta_ij / d - r_i r_j / (|r| d^2)
      const float cq = nq > 0.f ? (gtq[0] * rq[0] + gtq[1] * rq[1] + gtq[2] * rq[2]) / (nq * dq * dq) : 0.f;
      const float grq[3] = {gtq[0] / dq - rq[0] * cq, gtq[1] / dq - rq[1] * cq, gtq[2] / dq - rq[2] * cq};
      const float ct = nt > 0.f ? (gtt[0] * rt[0] + gtt[1] * rt[1] + gtt[2] * rt[2]) / (nt * dt * dt) : 0.f;
      const float grt[3] = {gtt[0] / dt - rt[0] * ct, gtt[1] / dt - rt[1] * ct, gtt[2] / dt - rt[2] * ct};
      gX -= grq[0] + grt[0]; gY -= grq[1] + grt[1]; gZ -= grq[2] + grt[2];
      gq0 = grq[0]; gq1 = grq[1]; gq2 = grq[2];
    }
    // depth-difference statistics + mean weight (multiview_aggregator.py:202-216) -> d/d weight, d/d depth difference
    if (g393) {
      const float* g = g393 + (size_t)n * ldg;
      const float gmd = g[2 * F], gvd = g[2 * F + 1], gmw = g[2 * F + 2];
      const float d = a_dd - mdd;
      gw += gmd * a_dd + gvd * (d * d - 2.f * a_dd * mdd * omW) + gmw / (float)V;
      gdd_out = a_wgt * (gmd + 2.f * gvd * (d - mdd * omW));
    }
  }
  // weights = vis / (sum vis + 1e-8): d/d vis_v = gw_v / (S + eps) - sum_u gw_u vis_u / (S + eps)^2
  const float den = vsum + 1e-8f;
  const float cross = wave_sum(vact ? gw * a_vis : 0.f) / (den * den);
  if (vact) {
    gvis_out = gw / den - cross;
    if (g_rgbv) gvis_out += g_rgbv[((size_t)n * V + vl) * 4 + 3];
    if (act) { g_vis[(size_t)vl * N + n] = gvis_out; g_dd[(size_t)vl * N + n] = gdd_out; }
  }
  gX = wave_sum(vact ? gX : 0.f); gY = wave_sum(vact ? gY : 0.f); gZ = wave_sum(vact ? gZ : 0.f);
  if (g_qc) { gq0 = wave_sum(vact ? gq0 : 0.f); gq1 = wave_sum(vact ? gq1 : 0.f); gq2 = wave_sum(vact ? gq2 : 0.f); }
  if (lane == 0 && act) {
    g_xyz[3 * (size_t)n] = gX; g_xyz[3 * (size_t)n + 1] = gY; g_xyz[3 * (size_t)n + 2] = gZ;
    if (g_qc) { g_qc[3 * (size_t)n] = gq0; g_qc[3 * (size_t)n + 1] = gq1; g_qc[3 * (size_t)n + 2] = gq2; }
  }
}


// The same with frozen maps (no scatter-adds) and EIGHT CONSECUTIVE SAMPLES PER WAVE, lane = 8 * sample + j — the layout of the forward's mv_stats8_kernel (mvagg.hip),
// for the same reason: with a wave per sample every bilinear tap is its own row fetch and every per-view total its own 64-lane reduction (50 of them per sample,
// back to back with the gathers they wait for: 0.70 ms per 512-ray pose step).  Here
//   phase A   lane (s, j) projects sample s into views j and j + 8: clamped texel offsets, tap validity and weights of the feature map and of the image -> one
//             24-dword slot per (sample, view) in LDS; the views' weights and their sums by three DPP steps over the sample's 8 lanes
//   phase B   per 32-channel chunk lane (s, j) owns channels 32 i + 4 j .. + 3 (one 16-byte load per tap, shared cache lines between the samples of a ray);
//             the per-view totals d/d ix, d/d iy, d/d weight stay in the lane (5 VT registers) over all chunks, the colour planes (lanes j < 3) and the
//             blend's 32 projected channels, and are summed over the sample's 8 lanes ONCE at the end
//   tail      lane (s, j) runs views j and j + 8 through pixel -> camera -> world, the view angles and the weight normalisation
// Needs the forward's statistics rows when g393 is given (the weighted means) and C % 4 == 0.
template <int VT>
__global__ __launch_bounds__(256) void mv_geom_backward8_kernel(const NlViews vw, const float* __restrict__ viewsdev, const float* __restrict__ images,
                                                                const float* __restrict__ feat, int C, const float* __restrict__ pfeat,
                                                                const float* __restrict__ xyz, int N, const float* __restrict__ vis_in,
                                                                const float* __restrict__ dd_in, const float* __restrict__ g393, int ldg,
                                                                const float* __restrict__ g_pf, const float* __restrict__ g_rgbv,
                                                                const float* __restrict__ g_ang, float* __restrict__ g_xyz, float* __restrict__ g_qc,
                                                                float* __restrict__ g_vis, float* __restrict__ g_dd, const float* __restrict__ stats) {
  __shared__ float4 slot[4][8][VT][6];   // [wave][sample][view]: feature map {offsets}, {validity}, {e, w, s, n}; image likewise
  __shared__ float swg[4][8][VT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s8 = lane >> 3, j = lane & 7;
  const int n_raw = blockIdx.x * 32 + wave * 8 + s8;
  const bool act = n_raw < N;
  const int n = act ? n_raw : N - 1;
  const int V = vw.V, F = C + 3;
  const float X = xyz[3 * (size_t)n], Y = xyz[3 * (size_t)n + 1], Z = xyz[3 * (size_t)n + 2];
  auto sum8 = [](float v) __attribute__((always_inline)) { return nl_sum8(v); };   // three DPP steps over the sample's 8 lanes
  struct Proj { float cz, zc, pxr, pyr; bool clx, cly; float4 p0, p1, p2; float xn, yn; };
  auto project = [&](int vl) __attribute__((always_inline)) {
    Proj q;
    q.p0 = *(const float4*)(viewsdev + 12 * vl); q.p1 = *(const float4*)(viewsdev + 12 * vl + 4); q.p2 = *(const float4*)(viewsdev + 12 * vl + 8);
    const float cx = fmaf(q.p0.z, Z, fmaf(q.p0.y, Y, q.p0.x * X)) + q.p0.w, cy = fmaf(q.p1.z, Z, fmaf(q.p1.y, Y, q.p1.x * X)) + q.p1.w;
    q.cz = fmaf(q.p2.z, Z, fmaf(q.p2.y, Y, q.p2.x * X)) + q.p2.w;
    q.zc = fmaxf(q.cz, 1e-8f);
    q.pxr = cx / q.zc; q.pyr = cy / q.zc;
    const float px = fminf(fmaxf(q.pxr, -1e6f), 1e6f), py = fminf(fmaxf(q.pyr, -1e6f), 1e6f);
    q.clx = !(q.pxr > -1e6f && q.pxr < 1e6f); q.cly = !(q.pyr > -1e6f && q.pyr < 1e6f);
    q.xn = 2.f * px / (float)(vw.Wimg - 1) - 1.f; q.yn = 2.f * py / (float)(vw.H - 1) - 1.f;
    return q;
  };
  // ---------------------------------------------------------------- phase A
  float a_vis[2], a_dd[2], a_wgt[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int v = j + 8 * p;
    const bool vact = v < V && v < VT;
    const int vl = vact ? v : 0;
    const Proj q = project(vl);
    const TapD tf = make_tapd(make_taps<true, false>(q.xn, q.yn, vw.w, vw.h), vw.w, vw.h);
    const TapD ti = make_tapd(make_taps<true, false>(q.xn, q.yn, vw.Wimg, vw.H), vw.Wimg, vw.H);
    if (v < VT) {
      float4* sl = slot[wave][s8][v];
      sl[0] = make_float4(__int_as_float(tf.o[0]), __int_as_float(tf.o[1]), __int_as_float(tf.o[2]), __int_as_float(tf.o[3]));
      sl[1] = make_float4(tf.m[0], tf.m[1], tf.m[2], tf.m[3]);
      sl[2] = make_float4(tf.e, tf.w, tf.s, tf.n);
      sl[3] = make_float4(__int_as_float(ti.o[0]), __int_as_float(ti.o[1]), __int_as_float(ti.o[2]), __int_as_float(ti.o[3]));
      sl[4] = make_float4(ti.m[0], ti.m[1], ti.m[2], ti.m[3]);
      sl[5] = make_float4(ti.e, ti.w, ti.s, ti.n);
    }
    a_vis[p] = vact ? vis_in[(size_t)vl * N + n] : 0.f;
    a_dd[p] = vact ? dd_in[(size_t)vl * N + n] : 0.f;
  }
  const float vsum = sum8(a_vis[0] + a_vis[1]);
  a_wgt[0] = a_vis[0] / (vsum + 1e-8f); a_wgt[1] = a_vis[1] / (vsum + 1e-8f);
  const float omW = 1.f - sum8(a_wgt[0] + a_wgt[1]);   // 1 - sum of the weights (> 0 by the 1e-8)
  const float mdd = sum8(a_dd[0] * a_wgt[0] + a_dd[1] * a_wgt[1]);
  if (j < VT) swg[wave][s8][j] = a_wgt[0];
  if (j + 8 < VT) swg[wave][s8][j + 8] = a_wgt[1];
  __syncthreads();

  // ---------------------------------------------------------------- phase B
  float aix[VT], aiy[VT], aixI[VT], aiyI[VT], aw[VT];
#pragma unroll
  for (int v = 0; v < VT; ++v) aix[v] = aiy[v] = aixI[v] = aiyI[v] = aw[v] = 0.f;
  const size_t fmap = (size_t)vw.h * vw.w, imap = (size_t)vw.H * vw.Wimg;
  auto tap4 = [](const float* base, size_t stride, const float4& so, const float4& sm, float4 (&t)[4]) __attribute__((always_inline)) {
    t[0] = *(const float4*)(base + (size_t)__float_as_int(so.x) * stride); t[1] = *(const float4*)(base + (size_t)__float_as_int(so.y) * stride);
    t[2] = *(const float4*)(base + (size_t)__float_as_int(so.z) * stride); t[3] = *(const float4*)(base + (size_t)__float_as_int(so.w) * stride);
    t[0].x *= sm.x; t[0].y *= sm.x; t[0].z *= sm.x; t[0].w *= sm.x;
    t[1].x *= sm.y; t[1].y *= sm.y; t[1].z *= sm.y; t[1].w *= sm.y;
    t[2].x *= sm.z; t[2].y *= sm.z; t[2].z *= sm.z; t[2].w *= sm.z;
    t[3].x *= sm.w; t[3].y *= sm.w; t[3].z *= sm.w; t[3].w *= sm.w;
  };
  if (g393) {
    const float* g = g393 + (size_t)n * ldg;
    const float* sr = stats + (size_t)n * ldg;
    for (int ch0 = 0; ch0 < C; ch0 += 32) {
      const int chr = ch0 + 4 * j;
      const bool cok = chr < C;
      const int ch = cok ? chr : 0;
      float mean[4], gm[4], gv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { mean[c] = cok ? sr[3 + ch + c] : 0.f; gm[c] = cok ? g[3 + ch + c] : 0.f; gv[c] = cok ? g[F + 3 + ch + c] : 0.f; }
#pragma unroll
      for (int v = 0; v < VT; ++v) {
        if (v < V) {
          const float4 so = slot[wave][s8][v][0], sm = slot[wave][s8][v][1], sf = slot[wave][s8][v][2];
          const float e = sf.x, w = sf.y, s = sf.z, nn = sf.w, wg = swg[wave][s8][v];
          float4 t[4];
          tap4(feat + (size_t)v * fmap * C + ch, (size_t)C, so, sm, t);
          const float t0[4] = {t[0].x, t[0].y, t[0].z, t[0].w}, t1[4] = {t[1].x, t[1].y, t[1].z, t[1].w}, t2[4] = {t[2].x, t[2].y, t[2].z, t[2].w},
                      t3[4] = {t[3].x, t[3].y, t[3].z, t[3].w};
          float six = 0.f, siy = 0.f, sw = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float x = (s * e * t0[c] + s * w * t1[c]) + (nn * e * t2[c] + nn * w * t3[c]);
            const float d = x - mean[c];
            const float gx = wg * (gm[c] + 2.f * gv[c] * (d - mean[c] * omW));
            six = fmaf(gx, s * (t1[c] - t0[c]) + nn * (t3[c] - t2[c]), six);
            siy = fmaf(gx, e * (t2[c] - t0[c]) + w * (t3[c] - t1[c]), siy);
            sw += gm[c] * x + gv[c] * (d * d - 2.f * x * mean[c] * omW);
          }
          aix[v] += six; aiy[v] += siy; aw[v] += sw;
        }
      }
    }
    // the colour planes: lanes j < 3
    {
      const int pl = j < 3 ? j : 0;
      const float on = j < 3 ? 1.f : 0.f;
      const float meanc = sr[pl], gmc = g[pl] * on, gvc = g[F + pl] * on;
#pragma unroll
      for (int v = 0; v < VT; ++v) {
        if (v < V) {
          const float4 so = slot[wave][s8][v][3], sm = slot[wave][s8][v][4], sf = slot[wave][s8][v][5];
          const float ei = sf.x, wi = sf.y, si = sf.z, ni = sf.w, wg = swg[wave][s8][v];
          const float* ib = images + (size_t)v * 3 * imap + (size_t)pl * imap;
          const float t0 = ib[__float_as_int(so.x)] * sm.x, t1 = ib[__float_as_int(so.y)] * sm.y, t2 = ib[__float_as_int(so.z)] * sm.z, t3 = ib[__float_as_int(so.w)] * sm.w;
          const float x = (si * ei * t0 + si * wi * t1) + (ni * ei * t2 + ni * wi * t3);
          const float d = x - meanc;
          const float gx = wg * (gmc + 2.f * gvc * (d - meanc * omW));
          aixI[v] += gx * (si * (t1 - t0) + ni * (t3 - t2));
          aiyI[v] += gx * (ei * (t2 - t0) + wi * (t3 - t1));
          aw[v] += gmc * x + gvc * (d * d - 2.f * x * meanc * omW);
        }
      }
    }
  }
  // ---------------------------------------------------------------- blend part: blend-projected feature taps (32 channels) and tapped colours
  if (g_pf) {
    const int pl = j < 3 ? j : 0;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (v < V) {
        const float4 so = slot[wave][s8][v][0], sm = slot[wave][s8][v][1], sf = slot[wave][s8][v][2];
        const float e = sf.x, w = sf.y, s = sf.z, nn = sf.w;
        float4 t[4];
        tap4(pfeat + (size_t)v * fmap * 32 + 4 * j, 32, so, sm, t);
        const float4 gx4 = *(const float4*)(g_pf + ((size_t)n * V + v) * 32 + 4 * j);
        const float t0[4] = {t[0].x, t[0].y, t[0].z, t[0].w}, t1[4] = {t[1].x, t[1].y, t[1].z, t[1].w}, t2[4] = {t[2].x, t[2].y, t[2].z, t[2].w},
                    t3[4] = {t[3].x, t[3].y, t[3].z, t[3].w}, gx[4] = {gx4.x, gx4.y, gx4.z, gx4.w};
        float six = 0.f, siy = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          six = fmaf(gx[c], s * (t1[c] - t0[c]) + nn * (t3[c] - t2[c]), six);
          siy = fmaf(gx[c], e * (t2[c] - t0[c]) + w * (t3[c] - t1[c]), siy);
        }
        aix[v] += six; aiy[v] += siy;
        if (g_rgbv) {
          const float4 io = slot[wave][s8][v][3], im = slot[wave][s8][v][4], iff = slot[wave][s8][v][5];
          const float ei = iff.x, wi = iff.y, si = iff.z, ni = iff.w;
          const float* ib = images + (size_t)v * 3 * imap + (size_t)pl * imap;
          const float u0 = ib[__float_as_int(io.x)] * im.x, u1 = ib[__float_as_int(io.y)] * im.y, u2 = ib[__float_as_int(io.z)] * im.z, u3 = ib[__float_as_int(io.w)] * im.w;
          const float gc = j < 3 ? g_rgbv[((size_t)n * V + v) * 4 + pl] : 0.f;
          aixI[v] += gc * (si * (u1 - u0) + ni * (u3 - u2));
          aiyI[v] += gc * (ei * (u2 - u0) + wi * (u3 - u1));
        }
      }
    }
  }
  // the sample's 8 lanes -> every lane holds the totals; lane (s, j) keeps those of views j and j + 8
  float gix[2] = {0.f, 0.f}, giy[2] = {0.f, 0.f}, gixI[2] = {0.f, 0.f}, giyI[2] = {0.f, 0.f}, gwv[2] = {0.f, 0.f};
#pragma unroll
  for (int v = 0; v < VT; ++v) {
    if (v < V) {
      const float a = sum8(aix[v]), b = sum8(aiy[v]), c = sum8(aixI[v]), d = sum8(aiyI[v]), e = sum8(aw[v]);
      if (j == (v & 7)) { gix[v >> 3] = a; giy[v >> 3] = b; gixI[v >> 3] = c; giyI[v >> 3] = d; gwv[v >> 3] = e; }
    }
  }
  // ---------------------------------------------------------------- per-view scalar backward: lane (s, j) = views j, j + 8
  float gX = 0.f, gY = 0.f, gZ = 0.f, gq0 = 0.f, gq1 = 0.f, gq2 = 0.f, gdd_out[2] = {0.f, 0.f}, gwa[2] = {0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int v = j + 8 * p;
    const bool vact = v < V && v < VT;
    const int vl = vact ? v : 0;
    float gw = gwv[p];
    if (vact) {
      const Proj q = project(vl);
      // pixel -> camera -> world (ibrnet.py:183-188).  ix = px (w - 1) / (Wimg - 1) for the feature map, = px for the image.
      float gpx = gix[p] * ((float)(vw.w - 1) / (float)(vw.Wimg - 1)) + gixI[p], gpy = giy[p] * ((float)(vw.h - 1) / (float)(vw.H - 1)) + giyI[p];
      if (q.clx) gpx = 0.f;
      if (q.cly) gpy = 0.f;
      const float gcx = gpx / q.zc, gcy = gpy / q.zc, gcz = q.cz > 1e-8f ? -(gpx * q.pxr + gpy * q.pyr) / q.zc : 0.f;
      gX += q.p0.x * gcx + q.p1.x * gcy + q.p2.x * gcz; gY += q.p0.y * gcx + q.p1.y * gcy + q.p2.y * gcz; gZ += q.p0.z * gcx + q.p1.z * gcy + q.p2.z * gcz;
      if (g_ang) {   // view angles (ibrnet.py:144-167)
        const float4 ga = *(const float4*)(g_ang + ((size_t)n * V + vl) * 4);
        float qc0 = vw.qcam[0], qc1 = vw.qcam[1], qc2 = vw.qcam[2];
        if (vw.qrows) { const float* qr = vw.qrows + 3 * (size_t)(n / vw.qS); qc0 = qr[0]; qc1 = qr[1]; qc2 = qr[2]; }
        const float rq[3] = {qc0 - X, qc1 - Y, qc2 - Z};
        const float nq = sqrtf(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2]), dq = nq + 1e-6f;
        const float tq[3] = {rq[0] / dq, rq[1] / dq, rq[2] / dq};
        const float rt[3] = {viewsdev[192 + 3 * vl] - X, viewsdev[192 + 3 * vl + 1] - Y, viewsdev[192 + 3 * vl + 2] - Z};
        const float nt = sqrtf(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]), dt = nt + 1e-6f;
        const float tt[3] = {rt[0] / dt, rt[1] / dt, rt[2] / dt};
        const float df[3] = {tq[0] - tt[0], tq[1] - tt[1], tq[2] - tt[2]};
        const float ndf = sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]), nd = fmaxf(ndf, 1e-6f);
        float gdf[3] = {ga.x / nd, ga.y / nd, ga.z / nd};
        if (ndf > 1e-6f) {   // u = df / |df|
          const float dotg = (ga.x * df[0] + ga.y * df[1] + ga.z * df[2]) / (nd * nd * nd);
          gdf[0] -= df[0] * dotg; gdf[1] -= df[1] * dotg; gdf[2] -= df[2] * dotg;
        }
        const float gtq[3] = {gdf[0] + ga.w * tt[0], gdf[1] + ga.w * tt[1], gdf[2] + ga.w * tt[2]};
        const float gtt[3] = {-gdf[0] + ga.w * tq[0], -gdf[1] + ga.w * tq[1], -gdf[2] + ga.w * tq[2]};
        // t = r / (|r| + 1e-6): dt_i/dr_j = delta_ij / d - r_i r_j / (|r| d^2)
        const float cq = nq > 0.f ? (gtq[0] * rq[0] + gtq[1] * rq[1] + gtq[2] * rq[2]) / (nq * dq * dq) : 0.f;
        const float grq[3] = {gtq[0] / dq - rq[0] * cq, gtq[1] / dq - rq[1] * cq, gtq[2] / dq - rq[2] * cq};
        const float ct = nt > 0.f ? (gtt[0] * rt[0] + gtt[1] * rt[1] + gtt[2] * rt[2]) / (nt * dt * dt) : 0.f;
        const float grt[3] = {gtt[0] / dt - rt[0] * ct, gtt[1] / dt - rt[1] * ct, gtt[2] / dt - rt[2] * ct};
        gX -= grq[0] + grt[0]; gY -= grq[1] + grt[1]; gZ -= grq[2] + grt[2];
        gq0 += grq[0]; gq1 += grq[1]; gq2 += grq[2];
      }
      if (g393) {   // depth-difference statistics + mean weight (multiview_aggregator.py:202-216) -> d/d weight, d/d depth difference
        const float* g = g393 + (size_t)n * ldg;
        const float gmd = g[2 * F], gvd = g[2 * F + 1], gmw = g[2 * F + 2];
        const float d = a_dd[p] - mdd;
        gw += gmd * a_dd[p] + gvd * (d * d - 2.f * a_dd[p] * mdd * omW) + gmw / (float)V;
        gdd_out[p] = a_wgt[p] * (gmd + 2.f * gvd * (d - mdd * omW));
      }
    }
    gwa[p] = vact ? gw : 0.f;
  }
  // weights = vis / (sum vis + 1e-8): d/d vis_v = gw_v / (S + eps) - sum_u gw_u vis_u / (S + eps)^2
  const float den = vsum + 1e-8f;
  const float cross = sum8(gwa[0] * a_vis[0] + gwa[1] * a_vis[1]) / (den * den);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int v = j + 8 * p;
    if (v < V && v < VT) {
      float go = gwa[p] / den - cross;
      if (g_rgbv) go += g_rgbv[((size_t)n * V + v) * 4 + 3];
      if (act) { g_vis[(size_t)v * N + n] = go; g_dd[(size_t)v * N + n] = gdd_out[p]; }
    }
  }
  gX = sum8(gX); gY = sum8(gY); gZ = sum8(gZ);
  if (g_qc) { gq0 = sum8(gq0); gq1 = sum8(gq1); gq2 = sum8(gq2); }
  if (j == 0 && act) {
    g_xyz[3 * (size_t)n] = gX; g_xyz[3 * (size_t)n + 1] = gY; g_xyz[3 * (size_t)n + 2] = gZ;
    if (g_qc) { g_qc[3 * (size_t)n] = gq0; g_qc[3 * (size_t)n + 1] = gq1; g_qc[3 * (size_t)n + 2] = gq2; }
  }
}

}  // namespace

namespace {
using namespace nlmv;

// backward of one 32-32-32-{1,2} decoder (visibility_decoder.py:64-97; ELU between the layers): recomputes the hidden layers, adds W0^T g_h1 to gx
// Scatter-add of per-lane values into a map where NEIGHBOURING LANES MOSTLY HIT THE SAME TEXEL (consecutive samples of a ray in one view): plain
// atomics serialise on that texel (7 ms per training step measured).  Segmented inclusive scan over the lanes of a group (runs of equal keys are
// contiguous along a ray), then only the last lane of every run issues the atomic: 5-8x fewer, and no two of them back to back on one address.
// key < 0: lane takes no part.  All lanes of the group must call.
template <int WIDTH, int NS>
struct SegMerge {
  bool mg[NS], last;
  __device__ __forceinline__ SegMerge(int key, int j) {
    // first lane of the CONTIGUOUS run this lane belongs to (equal keys further away, behind another texel, are a run of their own)
    // (every cross-lane read is its own statement: behind a short-circuit || the source lanes would be switched off)
    const int kprev = __shfl_up(key, 1, WIDTH), knext = __shfl_down(key, 1, WIDTH);
    int start = (j == 0 || kprev != key) ? j : 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) { const int o = __shfl_up(start, 1 << s, WIDTH); if (j >= (1 << s)) start = max(start, o); }
#pragma unroll
    for (int s = 0; s < NS; ++s) mg[s] = j - (1 << s) >= start;
    last = (j == WIDTH - 1) || knext != key;
  }
  __device__ __forceinline__ float sum(float v) const {
#pragma unroll
    for (int s = 0; s < NS; ++s) { const float o = __shfl_up(v, 1 << s, WIDTH); if (mg[s]) v += o; }
    return v;
  }
};

// trd (training): this decoder's 132 floats of the row [h1 32 | h2 32 | d pre-activation 1 32 | d pre-activation 2 32 | d outputs 2 | pad 2]
constexpr int DEC_TR_D = 132, DEC_TR_ROW = 32 + 4 * DEC_TR_D;
__device__ __forceinline__ void decoder_backward(const float* __restrict__ w, const float (&x)[32], float go0, float go1, float (&gx)[32], float* __restrict__ trd = nullptr) {
  float h1[32], h2[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    float a = w[1024 + j];
#pragma unroll
    for (int i = 0; i < 32; ++i) a = fmaf(w[j * 32 + i], x[i], a);
    h1[j] = nl_elu(a);
  }
  const float* w2 = w + 1024 + 32;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    float a = w2[1024 + j];
#pragma unroll
    for (int i = 0; i < 32; ++i) a = fmaf(w2[j * 32 + i], h1[i], a);
    h2[j] = nl_elu(a);
  }
  const float* w4 = w2 + 1024 + 32;
  float g1[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) g1[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    // d ELU(a)/da = a > 0 ? 1 : exp(a) = ELU(a) + 1
    const float gh2 = (go0 * w4[j] + go1 * w4[32 + j]) * (h2[j] > 0.f ? 1.f : h2[j] + 1.f);
    if (trd) { trd[j] = h1[j]; trd[32 + j] = h2[j]; trd[96 + j] = gh2; }
#pragma unroll
    for (int i = 0; i < 32; ++i) g1[i] = fmaf(w2[j * 32 + i], gh2, g1[i]);
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float gh1 = g1[j] * (h1[j] > 0.f ? 1.f : h1[j] + 1.f);
    if (trd) trd[64 + j] = gh1;
#pragma unroll
    for (int i = 0; i < 32; ++i) gx[i] = fmaf(w[j * 32 + i], gh1, gx[i]);
  }
  if (trd) { trd[128] = go0; trd[129] = go1; }
}

// One lane per (view, sample): backward of mv_vis_kernel (mvagg.hip).  Writes the view's contribution to d/d xyz into part (V, N, 3); a second
// kernel adds the views in a fixed order (no atomics: the gradient is bit-reproducible like the forward).
__global__ __launch_bounds__(256) void dec_backward_kernel(const NlViews vw, const float* __restrict__ visf, const float* __restrict__ dw,
                                                           const float* __restrict__ xyz, int N, const float* __restrict__ g_vis,
                                                           const float* __restrict__ g_dd, float* __restrict__ part,
                                                           float* __restrict__ tr /* training: (V*N, DEC_TR_ROW) rows, pre-zeroed, or null */,
                                                           float* __restrict__ sc_vis /* training: (V,vh,vw,32) +=, or null */) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (n >= N) return;
  float* po = part + ((size_t)v * N + n) * 3;
  const float gv = g_vis[(size_t)v * N + n], gd = g_dd[(size_t)v * N + n];
  if (gv == 0.f && gd == 0.f) { po[0] = 0.f; po[1] = 0.f; po[2] = 0.f; return; }
  const float X = xyz[3 * (size_t)n], Y = xyz[3 * (size_t)n + 1], Z = xyz[3 * (size_t)n + 2];
  const float* P = vw.P2[v];
  const float cx = fmaf(P[2], Z, fmaf(P[1], Y, P[0] * X)) + P[3], cy = fmaf(P[6], Z, fmaf(P[5], Y, P[4] * X)) + P[7];
  float depth = fmaf(P[10], Z, fmaf(P[9], Y, P[8] * X)) + P[11];
  const bool bad = fabsf(depth) < 1e-4f;
  if (bad) depth = 1e-3f;
  const float px = cx / depth, py = cy / depth;
  const bool outside = (px < -0.5f) | (px >= (float)vw.Wimg - 0.5f) | (py < -0.5f) | (py >= (float)vw.H - 0.5f);
  const bool valid = !bad && !outside;
  // visibility-map tap (border, align_corners = False) and its spatial derivative
  float x[32], dxv[32], dyv[32];
  const float xn = px / (float)(vw.Wimg - 1) * 2.f - 1.f, yn = py / (float)(vw.H - 1) * 2.f - 1.f;
  const float ixr = (xn + 1.f) * ((float)vw.vw / 2.f) - 0.5f, iyr = (yn + 1.f) * ((float)vw.vh / 2.f) - 0.5f;
  const bool cxl = !(ixr > 0.f && ixr < (float)(vw.vw - 1)), cyl = !(iyr > 0.f && iyr < (float)(vw.vh - 1));   // clamped by the border mode: no gradient
  {
    const Taps t = make_taps<false, true>(xn, yn, vw.vw, vw.vh);
    const TapD d = make_tapd(t, vw.vw, vw.vh);
    const float* base = visf + (size_t)v * vw.vh * vw.vw * 32;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float t0 = base[(size_t)d.o[0] * 32 + c] * d.m[0], t1 = base[(size_t)d.o[1] * 32 + c] * d.m[1];
      const float t2 = base[(size_t)d.o[2] * 32 + c] * d.m[2], t3 = base[(size_t)d.o[3] * 32 + c] * d.m[3];
      x[c] = valid ? (d.s * d.e * t0 + d.s * d.w * t1) + (d.n * d.e * t2 + d.n * d.w * t3) : 0.f;
      dxv[c] = d.s * (t1 - t0) + d.n * (t3 - t2);
      dyv[c] = d.e * (t2 - t0) + d.w * (t3 - t1);
    }
  }
  float m0, m1, v0, v1, aw, vs;
  float o[4][2];
  { float dummy; decoder(dw + 0 * DEC_STRIDE, x, o[0][0], o[0][1]); decoder(dw + 1 * DEC_STRIDE, x, o[1][0], o[1][1]);
    decoder(dw + 2 * DEC_STRIDE, x, o[2][0], dummy); decoder(dw + 3 * DEC_STRIDE, x, o[3][0], dummy); o[2][1] = o[3][1] = 0.f; }
  m0 = nl_softplus(o[0][0]); m1 = nl_softplus(o[0][1]); v0 = nl_softplus(o[1][0]) + 0.05f; v1 = nl_softplus(o[1][1]) + 0.05f;
  aw = nl_sigmoid(o[2][0]); vs = nl_sigmoid(o[3][0]);
  const float ni = -1.f / vw.near_, fi = -1.f / vw.far_, span = vw.far_ - vw.near_;
  const float tref = m0 * (fi - ni) + ni;
  const float refd_raw = -1.f / tref;
  const float refd = fminf(fmaxf(refd_raw, vw.near_), vw.far_);
  const float dn = (-1.f / fmaxf(depth, 1e-5f) - ni) / (fi - ni);
  const float u0 = (dn - m0) * v0, u1 = (dn - m1) * v1;
  const float th0 = tanhf(u0), th1 = tanhf(u1);
  const float c0 = (0.5f + 0.5f * th0) * vs, c1 = (0.5f + 0.5f * th1) * vs;
  // ---- backward of vis = valid * ((1 - c0) aw + (1 - c1) (1 - aw)) and dd = |depth - refd| / span
  const float gvv = valid ? gv : 0.f;
  const float gc0 = -gvv * aw, gc1 = -gvv * (1.f - aw);
  float g_aw = gvv * (c1 - c0);
  float g_vs = gc0 * (0.5f + 0.5f * th0) + gc1 * (0.5f + 0.5f * th1);
  const float gu0 = gc0 * vs * 0.5f * (1.f - th0 * th0), gu1 = gc1 * vs * 0.5f * (1.f - th1 * th1);
  float g_dn = gu0 * v0 + gu1 * v1;
  float g_m0 = -gu0 * v0, g_m1 = -gu1 * v1, g_v0 = gu0 * (dn - m0), g_v1 = gu1 * (dn - m1);
  const float sgn = depth > refd ? 1.f : (depth < refd ? -1.f : 0.f);
  float g_depth = gd * sgn / span;
  if (refd_raw > vw.near_ && refd_raw < vw.far_) g_m0 += (-gd * sgn / span) * ((fi - ni) / (tref * tref));   // d(-1/t)/dm0 = (fi - ni) / t^2
  if (depth > 1e-5f) g_depth += g_dn / (depth * depth * (fi - ni));
  // activations: softplus' = sigmoid, sigmoid' = y (1 - y)
  const float go00 = g_m0 * nl_sigmoid(o[0][0]), go01 = g_m1 * nl_sigmoid(o[0][1]);
  const float go10 = g_v0 * nl_sigmoid(o[1][0]), go11 = g_v1 * nl_sigmoid(o[1][1]);
  const float go2 = g_aw * aw * (1.f - aw), go3 = g_vs * vs * (1.f - vs);
  float gx[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) gx[c] = 0.f;
  float* trr = tr ? tr + ((size_t)v * N + n) * DEC_TR_ROW : nullptr;
  if (trr) {
#pragma unroll
    for (int c = 0; c < 32; ++c) trr[c] = x[c];
  }
  decoder_backward(dw + 0 * DEC_STRIDE, x, go00, go01, gx, trr ? trr + 32 : nullptr);
  decoder_backward(dw + 1 * DEC_STRIDE, x, go10, go11, gx, trr ? trr + 32 + DEC_TR_D : nullptr);
  decoder_backward(dw + 2 * DEC_STRIDE, x, go2, 0.f, gx, trr ? trr + 32 + 2 * DEC_TR_D : nullptr);
  decoder_backward(dw + 3 * DEC_STRIDE, x, go3, 0.f, gx, trr ? trr + 32 + 3 * DEC_TR_D : nullptr);
  float gix = 0.f, giy = 0.f;
  if (valid) {
#pragma unroll
    for (int c = 0; c < 32; ++c) { gix = fmaf(gx[c], dxv[c], gix); giy = fmaf(gx[c], dyv[c], giy); }
    if (sc_vis) {   // interpolate_feats' backward towards the DepthFusionNet map (border mode: a clamped texel receives the weight of every tap that maps to it).
      // Plain atomics: this is the exact-fp32 path (lanes leave early, so no cross-lane merging here; the MFMA kernel merges runs of equal texels)
      const Taps t = make_taps<false, true>(xn, yn, vw.vw, vw.vh);
      const TapD d = make_tapd(t, vw.vw, vw.vh);
      float* sb = sc_vis + (size_t)v * vw.vh * vw.vw * 32;
      const float wk[4] = {d.s * d.e * d.m[0], d.s * d.w * d.m[1], d.n * d.e * d.m[2], d.n * d.w * d.m[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (wk[k] != 0.f)
#pragma unroll
          for (int c = 0; c < 32; ++c) atomicAdd(sb + (size_t)d.o[k] * 32 + c, gx[c] * wk[k]);
    }
  }
  // ix = (2 px / (Wimg - 1)) vw / 2 - 0.5
  const float gpx = cxl ? 0.f : gix * (float)vw.vw / (float)(vw.Wimg - 1), gpy = cyl ? 0.f : giy * (float)vw.vh / (float)(vw.H - 1);
  float gcx = gpx / depth, gcy = gpy / depth;
  float gde = g_depth - (gpx * px + gpy * py) / depth;
  if (bad) { gde = 0.f; }   // the depth was replaced by a constant
  const float gX = P[0] * gcx + P[4] * gcy + P[8] * gde, gY = P[1] * gcx + P[5] * gcy + P[9] * gde, gZ = P[2] * gcx + P[6] * gcy + P[10] * gde;
  po[0] = gX; po[1] = gY; po[2] = gZ;
}

// MFMA version of dec_backward_kernel (non-fp32 modes): rows = (view, sample), 32 rows per wave like mv_vis_mfma_kernel.  Phase 1 decodes the tile
// (mvd_decode_tile, split-FP16) to get the six outputs the visibility formula's derivative needs; phase 2 walks the four decoders again, keeping each
// one's two hidden layers in registers (16 + 16 per lane), and multiplies back through the TRANSPOSED weights (bf16 hi / lo fragments packed with K in
// accumulator order, mvdec.h MVD_T; split-bf16: the incoming gradient's scale is arbitrary, fp16's range is not) into ONE accumulator whose register r
// holds d/d(channel this lane tapped into register r).  48 + 4 x 24 MFMAs per 32 rows instead of ~40 k FMAs per row.
//
// TRAIN: the 24 decoder tensors' gradients in the same pass.  gW = dA^T H contracts over the tile's ROWS, which live in the lanes; both operands are
// turned lane <-> register on the matrix pipe itself: D = S . P with S = the lane's own 16 values as the A operand (bf16 hi and lo planes separately: the
// products with 1.0 are exact, so D holds the plane transposed bit for bit) and P a constant 0 / 1 fragment that maps k-slot (s, hh, t) to the feature it
// carries.  D's lane = feature, its registers = the 32 rows in accumulator order — the same order for every operand, which is all a contraction needs.
// 4 MFMAs per transposed 32 x 32 operand + 6 per product (split-bf16, as wgrad.hip), accumulated over the wave's tiles in 9 x 16 registers (W1, W2 of the
// four decoders + one tile whose rows 2 d, 2 d + 1 are decoder d's output layer); biases = the transposed gradient's in-lane sums.  A wave leaves ONE
// partial set (DEC_WP floats) behind; dec_wpart_reduce_kernel adds them in a fixed order.  Before: a 560-float row per (view, sample) through HBM (1.5 GB
// per 65 k samples, written a dword per lane and row), a memset of it and 24 small launches.
constexpr int DEC_WP = 9 * 1024 + 9 * 64;   // per wave: [k 9][lane 64][r 16] accumulators + [k 9][lane 64] bias partials (k: W1 d 0..3, W2 d 4..7, output layers 8)
struct DecTrFrag { mvd_bf16x8 h[2], l[2]; };
// lane <-> register turn of one 32 (rows) x 32 (features) operand; sh / sl = the lane's 16 values as bf16 hi / lo in register order; rowsum += the feature's
// sum over this half's 16 rows
template <int STEPS>
__device__ __forceinline__ void dec_turn(const mvd_bf16x8 (&sh)[2], const mvd_bf16x8 (&sl)[2], const mvd_bf16x8 (&pm)[2], DecTrFrag& o, float* rowsum) {
  mvd_f32x16 dh, dl;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dh[r] = 0.f; dl[r] = 0.f; }
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    dh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh[s], pm[s], dh, 0, 0, 0);
    dl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl[s], pm[s], dl, 0, 0, 0);
  }
  if (rowsum) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += dh[r] + dl[r];
    *rowsum += a;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 8; ++t) { o.h[s][t] = (__bf16)dh[8 * s + t]; o.l[s][t] = (__bf16)dl[8 * s + t]; }
}
// acc[lane = b's feature][register ~ a's feature] += sum over the 32 rows a . b
__device__ __forceinline__ void dec_wacc(mvd_f32x16& acc, const DecTrFrag& a, const DecTrFrag& b) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[s], b.h[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s], b.l[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s], b.h[s], acc, 0, 0, 0);
  }
}
template <bool TRAIN>
__global__ __launch_bounds__(256) void dec_backward_mfma_kernel(const NlViews vw, const float* __restrict__ visf, const uint4* __restrict__ dpack,
                                                                const float* __restrict__ xyz, int N, int tiles_per_view, int total_tiles,
                                                                const float* __restrict__ g_vis, const float* __restrict__ g_dd,
                                                                float* __restrict__ part, float* __restrict__ wpart /* TRAIN: [wave][DEC_WP] */,
                                                                float* __restrict__ sc_vis) {
  __shared__ uint4 sw[MVD_LDS_UINT4 + 2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, j = lane & 31;
  __builtin_amdgcn_s_setreg(1473, 1);   // MODE.FP16_OVFL: the split-fp16 conversions of the recomputed forward saturate instead of overflowing to inf
  mvd_load_lds<true>(sw, dpack, tid, 256);
  for (int i = tid; i < 2048; i += 256) sw[MVD_LDS_UINT4 + i] = dpack[MVD_T + i];
  __syncthreads();
  const uint4* swt = sw + MVD_LDS_UINT4;
  const float* sf = reinterpret_cast<const float*>(sw + 2048);
  const float* b1 = sf, *b2 = sf + 128, *w4p = sf + 256;
  typedef MvdOps<true> OP;
  const float ni = -1.f / vw.near_, fi = -1.f / vw.far_, span = vw.far_ - vw.near_;
  // TRAIN: k-slot (s, hh, t) -> feature maps (identity for the tap's channels, accumulator order for hidden units / their gradients), weight-gradient accumulators
  mvd_bf16x8 pmI[2], pmU[2];
  mvd_f32x16 wacc[TRAIN ? 9 : 1];
  float bsum[TRAIN ? 9 : 1];
  if constexpr (TRAIN) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int r = 8 * s + t;
        pmI[s][t] = (__bf16)((16 * s + 8 * hh + t == j) ? 1.f : 0.f);
        pmU[s][t] = (__bf16)(((r & 3) + 8 * (r >> 2) + 4 * hh == j) ? 1.f : 0.f);
      }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      bsum[k] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) wacc[k][r] = 0.f;
    }
  }
  for (int tile = blockIdx.x * 4 + wave; tile < total_tiles; tile += gridDim.x * 4) {
    const int v = __builtin_amdgcn_readfirstlane(tile / tiles_per_view);
    const int n = (tile - v * tiles_per_view) * 32 + j;
    const bool live = n < N;
    const int nn = live ? n : N - 1;
    float* po = part + ((size_t)v * N + nn) * 3;
    const float gv = live ? g_vis[(size_t)v * N + nn] : 0.f, gd = live ? g_dd[(size_t)v * N + nn] : 0.f;
    if (__ballot(gv != 0.f || gd != 0.f) == 0ull) {
      if (!TRAIN && live && hh == 0) { po[0] = 0.f; po[1] = 0.f; po[2] = 0.f; }
      continue;
    }
    const float X = xyz[3 * (size_t)nn], Y = xyz[3 * (size_t)nn + 1], Z = xyz[3 * (size_t)nn + 2];
    const float* P = vw.P2[v];
    const float cx = fmaf(P[2], Z, fmaf(P[1], Y, P[0] * X)) + P[3], cy = fmaf(P[6], Z, fmaf(P[5], Y, P[4] * X)) + P[7];
    float depth = fmaf(P[10], Z, fmaf(P[9], Y, P[8] * X)) + P[11];
    const bool bad = fabsf(depth) < 1e-4f;
    if (bad) depth = 1e-3f;
    const float px = cx / depth, py = cy / depth;
    const bool outside = (px < -0.5f) | (px >= (float)vw.Wimg - 0.5f) | (py < -0.5f) | (py >= (float)vw.H - 0.5f);
    const bool valid = !bad && !outside;
    // this lane's 16 channels (8 hh .. + 7 and 16 + 8 hh .. + 7) of the tap and of its spatial derivative
    float x0[8], x1[8], dx[16], dy[16];
    const float xn = px / (float)(vw.Wimg - 1) * 2.f - 1.f, yn = py / (float)(vw.H - 1) * 2.f - 1.f;
    const float ixr = (xn + 1.f) * ((float)vw.vw / 2.f) - 0.5f, iyr = (yn + 1.f) * ((float)vw.vh / 2.f) - 0.5f;
    const bool cxl = !(ixr > 0.f && ixr < (float)(vw.vw - 1)), cyl = !(iyr > 0.f && iyr < (float)(vw.vh - 1));
    const TapD d = make_tapd(make_taps<false, true>(xn, yn, vw.vw, vw.vh), vw.vw, vw.vh);
    {
      const float* base = visf + (size_t)v * vw.vh * vw.vw * 32 + 8 * hh;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
          const int co = 16 * g + 4 * c4;
          float4 t[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            t[k] = *(const float4*)(base + (size_t)d.o[k] * 32 + co);
            t[k].x *= d.m[k]; t[k].y *= d.m[k]; t[k].z *= d.m[k]; t[k].w *= d.m[k];
          }
          float* xd = g ? x1 : x0;
          const float w00 = d.s * d.e, w01 = d.s * d.w, w10 = d.n * d.e, w11 = d.n * d.w;
#define NL_TAPC(C, I)                                                                                  \
          xd[4 * c4 + I] = valid ? (w00 * t[0].C + w01 * t[1].C) + (w10 * t[2].C + w11 * t[3].C) : 0.f; \
          dx[8 * g + 4 * c4 + I] = d.s * (t[1].C - t[0].C) + d.n * (t[3].C - t[2].C);                   \
          dy[8 * g + 4 * c4 + I] = d.e * (t[2].C - t[0].C) + d.w * (t[3].C - t[1].C);
          NL_TAPC(x, 0) NL_TAPC(y, 1) NL_TAPC(z, 2) NL_TAPC(w, 3)
#undef NL_TAPC
        }
    }
    float m0, m1, v0, v1, aw, vs;
    mvd_decode_tile<true>(sw, lane, x0, x1, m0, m1, v0, v1, vs, aw);
    // ---- derivative of the visibility formula and of the depth difference (as dec_backward_kernel)
    const float tref = m0 * (fi - ni) + ni;
    const float refd_raw = -1.f / tref;
    const float refd = fminf(fmaxf(refd_raw, vw.near_), vw.far_);
    const float dn = (-1.f / fmaxf(depth, 1e-5f) - ni) / (fi - ni);
    const float u0 = (dn - m0) * v0, u1 = (dn - m1) * v1;
    const float th0 = tanhf(u0), th1 = tanhf(u1);
    const float c0 = (0.5f + 0.5f * th0) * vs, c1 = (0.5f + 0.5f * th1) * vs;
    const float gvv = valid ? gv : 0.f;
    const float gc0 = -gvv * aw, gc1 = -gvv * (1.f - aw);
    const float g_aw = gvv * (c1 - c0);
    const float g_vs = gc0 * (0.5f + 0.5f * th0) + gc1 * (0.5f + 0.5f * th1);
    const float gu0 = gc0 * vs * 0.5f * (1.f - th0 * th0), gu1 = gc1 * vs * 0.5f * (1.f - th1 * th1);
    const float g_dn = gu0 * v0 + gu1 * v1;
    float g_m0 = -gu0 * v0;
    const float g_m1 = -gu1 * v1, g_v0 = gu0 * (dn - m0), g_v1 = gu1 * (dn - m1);
    const float sgn = depth > refd ? 1.f : (depth < refd ? -1.f : 0.f);
    float g_depth = gd * sgn / span;
    if (refd_raw > vw.near_ && refd_raw < vw.far_) g_m0 += (-gd * sgn / span) * ((fi - ni) / (tref * tref));
    if (depth > 1e-5f) g_depth += g_dn / (depth * depth * (fi - ni));
    // output activations' derivatives from the outputs themselves: softplus' = 1 - exp(-softplus), sigmoid' = y (1 - y)
    float go[4][2];
    go[0][0] = g_m0 * (1.f - expf(-m0)); go[0][1] = g_m1 * (1.f - expf(-m1));
    go[1][0] = g_v0 * (1.f - expf(-(v0 - 0.05f))); go[1][1] = g_v1 * (1.f - expf(-(v1 - 0.05f)));
    go[2][0] = g_aw * aw * (1.f - aw); go[2][1] = 0.f;
    go[3][0] = g_vs * vs * (1.f - vs); go[3][1] = 0.f;
    // ---- phase 2: one decoder at a time, hidden layers kept, transposed products accumulate d/dx
    OP::v8 xh[2], xl[2];
    OP::split(x0, xh[0], xl[0]);
    OP::split(x1, xh[1], xl[1]);
    DecTrFrag tx;   // TRAIN: the tap, lane = channel, k = rows
    if constexpr (TRAIN) {
      mvd_bf16x8 sh[2], sl[2];
      mvd_split_bf16(x0, sh[0], sl[0]);
      mvd_split_bf16(x1, sh[1], sl[1]);
      dec_turn<2>(sh, sl, pmI, tx, nullptr);
    }
    mvd_f32x16 gxa;
#pragma unroll
    for (int r = 0; r < 16; ++r) gxa[r] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      mvd_f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) { const float4 b = *(const float4*)(b1 + 32 * d + 8 * g + 4 * hh); acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w; }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const OP::v8 ah = __builtin_bit_cast(OP::v8, sw[MVD_W1 + (q * 4 + d) * 64 + lane]), al = __builtin_bit_cast(OP::v8, sw[MVD_W1 + 512 + (q * 4 + d) * 64 + lane]);
        acc = OP::mfma(al, xh[q], acc); acc = OP::mfma(ah, xl[q], acc); acc = OP::mfma(ah, xh[q], acc);
      }
      float h1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) h1[r] = nl_elu_fast(acc[r]);
      OP::v8 gh[2], gl[2];
      DecTrFrag th1, th2;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { float vv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) vv[t] = h1[8 * s2 + t];
        OP::split(vv, gh[s2], gl[s2]); }
      if constexpr (TRAIN) {
        mvd_bf16x8 sh[2], sl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { float vv[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) vv[t] = h1[8 * s2 + t];
          mvd_split_bf16(vv, sh[s2], sl[s2]); }
        dec_turn<2>(sh, sl, pmU, th1, nullptr);
      }
      mvd_f32x16 acc2;
#pragma unroll
      for (int g = 0; g < 4; ++g) { const float4 b = *(const float4*)(b2 + 32 * d + 8 * g + 4 * hh); acc2[4 * g] = b.x; acc2[4 * g + 1] = b.y; acc2[4 * g + 2] = b.z; acc2[4 * g + 3] = b.w; }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const OP::v8 ah = __builtin_bit_cast(OP::v8, sw[MVD_W2 + (d * 2 + s2) * 64 + lane]), al = __builtin_bit_cast(OP::v8, sw[MVD_W2 + 512 + (d * 2 + s2) * 64 + lane]);
        acc2 = OP::mfma(al, gh[s2], acc2); acc2 = OP::mfma(ah, gl[s2], acc2); acc2 = OP::mfma(ah, gh[s2], acc2);
      }
      // d/d(hidden 2 pre-activation): output weights (this half's 16 units) x ELU'
      float ga[16], h2v[TRAIN ? 16 : 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float h2 = nl_elu_fast(acc2[r]);
        const float g2 = go[d][0] * w4p[((d * 2 + 0) * 2 + hh) * 16 + r] + go[d][1] * w4p[((d * 2 + 1) * 2 + hh) * 16 + r];
        ga[r] = g2 * (h2 > 0.f ? 1.f : h2 + 1.f);
        if constexpr (TRAIN) h2v[r] = h2;
      }
      if constexpr (TRAIN) {   // output layer: rows 2 d, 2 d + 1 of the ninth tile = d(out 0), d(out 1) x hidden 2
        mvd_bf16x8 sh[2], sl[2], pg[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { float vv[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) vv[t] = h2v[8 * s2 + t];
          mvd_split_bf16(vv, sh[s2], sl[s2]); }
        dec_turn<2>(sh, sl, pmU, th2, nullptr);
        float gv8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) gv8[t] = (hh == 0 && t < 2) ? go[d][t] : 0.f;
        mvd_split_bf16(gv8, sh[0], sl[0]);
#pragma unroll
        for (int t = 0; t < 8; ++t) pg[0][t] = (__bf16)((hh == 0 && t < 2 && j == 2 * d + t) ? 1.f : 0.f);
        pg[1] = pg[0];   // (unused: one k-step)
        DecTrFrag tgo;
        dec_turn<1>(sh, sl, pg, tgo, &bsum[8]);
        dec_wacc(wacc[8], tgo, th2);
      }
      mvd_bf16x8 bh[2], bl[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { float vv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) vv[t] = ga[8 * s2 + t];
        mvd_split_bf16(vv, bh[s2], bl[s2]); }
      if constexpr (TRAIN) {
        DecTrFrag tda;
        dec_turn<2>(bh, bl, pmU, tda, &bsum[4 + d]);
        dec_wacc(wacc[4 + d], tda, th1);
      }
      mvd_f32x16 accb;
#pragma unroll
      for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const mvd_bf16x8 ah = __builtin_bit_cast(mvd_bf16x8, swt[(d * 2 + s2) * 64 + lane]), al = __builtin_bit_cast(mvd_bf16x8, swt[512 + (d * 2 + s2) * 64 + lane]);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s2], accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s2], accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s2], accb, 0, 0, 0);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { float vv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int r = 8 * s2 + t; vv[t] = accb[r] * (h1[r] > 0.f ? 1.f : h1[r] + 1.f);
        }
        mvd_split_bf16(vv, bh[s2], bl[s2]); }
      if constexpr (TRAIN) {
        DecTrFrag tda;
        dec_turn<2>(bh, bl, pmU, tda, &bsum[d]);
        dec_wacc(wacc[d], tda, tx);
      }
      if constexpr (!TRAIN)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const mvd_bf16x8 ah = __builtin_bit_cast(mvd_bf16x8, swt[1024 + (d * 2 + s2) * 64 + lane]), al = __builtin_bit_cast(mvd_bf16x8, swt[1024 + 512 + (d * 2 + s2) * 64 + lane]);
        gxa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s2], gxa, 0, 0, 0);
        gxa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s2], gxa, 0, 0, 0);
        gxa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s2], gxa, 0, 0, 0);
      }
    }
    if constexpr (TRAIN) continue;   // the weight-gradient pass ends here: input gradients and the map's scatter-add are the other instantiation's
    float gix = 0.f, giy = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { gix = fmaf(gxa[r], dx[r], gix); giy = fmaf(gxa[r], dy[r], giy); }
    gix += __shfl_xor(gix, 32, 64); giy += __shfl_xor(giy, 32, 64);
    if (!valid) { gix = 0.f; giy = 0.f; }
    if (sc_vis) {   // this lane's 16 channels of the map gradient; the 32 rows of the tile are consecutive samples: runs of equal texels are merged first
      float* sb = sc_vis + (size_t)v * vw.vh * vw.vw * 32 + 8 * hh;
      const float wk[4] = {d.s * d.e * d.m[0], d.s * d.w * d.m[1], d.n * d.e * d.m[2], d.n * d.w * d.m[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool on = valid && live && wk[k] != 0.f;
        const int key = on ? d.o[k] : -1 - j;
        const SegMerge<32, 5> sm(key, j);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float t = sm.sum(on ? gxa[r] * wk[k] : 0.f);
          if (sm.last && on) atomicAdd(sb + (size_t)key * 32 + (r < 8 ? r : 8 + r), t);
        }
      }
    }
    const float gpx = cxl ? 0.f : gix * (float)vw.vw / (float)(vw.Wimg - 1), gpy = cyl ? 0.f : giy * (float)vw.vh / (float)(vw.H - 1);
    const float gcx = gpx / depth, gcy = gpy / depth;
    float gde = g_depth - (gpx * px + gpy * py) / depth;
    if (bad) gde = 0.f;
    if (live && hh == 0) {
      po[0] = P[0] * gcx + P[4] * gcy + P[8] * gde; po[1] = P[1] * gcx + P[5] * gcy + P[9] * gde; po[2] = P[2] * gcx + P[6] * gcy + P[10] * gde;
    }
  }
  if constexpr (TRAIN) {
    float* wp = wpart + (size_t)(blockIdx.x * 4 + wave) * DEC_WP;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(float4*)(wp + ((size_t)k * 64 + lane) * 16 + 4 * q) = make_float4(wacc[k][4 * q], wacc[k][4 * q + 1], wacc[k][4 * q + 2], wacc[k][4 * q + 3]);
      wp[9 * 1024 + k * 64 + lane] = bsum[k];
    }
  }
}

// tensors T_DEC + 6 d + {0: W1 (32,32), 1: b1, 2: W2 (32,32), 3: b2, 4: W3 (2|1, 32), 5: b3} += the waves' partial sets, in a fixed order (8 interleaved
// slices of the waves per element, combined as a tree).  Block = 32 source elements x 8 slices.
struct DecGradPtrs { float* t[24]; };
__global__ __launch_bounds__(256) void dec_wpart_reduce_kernel(const float* __restrict__ wpart, int nwaves, const DecGradPtrs p) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, zz = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;   // [0, 9 * 1024): accumulators; [9 * 1024, 9 * 1024 + 9 * 32): biases (both halves of a lane pair)
  float s = 0.f;
  if (e < 9 * 1024) {
    for (int w = zz; w < nwaves; w += 8) s += wpart[(size_t)w * DEC_WP + e];
  } else if (e < 9 * 1024 + 9 * 32) {
    const int k = (e - 9 * 1024) >> 5, n = (e - 9 * 1024) & 31;
    for (int w = zz; w < nwaves; w += 8) s += wpart[(size_t)w * DEC_WP + 9 * 1024 + k * 64 + n] + wpart[(size_t)w * DEC_WP + 9 * 1024 + k * 64 + 32 + n];
  }
  red[zz][el] = s;
  __syncthreads();
  if (zz != 0 || e >= 9 * 1024 + 9 * 32) return;
  const float v = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
  if (e < 9 * 1024) {
    const int k = e >> 10, lane = (e >> 4) & 63, r = e & 15;
    const int n = lane & 31, m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (k < 8) { float* o = p.t[6 * (k & 3) + (k < 4 ? 0 : 2)]; if (o) o[m * 32 + n] += v; }
    else if (m < 8) {
      const int d = m >> 1, oo = m & 1;
      float* o = p.t[6 * d + 4];
      if (o && (d < 2 || oo == 0)) o[oo * 32 + n] += v;
    }
  } else {
    const int k = (e - 9 * 1024) >> 5, n = (e - 9 * 1024) & 31;
    if (k < 8) { float* o = p.t[6 * (k & 3) + (k < 4 ? 1 : 3)]; if (o) o[n] += v; }
    else if (n < 8) {
      const int d = n >> 1, oo = n & 1;
      float* o = p.t[6 * d + 5];
      if (o && (d < 2 || oo == 0)) o[oo] += v;
    }
  }
}

__global__ void view_sum_kernel(const float* __restrict__ part, int V, size_t n3, float* __restrict__ g_xyz) {   // g_xyz += sum_v part[v], v ascending
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  float a = g_xyz[i];
  for (int v = 0; v < V; ++v) a += part[(size_t)v * n3 + i];
  g_xyz[i] = a;
}

// One lane per sample: backward of blend_kernel (heads.hip; model.py:532-538).  hA (N,32) per-sample part of layer 1, h1 (N*V,32) per-(sample, view)
// part, rgbv (N*V,4) = [r,g,b,vis]; blw [32][8] = layer-1 columns of [rgb 3 | vis 1 | angle 4].
// -> g_hA (N,32), g_pf (N*V,32) [= d/d(layer-1 pre-activation) = d/d(blend-projected feature tap)], g_rgbv (N*V,4), g_ang (N*V,4)
__global__ __launch_bounds__(256) void blend_backward_kernel(const float* __restrict__ hA, const float* __restrict__ h1, const float* __restrict__ rgbv, int N, int V,
                                                             const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w4,
                                                             const float* __restrict__ b4, const float* __restrict__ blw, const float* __restrict__ g_rgb_s,
                                                             float* __restrict__ g_hA, float* __restrict__ g_pf, float* __restrict__ g_rgbv,
                                                             float* __restrict__ g_ang,
                                                             float* __restrict__ tr /* training: (N*V, 68) [layer-1 output 32 | d a2 16 | layer-2 output 16 | d logit | pad 3] or null */) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float xa[32];
#pragma unroll
  for (int i4 = 0; i4 < 8; ++i4) { const float4 t = *(const float4*)(hA + (size_t)n * 32 + 4 * i4); xa[4 * i4] = t.x; xa[4 * i4 + 1] = t.y; xa[4 * i4 + 2] = t.z; xa[4 * i4 + 3] = t.w; }
  auto logit = [&](int v, float (&pre1)[32], float (&a2)[16]) __attribute__((always_inline)) {
    const float* h = h1 + ((size_t)n * V + v) * 32;
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { pre1[i] = xa[i] + h[i]; x[i] = nl_lrelu(pre1[i]); }
    float o = b4[0];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float a = b2[j];
#pragma unroll
      for (int i = 0; i < 32; ++i) a = fmaf(w2[j * 32 + i], x[i], a);
      a2[j] = a;
      o = fmaf(w4[j], nl_lrelu(a), o);
    }
    return o;
  };
  float lg[NL_MAX_VIEWS];
  float mx = -3.4e38f;
  for (int v = 0; v < V; ++v) {
    float pre1[32], a2[16];
    float o = logit(v, pre1, a2);
    if (rgbv[((size_t)n * V + v) * 4 + 3] == 0.f) o = -1e9f;
    lg[v] = o; mx = fmaxf(mx, o);
  }
  float den = 0.f;
  for (int v = 0; v < V; ++v) { lg[v] = expf(lg[v] - mx); den += lg[v]; }
  const float gr = g_rgb_s[3 * (size_t)n], gg = g_rgb_s[3 * (size_t)n + 1], gb = g_rgb_s[3 * (size_t)n + 2];
  float dot = 0.f;
  for (int v = 0; v < V; ++v) {
    lg[v] /= den;
    const float* c = rgbv + ((size_t)n * V + v) * 4;
    dot += lg[v] * (gr * c[0] + gg * c[1] + gb * c[2]);
  }
  float gA[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) gA[i] = 0.f;
  for (int v = 0; v < V; ++v) {
    const float* c = rgbv + ((size_t)n * V + v) * 4;
    const float bw = lg[v];
    const float glog = (c[3] == 0.f) ? 0.f : bw * ((gr * c[0] + gg * c[1] + gb * c[2]) - dot);   // masked logits are constants
    float pre1[32], a2[16];
    (void)logit(v, pre1, a2);
    float g1[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) g1[i] = 0.f;
    float* trr = tr ? tr + ((size_t)n * V + v) * 68 : nullptr;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float ga = glog * w4[j] * (a2[j] > 0.f ? 1.f : 0.01f);
      if (trr) { trr[32 + j] = ga; trr[48 + j] = nl_lrelu(a2[j]); }
#pragma unroll
      for (int i = 0; i < 32; ++i) g1[i] = fmaf(w2[j * 32 + i], ga, g1[i]);
    }
    if (trr) {
#pragma unroll
      for (int i = 0; i < 32; ++i) trr[i] = nl_lrelu(pre1[i]);
      trr[64] = glog; trr[65] = 0.f; trr[66] = 0.f; trr[67] = 0.f;
    }
    float go[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float* pf = g_pf + ((size_t)n * V + v) * 32;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float g = g1[i] * (pre1[i] > 0.f ? 1.f : 0.01f);
      pf[i] = g;
      gA[i] += g;
#pragma unroll
      for (int k = 0; k < 8; ++k) go[k] = fmaf(blw[i * 8 + k], g, go[k]);
    }
    *(float4*)(g_rgbv + ((size_t)n * V + v) * 4) = make_float4(go[0] + bw * gr, go[1] + bw * gg, go[2] + bw * gb, go[3]);
    *(float4*)(g_ang + ((size_t)n * V + v) * 4) = make_float4(go[4], go[5], go[6], go[7]);
  }
#pragma unroll
  for (int i4 = 0; i4 < 8; ++i4) *(float4*)(g_hA + (size_t)n * 32 + 4 * i4) = make_float4(gA[4 * i4], gA[4 * i4 + 1], gA[4 * i4 + 2], gA[4 * i4 + 3]);
}

// training: the 8 per-(sample, view) inputs of rgb_blending_mlp.0 next to the projected features: [r, g, b, visibility | view-angle features 4]
// (ibrnet.py:144-167 as mv_stats8_kernel evaluates them) -> x8 (N*V, 8)
__global__ void blend_inputs8_kernel(const NlViews vw, const float* __restrict__ viewsdev, const float* __restrict__ xyz, int N, const float* __restrict__ rgbv,
                                     float* __restrict__ x8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * vw.V) return;
  const int n = i / vw.V, v = i - n * vw.V;
  const float X = xyz[3 * (size_t)n], Y = xyz[3 * (size_t)n + 1], Z = xyz[3 * (size_t)n + 2];
  float qc0 = vw.qcam[0], qc1 = vw.qcam[1], qc2 = vw.qcam[2];
  if (vw.qrows) { const float* qr = vw.qrows + 3 * (size_t)(n / vw.qS); qc0 = qr[0]; qc1 = qr[1]; qc2 = qr[2]; }
  float tq[3] = {qc0 - X, qc1 - Y, qc2 - Z};
  const float rq = 1.f / (sqrtf(tq[0] * tq[0] + tq[1] * tq[1] + tq[2] * tq[2]) + 1e-6f);
  tq[0] *= rq; tq[1] *= rq; tq[2] *= rq;
  float tt[3] = {viewsdev[192 + 3 * v] - X, viewsdev[192 + 3 * v + 1] - Y, viewsdev[192 + 3 * v + 2] - Z};
  const float rt = 1.f / (sqrtf(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]) + 1e-6f);
  tt[0] *= rt; tt[1] *= rt; tt[2] *= rt;
  const float df[3] = {tq[0] - tt[0], tq[1] - tt[1], tq[2] - tt[2]};
  const float rd = 1.f / fmaxf(sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]), 1e-6f);
  const float4 c = *(const float4*)(rgbv + (size_t)i * 4);
  *(float4*)(x8 + (size_t)i * 8) = c;
  *(float4*)(x8 + (size_t)i * 8 + 4) = make_float4(df[0] * rd, df[1] * rd, df[2] * rd, tq[0] * tt[0] + tq[1] * tt[1] + tq[2] * tt[2]);
}
// g (32, W + F + 5) columns [W, W+3) | W+F | [W+F+1, W+F+5) += t (32, 8)
__global__ void blw_unpack_kernel(const float* __restrict__ t, float* __restrict__ g, int W, int F) {
  const int i = threadIdx.x;   // 256 = 32 x 8
  const int m = i >> 3, k = i & 7;
  const int col = k < 3 ? W + k : W + F + (k - 3);
  g[(size_t)m * (W + F + 5) + col] += t[i];
}

// g *= ELU'(pre-activation), from the layer's OUTPUT e: e > 0 ? 1 : e + 1
__global__ void elu_mask_kernel(float4* __restrict__ g, const float4* __restrict__ e, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = g[i];
  const float4 b = e[i];
  a.x *= b.x > 0.f ? 1.f : b.x + 1.f; a.y *= b.y > 0.f ? 1.f : b.y + 1.f; a.z *= b.z > 0.f ? 1.f : b.z + 1.f; a.w *= b.w > 0.f ? 1.f : b.w + 1.f;
  g[i] = a;
}

}  // namespace

int nl_launch_mv_geom_backward(const NlViews& vw, const float* viewsdev, const float* images, const float* feat, int C, const float* pfeat, const float* xyz,
                               int64_t N, const float* vis_in, const float* dd_in, const float* g393, int ldg, const float* g_pf, const float* g_rgbv,
                               const float* g_ang, float* g_xyz, float* g_qc, float* g_vis, float* g_dd, float* sc_feat, float* sc_pfeat, const float* stats,
                               hipStream_t st) {
  if (N <= 0) return NL_OK;
  if (C > 192) return NL_ERR_UNSUPPORTED;
  const bool sc = sc_feat || sc_pfeat;
  if (!sc && (!g393 || stats) && (C & 3) == 0 && vw.V <= 16 && (!g_pf || pfeat)) {   // frozen maps: eight samples per wave
    dim3 grid8((unsigned)nl_cdiv(N, 32));
#define NL_MGB8(VT) hipLaunchKernelGGL((mv_geom_backward8_kernel<VT>), grid8, dim3(256), 0, st, vw, viewsdev, images, feat, C, pfeat, xyz, (int)N, vis_in, dd_in, g393, \
                                       ldg, g_pf, g_rgbv, g_ang, g_xyz, g_qc, g_vis, g_dd, stats)
    if (vw.V <= 4) NL_MGB8(4); else if (vw.V <= 8) NL_MGB8(8); else if (vw.V <= 10) NL_MGB8(10); else NL_MGB8(16);
#undef NL_MGB8
    NL_LAUNCH_CHECK();
    return NL_OK;
  }
  dim3 grid((unsigned)nl_cdiv(N, sc ? 8 : 4));
#define NL_MGB(VT)                                                                                                                                            \
  do {                                                                                                                                                        \
    if (sc) hipLaunchKernelGGL((mv_geom_backward_kernel<VT, true>), grid, dim3(512), 0, st, vw, viewsdev, images, feat, C, pfeat, xyz, (int)N, vis_in, dd_in, g393, \
                               ldg, g_pf, g_rgbv, g_ang, g_xyz, g_qc, g_vis, g_dd, sc_feat, sc_pfeat, stats);                                                \
    else hipLaunchKernelGGL((mv_geom_backward_kernel<VT, false>), grid, dim3(256), 0, st, vw, viewsdev, images, feat, C, pfeat, xyz, (int)N, vis_in, dd_in, g393,   \
                            ldg, g_pf, g_rgbv, g_ang, g_xyz, g_qc, g_vis, g_dd, sc_feat, sc_pfeat, stats);                                                   \
  } while (0)
  if (vw.V <= 4) NL_MGB(4); else if (vw.V <= 8) NL_MGB(8); else if (vw.V <= 10) NL_MGB(10); else NL_MGB(16);
#undef NL_MGB
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_dec_train_row(void) { return DEC_TR_ROW; }
// decw: the 24 decoder tensors' gradient pointers (T_DEC order; entries may be null) or null.  fp32 mode emits the rows `tr` for abi.hip's dec_wgrads; the MFMA
// kernel accumulates the weight gradients itself (one partial set per wave in `scratch`, >= nl_dec_wpart_floats() floats) and adds them here.
size_t nl_dec_wpart_floats(void) { return (size_t)1024 * DEC_WP; }
int nl_launch_dec_backward(const NlViews& vw, const float* visf_hwc, const float* dec_w, const void* dpack, const float* xyz, int64_t N, const float* g_vis,
                           const float* g_dd, float* part /*(V,N,3) scratch*/, float* g_xyz, float* tr, float* const* decw, float* scratch, size_t scratch_floats,
                           float* sc_vis, hipStream_t st) {
  if (N <= 0) return NL_OK;
  if (dpack) {   // non-fp32 modes: the decoders on the matrix pipe
    const int tpv = (int)nl_cdiv(N, 32), total = tpv * vw.V;
    if (decw) {
      const int blocks = (int)(nl_cdiv(total, 4) < 256 ? nl_cdiv(total, 4) : 256);   // one workgroup per CU is all that is resident (276+ registers)
      if (!scratch || scratch_floats < (size_t)blocks * 4 * DEC_WP) return NL_ERR_WORKSPACE;
      hipLaunchKernelGGL(dec_backward_mfma_kernel<true>, dim3(blocks), dim3(256), 0, st, vw, visf_hwc, (const uint4*)dpack, xyz, (int)N, tpv, total, g_vis, g_dd, part,
                         scratch, nullptr);
      DecGradPtrs gp;
      for (int i = 0; i < 24; ++i) gp.t[i] = decw[i];
      hipLaunchKernelGGL(dec_wpart_reduce_kernel, dim3((unsigned)nl_cdiv(9 * 1024 + 9 * 32, 32)), dim3(256), 0, st, scratch, blocks * 4, gp);
    }
    {
      const int blocks = (int)(nl_cdiv(total, 4) < 2048 ? nl_cdiv(total, 4) : 2048);
      hipLaunchKernelGGL(dec_backward_mfma_kernel<false>, dim3(blocks), dim3(256), 0, st, vw, visf_hwc, (const uint4*)dpack, xyz, (int)N, tpv, total, g_vis, g_dd, part,
                         nullptr, sc_vis);
    }
  } else {
    if (tr) NL_CHECK_HIP(hipMemsetAsync(tr, 0, sizeof(float) * (size_t)vw.V * N * DEC_TR_ROW, st));   // rows without a gradient are skipped by the kernel
    dim3 grid((unsigned)nl_cdiv(N, 256), (unsigned)vw.V);
    hipLaunchKernelGGL(dec_backward_kernel, grid, dim3(256), 0, st, vw, visf_hwc, dec_w, xyz, (int)N, g_vis, g_dd, part, tr, sc_vis);
  }
  hipLaunchKernelGGL(view_sum_kernel, dim3((unsigned)nl_cdiv(3 * N, 256)), dim3(256), 0, st, part, vw.V, (size_t)3 * N, g_xyz);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_blend_backward(const float* hA, const float* h1, const float* rgbv, int64_t N, int V, const float* w2, const float* b2, const float* w4,
                             const float* b4, const float* blw, const float* g_rgb_s, float* g_hA, float* g_pf, float* g_rgbv, float* g_ang, float* tr,
                             hipStream_t st) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(blend_backward_kernel, dim3((unsigned)nl_cdiv(N, 256)), dim3(256), 0, st, hA, h1, rgbv, (int)N, V, w2, b2, w4, b4, blw, g_rgb_s, g_hA, g_pf,
                     g_rgbv, g_ang, tr);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_blend_inputs8(const NlViews& vw, const float* viewsdev, const float* xyz, int64_t N, const float* rgbv, float* x8, hipStream_t st) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(blend_inputs8_kernel, dim3((unsigned)nl_cdiv(N * vw.V, 256)), dim3(256), 0, st, vw, viewsdev, xyz, (int)N, rgbv, x8);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
int nl_launch_blw_unpack(const float* t, float* g, int W, int F, hipStream_t st) {
  hipLaunchKernelGGL(blw_unpack_kernel, dim3(1), dim3(256), 0, st, t, g, W, F);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_elu_mask(float* g, const float* e, size_t n, hipStream_t st) {
  if (n == 0) return NL_OK;
  hipLaunchKernelGGL(elu_mask_kernel, dim3((unsigned)nl_cdiv((int64_t)(n / 4), 256)), dim3(256), 0, st, (float4*)g, (const float4*)e, n / 4);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

// =====================================================================================================================
// Ray U-Net (row a13; ray_unet.py:5-69), gradient w.r.t. its input with frozen weights: between the transposed-weight convolutions (segment
// GEMMs, abi.hip: do_unet_backward) sits the backward of [LayerNorm over the ray's whole (L x C) slab, per-(position, channel) affine] -> ELU ->
// optional MaxPool1d(2).
namespace {

// one block per ray.  x (L, Cc) the layer's pre-LayerNorm output (recomputed by the unfused forward); g_out: gradient w.r.t. the block's output —
// (L/2, Cc) rows when pooled, (L, Cc) otherwise — with row stride ldgo (a column window of a wider gradient matrix is fine); g_x (L, Cc) contiguous.
// VEC: four channels per thread and step (Cc % 4 == 0, 16-byte aligned rows): the three passes over the slab are latency-bound (512 rays = 512 blocks), and a
// quarter of the load instructions is a third of the time (7 launches per pose step: 0.36 -> 0.2 ms).
template <bool VEC>
__global__ __launch_bounds__(256) void ln_slab_elu_backward_kernel(const float* __restrict__ xin, int L, int Cc, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps, const float* __restrict__ g_out, int ldgo,
                                                                   int pool, float* __restrict__ g_x,
                                                                   float* __restrict__ aff /* training: (R, 2 L Cc) [d y * xhat | d y] per element, or null */) {
  __shared__ float red[8];
  constexpr int VW = VEC ? 4 : 1;
  struct V4 { float v[VW]; };
  auto ld = [](const float* p) __attribute__((always_inline)) {
    V4 o;
    if constexpr (VEC) { const float4 t = *(const float4*)p; o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w; } else o.v[0] = *p;
    return o;
  };
  auto st = [](float* p, const V4& o) __attribute__((always_inline)) {
    if constexpr (VEC) *(float4*)p = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]); else *p = o.v[0];
  };
  const int r = blockIdx.x;
  const int n = L * Cc;
  const float* x = xin + (size_t)r * n;
  float* gx = g_x + (size_t)r * n;
  const float* go = g_out + (size_t)r * (pool ? L / 2 : L) * ldgo;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float x0 = x[0];
  float s = 0.f, q = 0.f;
  for (int i = VW * tid; i < n; i += VW * 256) {
    const V4 xv = ld(x + i);
#pragma unroll
    for (int e = 0; e < VW; ++e) { const float d = xv.v[e] - x0; s += d; q += d * d; }
  }
  s = wave_sum(s); q = wave_sum(q);
  if (lane == 0) { red[wave] = s; red[4 + wave] = q; }
  __syncthreads();
  const float ms = (red[0] + red[1] + red[2] + red[3]) / (float)n;
  const float mean = x0 + ms;
  const float var = fmaxf((red[4] + red[5] + red[6] + red[7]) / (float)n - ms * ms, 0.f);
  const float rstd = 1.f / sqrtf(var + eps);
  __syncthreads();
  // pass B: g_xhat = g_y * gamma into g_x (scratch), and the two slab sums of the LayerNorm backward
  float s1 = 0.f, s2 = 0.f;
  auto one = [&](int i, const V4& xv, const V4& gm, const V4& bt, const V4& ge) __attribute__((always_inline)) {
    V4 gh, a0, a1;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const float xh = (xv.v[e] - mean) * rstd;
      const float y = xh * gm.v[e] + bt.v[e];
      const float gy = ge.v[e] * (y > 0.f ? 1.f : expf(y));     // ELU'
      gh.v[e] = gy * gm.v[e];
      s1 += gh.v[e]; s2 += gh.v[e] * xh;
      a0.v[e] = gy * xh; a1.v[e] = gy;
    }
    st(gx + i, gh);
    if (aff) { st(aff + (size_t)r * 2 * n + i, a0); st(aff + (size_t)r * 2 * n + n + i, a1); }
  };
  if (pool) {
    const int half = (L / 2) * Cc;
    for (int i = VW * tid; i < half; i += VW * 256) {
      const int p = i / Cc, c = i - p * Cc;
      const int i0 = 2 * p * Cc + c, i1 = i0 + Cc;
      const V4 xa = ld(x + i0), xb = ld(x + i1), ga = ld(gamma + i0), gb = ld(gamma + i1), ba = ld(beta + i0), bb = ld(beta + i1), g = ld(go + (size_t)p * ldgo + c);
      V4 g0, g1;
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        const float a = nl_elu((xa.v[e] - mean) * rstd * ga.v[e] + ba.v[e]);
        const float b = nl_elu((xb.v[e] - mean) * rstd * gb.v[e] + bb.v[e]);
        const bool first = a >= b;     // max_pool1d keeps the first of two equal values
        g0.v[e] = first ? g.v[e] : 0.f; g1.v[e] = first ? 0.f : g.v[e];
      }
      one(i0, xa, ga, ba, g0);
      one(i1, xb, gb, bb, g1);
    }
  } else {
    for (int i = VW * tid; i < n; i += VW * 256) {
      const int t = i / Cc, c = i - t * Cc;
      one(i, ld(x + i), ld(gamma + i), ld(beta + i), ld(go + (size_t)t * ldgo + c));
    }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
  __syncthreads();
  const float m1 = (red[0] + red[1] + red[2] + red[3]) / (float)n, m2 = (red[4] + red[5] + red[6] + red[7]) / (float)n;
  // pass C (every element was written by this same thread in pass B: same index striding)
  auto fin = [&](int i) __attribute__((always_inline)) {
    const V4 xv = ld(x + i);
    V4 gv = ld(gx + i);
#pragma unroll
    for (int e = 0; e < VW; ++e) gv.v[e] = rstd * (gv.v[e] - m1 - (xv.v[e] - mean) * rstd * m2);
    st(gx + i, gv);
  };
  if (pool) {
    const int half = (L / 2) * Cc;
    for (int i = VW * tid; i < half; i += VW * 256) {
      const int p = i / Cc, c = i - p * Cc;
      const int i0 = 2 * p * Cc + c;
      fin(i0); fin(i0 + Cc);
    }
  } else {
    for (int i = VW * tid; i < n; i += VW * 256) fin(i);
  }
}

// out[r][c] = a[r][c] + b[r][c] over (rows, cols) windows with row strides (cols % 4 == 0, 16-byte aligned rows)
__global__ void add2d_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ o, int ldo, int rows, int cols4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols4) return;
  const int r = (int)(i / cols4), c = (int)(i - (size_t)r * cols4) * 4;
  const float4 x = *(const float4*)(a + (size_t)r * lda + c), y = *(const float4*)(b + (size_t)r * ldb + c);
  *(float4*)(o + (size_t)r * ldo + c) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

}  // namespace

int nl_launch_ln_slab_elu_backward(const float* x, int64_t R, int L, int Cc, const float* gamma, const float* beta, float eps, const float* g_out, int ldgo, int pool,
                                   float* g_x, float* aff, hipStream_t st) {
  if (R <= 0) return NL_OK;
  const bool vec = (Cc & 3) == 0 && (ldgo & 3) == 0 && ((((size_t)x) | ((size_t)gamma) | ((size_t)beta) | ((size_t)g_out) | ((size_t)g_x) | ((size_t)aff)) & 15) == 0;
  if (vec) hipLaunchKernelGGL(ln_slab_elu_backward_kernel<true>, dim3((unsigned)R), dim3(256), 0, st, x, L, Cc, gamma, beta, eps, g_out, ldgo, pool, g_x, aff);
  else hipLaunchKernelGGL(ln_slab_elu_backward_kernel<false>, dim3((unsigned)R), dim3(256), 0, st, x, L, Cc, gamma, beta, eps, g_out, ldgo, pool, g_x, aff);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
// g (Cc, L) += t (L, Cc)^T   (the library keeps the U-Net's LayerNorm tables position-major; the state_dict is channel-major)
static __global__ void table_add_t_kernel(const float* __restrict__ t, float* __restrict__ g, int L, int Cc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L * Cc) return;
  const int c = i / L, l = i - c * L;
  g[i] += t[(size_t)l * Cc + c];
}
int nl_launch_table_add_t(const float* t, float* g, int L, int Cc, hipStream_t st) {
  hipLaunchKernelGGL(table_add_t_kernel, dim3((unsigned)nl_cdiv((int64_t)L * Cc, 256)), dim3(256), 0, st, t, g, L, Cc);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_add2d(const float* a, int lda, const float* b, int ldb, float* o, int ldo, int64_t rows, int cols, hipStream_t st) {
  if (rows <= 0) return NL_OK;
  const int64_t n = rows * (cols / 4);
  hipLaunchKernelGGL(add2d_kernel, dim3((unsigned)nl_cdiv(n, 256)), dim3(256), 0, st, a, lda, b, ldb, o, ldo, (int)rows, cols / 4);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
