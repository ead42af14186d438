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
                                                              float* __restrict__ gx) {
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
                                                                    float* __restrict__ g_dir) {
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
  float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gd0 = 0.f, gd1 = 0.f, gd2 = 0.f;
  for (int k = 0; k < K; ++k) {
    const bool have = k < M;
    const int i = idx[(size_t)n * K + k];
    const float* grow = gX + ((size_t)n * K + k) * ldg;
    const float nx = have ? sp_xyz[3 * (size_t)i] : 0.f, ny = have ? sp_xyz[3 * (size_t)i + 1] : 0.f, nz = have ? sp_xyz[3 * (size_t)i + 2] : 0.f;
    const float off[3] = {(qx - nx) * inv_span, (qy - ny) * inv_span, (qz - nz) * inv_span};
    // ---- positional encoding: column j of lane j < 63
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
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
      c0 = ax == 0 ? t : 0.f; c1 = ax == 1 ? t : 0.f; c2 = ax == 2 ? t : 0.f;
    }
    gx0 += wave_sum(c0); gx1 += wave_sum(c1); gx2 += wave_sum(c2);
    // ---- ray_diff_fc (4 -> 16 -> 27, LeakyReLU after both): forward values recomputed
    if (dir) {   // (wave-uniform)
      const float ndx = have ? sp_dir[4 * (size_t)i] : 0.f, ndy = have ? sp_dir[4 * (size_t)i + 1] : 0.f, ndz = have ? sp_dir[4 * (size_t)i + 2] : 0.f;
      const float rr0 = dx - ndx, rr1 = dy - ndy, rr2 = dz - ndz;
      const float nrm = sqrtf(rr0 * rr0 + rr1 * rr1 + rr2 * rr2), nr = nrm + 1e-8f;
      const float r0 = rr0 / nr, r1 = rr1 / nr, r2 = rr2 / nr, r3 = dx * ndx + dy * ndy + dz * ndz;
      float a1[16], h[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a = rd_w[64 + j];
        a = fmaf(rd_w[j * 4 + 0], r0, a); a = fmaf(rd_w[j * 4 + 1], r1, a); a = fmaf(rd_w[j * 4 + 2], r2, a); a = fmaf(rd_w[j * 4 + 3], r3, a);
        a1[j] = a; h[j] = nl_lrelu(a);
      }
      const float* w2 = rd_w + 80;
      float ga2 = 0.f;
      if (lane < 27) {
        float a = w2[27 * 16 + lane];
#pragma unroll
        for (int j = 0; j < 16; ++j) a = fmaf(w2[lane * 16 + j], h[j], a);
        ga2 = grow[63 + lane] * (a > 0.f ? 1.f : 0.01f);
      }
      float gr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float gh = wave_sum(lane < 27 ? w2[lane * 16 + j] * ga2 : 0.f);
        const float ga1 = gh * (a1[j] > 0.f ? 1.f : 0.01f);
        gr[0] = fmaf(rd_w[j * 4 + 0], ga1, gr[0]); gr[1] = fmaf(rd_w[j * 4 + 1], ga1, gr[1]);
        gr[2] = fmaf(rd_w[j * 4 + 2], ga1, gr[2]); gr[3] = fmaf(rd_w[j * 4 + 3], ga1, gr[3]);
      }
      // u = rr / (|rr| + 1e-8): du_i/drr_j = delta_ij / nr - rr_i rr_j / (|rr| nr^2)  (0 at rr = 0, like torch.norm's subgradient)
      const float gdot = gr[0] * rr0 + gr[1] * rr1 + gr[2] * rr2;
      const float cc = nrm > 0.f ? gdot / (nrm * nr * nr) : 0.f;
      gd0 += gr[0] / nr - rr0 * cc + gr[3] * ndx;
      gd1 += gr[1] / nr - rr1 * cc + gr[3] * ndy;
      gd2 += gr[2] / nr - rr2 * cc + gr[3] * ndz;
    }
  }
  if (lane == 0) {
    g_xyz[3 * (size_t)n] = gx0 * inv_span; g_xyz[3 * (size_t)n + 1] = gx1 * inv_span; g_xyz[3 * (size_t)n + 2] = gx2 * inv_span;
    if (g_dir) { g_dir[3 * (size_t)n] = gd0; g_dir[3 * (size_t)n + 1] = gd1; g_dir[3 * (size_t)n + 2] = gd2; }
  }
}

}  // namespace

int nl_launch_ln_agg_backward(const float* FC, const float* G, const float* gy, int64_t N, int W, const float* gamma, float eps, const float* wscale, float* gx,
                              hipStream_t st) {
  if (N <= 0) return NL_OK;
  dim3 grid((unsigned)nl_cdiv(N, 4));
  if (W <= 64) hipLaunchKernelGGL(ln_agg_backward_kernel<1>, grid, dim3(256), 0, st, FC, G, gy, (int)N, W, gamma, eps, wscale, gx);
  else if (W <= 128) hipLaunchKernelGGL(ln_agg_backward_kernel<2>, grid, dim3(256), 0, st, FC, G, gy, (int)N, W, gamma, eps, wscale, gx);
  else if (W <= 256) hipLaunchKernelGGL(ln_agg_backward_kernel<4>, grid, dim3(256), 0, st, FC, G, gy, (int)N, W, gamma, eps, wscale, gx);
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
                                    float* g_dir, hipStream_t st) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(point_encode_backward_kernel, dim3((unsigned)nl_cdiv(N, 4)), dim3(256), 0, st, xyz, dir, dir_stride, dir_div > 0 ? dir_div : 1, (int)N, K,
                     (int)(M > 0x7fffffff ? 0x7fffffff : M), idx, sp_xyz, sp_dir, rd_w, inv_span, gX, ldg, g_xyz, dir ? g_dir : nullptr);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
