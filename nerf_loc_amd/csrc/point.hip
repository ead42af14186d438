// Neural-point branch glue kernels.  The Linear layers between them
// (base_mlp 285->W->W->W, k/v/q projections, fc) run on the MFMA segment-GEMM.
//
// Algebraic note (holds for the reference as written, verified against it in tests): the attention
// query is the SAME multi-view feature for all K neighbours of a sample (model.py:413-414 repeats it),
// so MultiHeadAttention's output, the residual + LayerNorm and hence base_mlp_agg_weight's logits are
// identical across the K rows; softmax over K (model.py:415) is therefore exactly 1/K and
// feature_agg = feature * sum_k w_k (model.py:419-427).  We compute fc/LayerNorm once per sample and
// keep the reference's weight arithmetic (1/clamp(dist), *1/K, *confidence, normalise) for sum_k w_k.
#include "common.h"

namespace {

// One wave per sample; rows (n,k), k < K.  X row layout: [feature F | posenc 63 | ray_diff_fc 27 | pad] (ld = ldx)
__global__ __launch_bounds__(256) void point_encode_kernel(
    const float* __restrict__ xyz, const float* __restrict__ dir, int dir_stride, int dir_div, int N, int K, int M,
    const int* __restrict__ idx, const float* __restrict__ d2,
    const float* __restrict__ sp_xyz, const float* __restrict__ sp_feat, int F, const float* __restrict__ sp_conf,
    const float* __restrict__ sp_dir, const float* __restrict__ rd_w /* W0[16][4], b0[16], W2[27][16], b2[27] */,
    float inv_span, float* __restrict__ X, int ldx, float* __restrict__ wscale) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float qx = xyz[3 * (size_t)n], qy = xyz[3 * (size_t)n + 1], qz = xyz[3 * (size_t)n + 2];
  float dx, dy, dz;
  if (dir) {
    const size_t dr = (size_t)(n / dir_div) * dir_stride;
    dx = dir[dr]; dy = dir[dr + 1]; dz = dir[dr + 2];
  } else {  // model.py:391-392: nearest neighbour's direction (zero-filled when M == 0)
    int i0 = idx[(size_t)n * K];
    bool ok = M > 0;
    dx = ok ? sp_dir[4 * (size_t)i0] : 0.f; dy = ok ? sp_dir[4 * (size_t)i0 + 1] : 0.f; dz = ok ? sp_dir[4 * (size_t)i0 + 2] : 0.f;
  }
  float wsum_raw = 0.f;
  float wk_store = 0.f;  // lane k keeps w_k
  for (int k = 0; k < K; ++k) {
    const bool have = k < M;  // knn_gather zero-fills columns k >= len (knn_utils.py:211-220)
    const int i = idx[(size_t)n * K + k];
    float* row = X + ((size_t)n * K + k) * ldx;
    // feature copy
    for (int c = lane; c < F; c += 64) row[c] = have ? sp_feat[(size_t)i * F + c] : 0.f;
    const float nx = have ? sp_xyz[3 * (size_t)i] : 0.f, ny = have ? sp_xyz[3 * (size_t)i + 1] : 0.f, nz = have ? sp_xyz[3 * (size_t)i + 2] : 0.f;
    const float off[3] = {(qx - nx) * inv_span, (qy - ny) * inv_span, (qz - nz) * inv_span};
    // positional encoding (utils.py:5-35): [x, sin(2^0 x), cos(2^0 x), ..., sin(2^9 x), cos(2^9 x)]
    if (lane < 63) {
      float val;
      if (lane < 3) val = lane == 0 ? off[0] : (lane == 1 ? off[1] : off[2]);
      else {
        const int j = lane - 3, f = j / 6, r = j - 6 * f;
        const int ax = r >= 3 ? r - 3 : r;
        const float o = ax == 0 ? off[0] : (ax == 1 ? off[1] : off[2]);
        const float arg = o * (float)(1 << f);
        val = r < 3 ? sinf(arg) : cosf(arg);
      }
      row[F + lane] = val;
    }
    // ray_diff (model.py:396-399) -> ray_diff_fc (model.py:36-39)
    const float ndx = have ? sp_dir[4 * (size_t)i] : 0.f, ndy = have ? sp_dir[4 * (size_t)i + 1] : 0.f, ndz = have ? sp_dir[4 * (size_t)i + 2] : 0.f;
    float r0 = dx - ndx, r1 = dy - ndy, r2 = dz - ndz;
    const float nr = sqrtf(r0 * r0 + r1 * r1 + r2 * r2) + 1e-8f;
    r0 /= nr; r1 /= nr; r2 /= nr;
    const float r3 = dx * ndx + dy * ndy + dz * ndz;
    if (lane < 27) {
      float h[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a = rd_w[64 + j];
        a = fmaf(rd_w[j * 4 + 0], r0, a); a = fmaf(rd_w[j * 4 + 1], r1, a);
        a = fmaf(rd_w[j * 4 + 2], r2, a); a = fmaf(rd_w[j * 4 + 3], r3, a);
        h[j] = nl_lrelu(a);
      }
      const float* w2 = rd_w + 80;
      float a = w2[27 * 16 + lane];
#pragma unroll
      for (int j = 0; j < 16; ++j) a = fmaf(w2[lane * 16 + j], h[j], a);
      row[F + 63 + lane] = nl_lrelu(a);
    } else if (lane >= 32 && lane < 32 + (ldx - F - 90)) {
      row[F + 90 + (lane - 32)] = 0.f;
    }
    // aggregation weight of neighbour k (model.py:419-425) with correlation == 1/K
    const float dist = sqrtf(d2[(size_t)n * K + k]);
    const float conf = have ? sp_conf[i] : 0.f;
    float w = 1.f / fmaxf(dist, 1e-8f);
    w = w * (1.f / (float)K);
    w = w * conf;
    wsum_raw += w;
    if (lane == k) wk_store = w;
  }
  // weights / clamp(sum) then summed again (model.py:426-427)
  const float den = fmaxf(wsum_raw, 1e-8f);
  float wn = (lane < K) ? wk_store / den : 0.f;
  float tot = wave_sum(wn);
  if (lane == 0) wscale[n] = tot;
}

// One wave per sample. Q (N,128) ; KV (N*K, 256) = [k-proj 128 | v-proj 128] ; O (N,128)
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ Q, const float* __restrict__ KV, int N, int K,
                                                   float* __restrict__ O, unsigned* __restrict__ logit_amax) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  // lane owns dims {2*lane, 2*lane+1} of the 128-wide projection; head = lane / 16 (d_k = 32)
  const float temp = 5.656854249492381f;  // d_k ** 0.5 (ibrnet.py:84)
  const float2 q = *(const float2*)(Q + (size_t)n * 128 + 2 * lane);
  const float q0 = q.x / temp, q1 = q.y / temp;
  float sc[NL_KNN_MAX_K];
  float2 vv[NL_KNN_MAX_K];
  float mx = -3.4e38f;
  float lmax = 0.f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) {
    if (k < K) {
      const float* r = KV + ((size_t)n * K + k) * 256;
      const float2 kk = *(const float2*)(r + 2 * lane);
      vv[k] = *(const float2*)(r + 128 + 2 * lane);
      float p = q0 * kk.x + q1 * kk.y;
      p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 8, 64);
      sc[k] = p;
      lmax = (p == p) ? fmaxf(lmax, fabsf(p)) : __builtin_inff();   // conditioning indicator (nl_frame_diagnostics); a NaN logit counts as +inf
      mx = fmaxf(mx, p);
    } else { sc[k] = -3.4e38f; vv[k] = make_float2(0.f, 0.f); }
  }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) { sc[k] = k < K ? expf(sc[k] - mx) : 0.f; den += sc[k]; }
  float o0 = 0.f, o1 = 0.f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) { float a = sc[k] / den; o0 += a * vv[k].x; o1 += a * vv[k].y; }
  *(float2*)(O + (size_t)n * 128 + 2 * lane) = make_float2(o0, o1);
  if (logit_amax) {   // (read first, L2-resident: an atomic only when this wave raises the maximum)
    const float m = wave_max(lmax);
    if (lane == 0 && __float_as_uint(m) > *(volatile const unsigned*)logit_amax) atomicMax(logit_amax, __float_as_uint(m));
  }
}

// One wave per sample: y = LayerNorm(fc + G; eps) * gamma + beta, times wscale -> feature_agg
template <int WPL>  // W / 64 values per lane (W multiple of 64), or W==32 handled by WPL=1 with half wave
__global__ __launch_bounds__(256) void ln_agg_kernel(const float* __restrict__ FC, const float* __restrict__ G, int N, int W,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     const float* __restrict__ wscale, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float x[WPL];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < WPL; ++j) {
    int c = lane + 64 * j;
    x[j] = c < W ? FC[(size_t)n * W + c] + G[(size_t)n * W + c] : 0.f;
    s += x[j];
  }
  const float mean = wave_sum(s) / (float)W;
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < WPL; ++j) { int c = lane + 64 * j; float d = c < W ? x[j] - mean : 0.f; v += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(v) / (float)W + eps);
  const float sc = wscale ? wscale[n] : 1.f;
#pragma unroll
  for (int j = 0; j < WPL; ++j) {
    int c = lane + 64 * j;
    if (c < W) out[(size_t)n * W + c] = ((x[j] - mean) * rstd * gamma[c] + beta[c]) * sc;
  }
}

}  // namespace

int nl_launch_point_encode(const float* xyz, const float* dir, int dir_stride, int dir_div, int64_t N, int K, int64_t M, const int* idx,
                           const float* d2, const float* sp_xyz, const float* sp_feat, int F, const float* sp_conf,
                           const float* sp_dir, const float* rd_w, float inv_span, float* X, int ldx, float* wscale, hipStream_t st) {
  if (N <= 0) return NL_OK;
  if (ldx - F - 90 < 0 || ldx - F - 90 > 32 || K > NL_KNN_MAX_K) return NL_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(point_encode_kernel, dim3((unsigned)nl_cdiv(N, 4)), dim3(256), 0, st, xyz, dir, dir_stride, dir_div > 0 ? dir_div : 1, (int)N, K,
                     (int)(M > 0x7fffffff ? 0x7fffffff : M), idx, d2, sp_xyz, sp_feat, F, sp_conf, sp_dir, rd_w, inv_span, X, ldx, wscale);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_attn(const float* Q, const float* KV, int64_t N, int K, float* O, hipStream_t st, unsigned* logit_amax) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(attn_kernel, dim3((unsigned)nl_cdiv(N, 4)), dim3(256), 0, st, Q, KV, (int)N, K, O, logit_amax);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_ln_agg(const float* FC, const float* G, int64_t N, int W, const float* gamma, const float* beta, float eps,
                     const float* wscale, float* out, hipStream_t st) {
  if (N <= 0) return NL_OK;
  dim3 grid((unsigned)nl_cdiv(N, 4));
  if (W <= 64) hipLaunchKernelGGL(ln_agg_kernel<1>, grid, dim3(256), 0, st, FC, G, (int)N, W, gamma, beta, eps, wscale, out);
  else if (W <= 128) hipLaunchKernelGGL(ln_agg_kernel<2>, grid, dim3(256), 0, st, FC, G, (int)N, W, gamma, beta, eps, wscale, out);
  else if (W <= 256) hipLaunchKernelGGL(ln_agg_kernel<4>, grid, dim3(256), 0, st, FC, G, (int)N, W, gamma, beta, eps, wscale, out);
  else return NL_ERR_UNSUPPORTED;
  NL_LAUNCH_CHECK();
  return NL_OK;
}
