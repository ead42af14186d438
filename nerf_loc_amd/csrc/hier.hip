// Hierarchical sampling branch.
//   coarse_view_kernel   one lane per (view, ray, coarse sample): un-normalised K^-1[u,v,1] ray points at camera
//                        z-depth (depth_fusion.py:9-45), NeuRay projection + visibility-map tap + decoders,
//                        hit-probability / visibility on inverse-depth intervals (visibility_decoder.py:6-51,150-181)
//   coarse_ray_kernel    one wave per ray: mask / ground-state handling, visibility-weighted mean over views,
//                        sigmoid, exclusive-cumprod compositing -> weights (R,Sc), depth_coarse
//                        (multiview_aggregator.py:136-154, model.py:490-491)
//   sample_pdf_kernel    one wave per ray: inverse-CDF sampling of N_importance depths from interval mid-points with
//                        caller-provided uniforms (utils.py:73-112), concatenated with the base depths and sorted
//                        (model.py:492-495) by a bitonic network in LDS.
#include "mvdec.h"

namespace {
using namespace nlmv;

struct QueryCam { float w2c[12]; float kinv[9]; };

__global__ __launch_bounds__(256) void coarse_view_kernel(const NlViews vw, const QueryCam qc, const float* __restrict__ visf,
                                                          const float* __restrict__ dw, const float* __restrict__ pix /*(R,2)*/,
                                                          const float* __restrict__ zc /*(R,Sc)*/, int R, int Sc,
                                                          float* __restrict__ alpha_out /*(V,R*Sc)*/, float* __restrict__ vis_out,
                                                          float* __restrict__ mask_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (i >= R * Sc) return;
  const int r = i / Sc, s = i - r * Sc;
  // coords2rays: rot = w2c[:, :3]^T, centre = -rot @ t, direction = rot @ Kinv @ [u, v, 1]  (not normalised)
  const float u = pix[2 * r], vv = pix[2 * r + 1];
  const float* K = qc.kinv;
  const float c0 = K[0] * u + K[1] * vv + K[2], c1 = K[3] * u + K[4] * vv + K[5], c2 = K[6] * u + K[7] * vv + K[8];
  const float* Wm = qc.w2c;
  const float t0 = Wm[3], t1 = Wm[7], t2 = Wm[11];
  float cen[3], dir[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float r0 = Wm[a], r1 = Wm[4 + a], r2 = Wm[8 + a];   // row a of rot = column a of R
    cen[a] = -(r0 * t0 + r1 * t1 + r2 * t2);
    dir[a] = (r0 * c0 + r1 * c1 + r2 * c2 + cen[a]) - cen[a];
  }
  const float z = zc[i];
  const float X = cen[0] + dir[0] * z, Y = cen[1] + dir[1] * z, Z = cen[2] + dir[2] * z;
  float px, py, depth;
  const bool valid = project_neuray(vw.P2[v], X, Y, Z, vw.Wimg, vw.H, px, py, depth);
  float x[32];
  sample_visf(visf + (size_t)v * vw.vh * vw.vw * 32, vw.vh, vw.vw, vw.Wimg, vw.H, px, py, valid, x);
  float m0, m1, v0, v1, aw, vs;
  decode_all(dw, x, m0, m1, v0, v1, vs, aw);
  // query-side interval lengths in normalised inverse depth (depth2inv_dists); last = 1e6
  const float ni = -1.f / vw.near_, fi = -1.f / vw.far_;
  auto dinv = [&](float zz) { return (-1.f / zz - ni) / (fi - ni); };
  const float d_s = dinv(z);
  const float int_s = (s + 1 < Sc) ? dinv(zc[i + 1]) - d_s : 1e6f;
  const int sp = s > 0 ? s - 1 : 0;
  const float int_p = (sp + 1 < Sc) ? dinv(zc[(size_t)r * Sc + sp + 1]) - dinv(zc[(size_t)r * Sc + sp]) : 1e6f;
  // reference-side normalised depth of the projected sample; near/far of its interval (get_near_far_points, is_ref)
  const float dn = (-1.f / fmaxf(depth, 1e-5f) - ni) / (fi - ni);
  const float nearp = dn - int_p / 2.f, farp = dn + int_s / 2.f;
  const float a0 = (0.5f + 0.5f * tanhf((nearp - m0) * v0)) * vs, a1 = (0.5f + 0.5f * tanhf((nearp - m1) * v1)) * vs;
  const float b0 = (0.5f + 0.5f * tanhf((farp - m0) * v0)) * vs, b1 = (0.5f + 0.5f * tanhf((farp - m1) * v1)) * vs;
  const float visib = (1.f - a0) * aw + (1.f - a1) * (1.f - aw);
  const float hit = (b0 - a0) * aw + (b1 - a1) * (1.f - aw);
  const float eps = 1e-5f;
  const float alpha = logf(hit / (visib - hit + eps) + eps);
  alpha_out[(size_t)v * R * Sc + i] = alpha;
  vis_out[(size_t)v * R * Sc + i] = visib;
  mask_out[(size_t)v * R * Sc + i] = valid ? 1.f : 0.f;
}

// MFMA version (bf16x3 / bf16 modes): rows = (view, ray, coarse sample), 32 rows per wave, decoders = mvd_decode_tile (mvdec.h) like
// mv_vis_mfma_kernel — the per-lane fp32 decoders above cost ~4 400 FMAs per row (2.2 ms of BASELINE config 5's step).
template <bool X3>
__global__ __launch_bounds__(256) void coarse_view_mfma_kernel(const NlViews vw, const QueryCam qc, const float* __restrict__ visf,
                                                               const uint4* __restrict__ dpack, const float* __restrict__ pix /*(R,2)*/,
                                                               const float* __restrict__ zc /*(R,Sc)*/, int R, int Sc, int tiles_per_view,
                                                               int total_tiles, float* __restrict__ alpha_out /*(V,R*Sc)*/,
                                                               float* __restrict__ vis_out, float* __restrict__ mask_out) {
  __shared__ uint4 sw[MVD_LDS_UINT4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, j = lane & 31;
  if (X3) __builtin_amdgcn_s_setreg(1473, 1);   // MODE.FP16_OVFL: the split-fp16 conversions saturate instead of overflowing to inf
  mvd_load_lds<X3>(sw, dpack, tid, 256);
  __syncthreads();
  const int NR = R * Sc;
  // coords2rays: rot = w2c[:, :3]^T, centre = -rot @ t (depth_fusion.py:9-45)
  const float* Wm = qc.w2c;
  const float* K = qc.kinv;
  float cen[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) cen[a] = -(Wm[a] * Wm[3] + Wm[4 + a] * Wm[7] + Wm[8 + a] * Wm[11]);
  const float ni = -1.f / vw.near_, fi = -1.f / vw.far_;
  auto dinv = [&](float zz) { return (-1.f / zz - ni) / (fi - ni); };
  for (int tile = blockIdx.x * 4 + wave; tile < total_tiles; tile += gridDim.x * 4) {
    const int v = __builtin_amdgcn_readfirstlane(tile / tiles_per_view);
    const int i = (tile - v * tiles_per_view) * 32 + j;
    const bool live = i < NR;
    const int ii = live ? i : NR - 1;
    const int r = ii / Sc, s = ii - r * Sc;
    const float u = pix[2 * r], vv = pix[2 * r + 1];
    const float c0 = K[0] * u + K[1] * vv + K[2], c1 = K[3] * u + K[4] * vv + K[5], c2 = K[6] * u + K[7] * vv + K[8];
    float dir[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) dir[a] = (Wm[a] * c0 + Wm[4 + a] * c1 + Wm[8 + a] * c2 + cen[a]) - cen[a];   // not normalised; same rounding as the reference's pts - centre
    const float z = zc[ii];
    const float X = cen[0] + dir[0] * z, Y = cen[1] + dir[1] * z, Z = cen[2] + dir[2] * z;
    float px, py, depth;
    const bool valid = project_neuray(vw.P2[v], X, Y, Z, vw.Wimg, vw.H, px, py, depth);
    const size_t o = (size_t)v * NR + i;
    if (__ballot(valid && live) == 0ull) {   // no row of the tile projects into the view: masked out downstream (coarse_ray_kernel multiplies by the mask)
      if (live && hh == 0) { alpha_out[o] = 0.f; vis_out[o] = 0.f; mask_out[o] = 0.f; }
      continue;
    }
    float x0[8], x1[8];
    mvd_tap16(visf + (size_t)v * vw.vh * vw.vw * 32 + 8 * hh, vw.vh, vw.vw, vw.Wimg, vw.H, px, py, valid, x0, x1);
    float m0, m1, v0, v1, aw, vs;
    mvd_decode_tile<X3>(sw, lane, x0, x1, m0, m1, v0, v1, vs, aw);
    // query-side interval lengths in normalised inverse depth (depth2inv_dists); last = 1e6
    const float d_s = dinv(z);
    const float int_s = (s + 1 < Sc) ? dinv(zc[ii + 1]) - d_s : 1e6f;
    const int sp = s > 0 ? s - 1 : 0;
    const float int_p = (sp + 1 < Sc) ? dinv(zc[(size_t)r * Sc + sp + 1]) - dinv(zc[(size_t)r * Sc + sp]) : 1e6f;
    const float dn = (-1.f / fmaxf(depth, 1e-5f) - ni) / (fi - ni);
    const float nearp = dn - int_p / 2.f, farp = dn + int_s / 2.f;
    const float a0 = (0.5f + 0.5f * tanhf((nearp - m0) * v0)) * vs, a1 = (0.5f + 0.5f * tanhf((nearp - m1) * v1)) * vs;
    const float b0 = (0.5f + 0.5f * tanhf((farp - m0) * v0)) * vs, b1 = (0.5f + 0.5f * tanhf((farp - m1) * v1)) * vs;
    const float visib = (1.f - a0) * aw + (1.f - a1) * (1.f - aw);
    const float hit = (b0 - a0) * aw + (b1 - a1) * (1.f - aw);
    const float eps = 1e-5f;
    const float alpha = logf(hit / (visib - hit + eps) + eps);
    if (live && hh == 0) { alpha_out[o] = alpha; vis_out[o] = visib; mask_out[o] = valid ? 1.f : 0.f; }
  }
}

__global__ __launch_bounds__(256) void coarse_ray_kernel(const float* __restrict__ alpha_in, const float* __restrict__ vis_in,
                                                         const float* __restrict__ mask_in, const float* __restrict__ zc, int V, int R,
                                                         int Sc, float* __restrict__ weights, float* __restrict__ depth_coarse) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  // Sc <= 64: one sample per lane
  float a = 0.f;
  const bool on = lane < Sc;
  if (on) {
    const size_t i = (size_t)r * Sc + lane;
    float num = 0.f, den = 0.f;
    int nmask = 0;
    for (int v = 0; v < V; ++v) {
      const float m = mask_in[(size_t)v * R * Sc + i];
      const float al = alpha_in[(size_t)v * R * Sc + i] * m + (1.f - m) * -15.f;
      const float vs = vis_in[(size_t)v * R * Sc + i] * m;
      num += al * vs;
      den += vs;
      nmask += m != 0.f;
    }
    float al = num / fmaxf(den, 1e-8f);
    const float inv = nmask == 0 ? 1.f : 0.f;
    al = al * (1.f - inv) + inv * -15.f;
    a = nl_sigmoid(al);
  }
  // exclusive product scan of (1 - a)
  float inc = on ? 1.f - a : 1.f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(inc, o, 64);
    if (lane >= o) inc *= t;
  }
  float T = __shfl_up(inc, 1, 64);
  if (lane == 0) T = 1.f;
  const float w = on ? a * T : 0.f;
  if (on) weights[(size_t)r * Sc + lane] = w;
  const float d = wave_sum(on ? w * zc[(size_t)r * Sc + lane] : 0.f);
  if (lane == 0 && depth_coarse) depth_coarse[r] = d;
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ zc, const float* __restrict__ wc, int Sc,
                                                         const float* __restrict__ u, int Ni, const float* __restrict__ zb, int Sb,
                                                         int R, float* __restrict__ z_out) {
  __shared__ float s_cdf[4][64];
  __shared__ float s_bins[4][64];
  __shared__ float s_sort[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wv;
  if (r >= R) return;
  const int nb = Sc - 2;   // weights[:, 1:-1]; bins = Sc - 1 mid points
  const float eps = 1e-5f;
  // pdf / cdf (utils.py:86-91): sequential cumsum like torch.cumsum
  float wgt = (lane < nb) ? wc[(size_t)r * Sc + 1 + lane] + eps : 0.f;
  const float tot = wave_sum(wgt);
  const float pdf = wgt / tot;
  s_bins[wv][lane] = (lane < Sc - 1) ? 0.5f * (zc[(size_t)r * Sc + lane] + zc[(size_t)r * Sc + lane + 1]) : 0.f;
  s_cdf[wv][lane] = pdf;
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    float run = 0.f;
    float prev = 0.f;
    for (int i = 0; i < nb; ++i) { const float p = s_cdf[wv][i]; s_cdf[wv][i] = prev; run += p; prev = run; }
    s_cdf[wv][nb] = prev;   // cdf has nb + 1 entries: [0, c1, ..., c_nb]
  }
  __builtin_amdgcn_wave_barrier();
  const int S = Sb + Ni;
  for (int i = lane; i < 256; i += 64) s_sort[wv][i] = i < Sb ? zb[(size_t)r * Sb + i] : 3.4e38f;
  for (int i = lane; i < Ni; i += 64) {
    const float uu = u[(size_t)r * Ni + i];
    // searchsorted(cdf, u, right=True): first index with cdf[idx] > u, over nb + 1 entries
    int lo = 0, hi = nb + 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_cdf[wv][mid] > uu) hi = mid; else lo = mid + 1; }
    const int below = lo - 1 < 0 ? 0 : lo - 1;
    const int above = lo > nb ? nb : lo;
    const float c0 = s_cdf[wv][below], c1 = s_cdf[wv][above];
    const float b0 = s_bins[wv][below], b1 = s_bins[wv][above];
    float den = c1 - c0;
    if (den < eps) den = 1.f;
    s_sort[wv][Sb + i] = b0 + (uu - c0) / den * (b1 - b0);
  }
  __builtin_amdgcn_wave_barrier();
  // bitonic sort of 256 keys by one wave (4 keys per lane), ascending
  for (int k = 2; k <= 256; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int t = lane + 64 * e;             // 128 compare-exchange pairs per step
        const int i = 2 * j * (t / j) + (t % j);
        const int p = i + j;
        const bool up = ((i & k) == 0);
        const float x = s_sort[wv][i], y = s_sort[wv][p];
        if ((x > y) == up) { s_sort[wv][i] = y; s_sort[wv][p] = x; }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  for (int i = lane; i < S; i += 64) z_out[(size_t)r * S + i] = s_sort[wv][i];
}

}  // namespace

int nl_launch_coarse_weights(const NlViews& vw, const float* w2c_kinv_host, const float* visf_hwc, const float* dec_w, const void* dpack,
                             int precision, const float* pix, const float* zc, int64_t R, int Sc, float* ws_alpha, float* ws_vis, float* ws_mask,
                             float* weights, float* depth_coarse, hipStream_t st) {
  if (R <= 0) return NL_OK;
  if (Sc < 3 || Sc > 64) return NL_ERR_UNSUPPORTED;
  QueryCam qc;
  for (int i = 0; i < 12; ++i) qc.w2c[i] = w2c_kinv_host[i];
  for (int i = 0; i < 9; ++i) qc.kinv[i] = w2c_kinv_host[12 + i];
  if (precision == NL_PREC_F32 || R * Sc > 0x7fffffffll / 32) {
    dim3 grid((unsigned)nl_cdiv(R * Sc, 256), (unsigned)vw.V);
    hipLaunchKernelGGL(coarse_view_kernel, grid, dim3(256), 0, st, vw, qc, visf_hwc, dec_w, pix, zc, (int)R, Sc, ws_alpha, ws_vis, ws_mask);
  } else {
    const int tpv = (int)nl_cdiv(R * Sc, 32), total = tpv * vw.V;
    const int blocks = (int)(nl_cdiv(total, 4) < 2048 ? nl_cdiv(total, 4) : 2048);
    // always the split-fp16 decoders, in the throughput mode too: the coarse weights only place the resampled depths, and depths that
    // move make every per-sample output incomparable — single-bf16 decoders here cost the bf16 mode 3e-2 on `weights` for 0.3 ms
    hipLaunchKernelGGL(coarse_view_mfma_kernel<true>, dim3(blocks), dim3(256), 0, st, vw, qc, visf_hwc, (const uint4*)dpack, pix, zc, (int)R, Sc, tpv,
                       total, ws_alpha, ws_vis, ws_mask);
  }
  hipLaunchKernelGGL(coarse_ray_kernel, dim3((unsigned)nl_cdiv(R, 4)), dim3(256), 0, st, ws_alpha, ws_vis, ws_mask, zc, vw.V, (int)R, Sc,
                     weights, depth_coarse);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_sample_pdf(const float* zc, const float* wc, int Sc, const float* u, int Ni, const float* zb, int Sb, int64_t R,
                         float* z_out, hipStream_t st) {
  if (R <= 0) return NL_OK;
  if (Sc < 3 || Sc > 64 || Sb + Ni > 256 || Ni < 1) return NL_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)nl_cdiv(R, 4)), dim3(256), 0, st, zc, wc, Sc, u, Ni, zb, Sb, (int)R, z_out);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
