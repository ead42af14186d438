// Multi-view (NeuRay-style) feature aggregation, SURVEY.md §8 rows a4-a7 (+ a15's view angles, a18's mask).
//
//   mv_vis_kernel    one lane per (view, sample): NeuRay projection (depth_fusion.py:78-126), border/align_corners=False
//                    bilinear tap of the 32-channel visibility map (depth_fusion.py:60-76, neuray_ops.py:14-36),
//                    the four 32-32-32-{2,2,1,1} decoders (visibility_decoder.py:64-107) with weights broadcast from
//                    SGPRs, visibility CDF (:109-138) and |depth - ref_depth| (:140-148, multiview_aggregator.py:83-84).
//   mv_stats_kernel  one wave per sample, lanes across channels: IBRNet projection (ibrnet.py:169-192), zeros/align_corners=True
//                    bilinear taps of rgb + C feature channels (ibrnet.py:214-222; channels-last map => each tap row is one
//                    contiguous 768-B read), visibility-weighted mean/variance over views (multiview_aggregator.py:199-216),
//                    view-angle features (ibrnet.py:144-167) and the in-bounds count for the ray mask (model.py:563-573).
// The 393->64->W out_fc (multiview_aggregator.py:31-36) then runs on the MFMA segment-GEMM.
#include "mvdec.h"

namespace {
using namespace nlmv;

__global__ void chw_to_hwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int V, int Cc, int HW) {
  // (V,Cc,HW) -> (V,HW,Cc)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)V * Cc * HW;
  if (i >= total) return;
  int c = (int)(i % Cc);
  size_t t = i / Cc;
  int p = (int)(t % HW);
  int v = (int)(t / HW);
  dst[i] = src[((size_t)v * Cc + c) * HW + p];
}

__global__ __launch_bounds__(256) void mv_vis_kernel(const NlViews vw, const float* __restrict__ visf /*(V,h,w,32)*/,
                                                     const float* __restrict__ dw, const float* __restrict__ xyz, int N,
                                                     float* __restrict__ vis_out /*(V,N)*/, float* __restrict__ dd_out /*(V,N)*/) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (n >= N) return;
  const float X = xyz[3 * (size_t)n], Y = xyz[3 * (size_t)n + 1], Z = xyz[3 * (size_t)n + 2];
  float px, py, depth;
  const bool valid = project_neuray(vw.P2[v], X, Y, Z, vw.Wimg, vw.H, px, py, depth);
  float x[32];
  sample_visf(visf + (size_t)v * vw.vh * vw.vw * 32, vw.vh, vw.vw, vw.Wimg, vw.H, px, py, valid, x);
  float m0, m1, v0, v1, aw, vs;
  decode_all(dw, x, m0, m1, v0, v1, vs, aw);

  const float ni = -1.f / vw.near_, fi = -1.f / vw.far_;
  float refd = -1.f / (m0 * (fi - ni) + ni);
  refd = fminf(fmaxf(refd, vw.near_), vw.far_);
  const float dd = fabsf(depth - refd) / (vw.far_ - vw.near_);
  const float dn = (-1.f / fmaxf(depth, 1e-5f) - ni) / (fi - ni);
  const float c0 = (0.5f + 0.5f * tanhf((dn - m0) * v0)) * vs;
  const float c1 = (0.5f + 0.5f * tanhf((dn - m1) * v1)) * vs;
  float vis = (1.f - c0) * aw + (1.f - c1) * (1.f - aw);
  vis = valid ? vis : 0.f;
  vis_out[(size_t)v * N + n] = vis;
  dd_out[(size_t)v * N + n] = dd;
}

template <int VT>
__global__ __launch_bounds__(256) void mv_stats_kernel(const NlViews vw, const float* __restrict__ images /*(V,3,H,W)*/,
                                                       const float* __restrict__ feat /*(V,h,w,C)*/, int C,
                                                       const float* __restrict__ xyz, int N,
                                                       const float* __restrict__ vis_in, const float* __restrict__ dd_in,
                                                       float* __restrict__ g393, int ldg, float* __restrict__ rgb_feat /*(N*V,196) or null*/,
                                                       float* __restrict__ vis_ang /*(N*V,8) or null*/, int* __restrict__ valid_s,
                                                       const float* __restrict__ pfeat /*(V,h,w,32) blend-projected feature map*/,
                                                       const float* __restrict__ blw /*[32][8] = W[rgb3|vis1|ang4], then bias[32]*/,
                                                       float* __restrict__ bl1 /*(N*V,32) or null*/, float* __restrict__ rgbv /*(N*V,4) or null*/) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int V = vw.V;
  const float X = xyz[3 * (size_t)n], Y = xyz[3 * (size_t)n + 1], Z = xyz[3 * (size_t)n + 2];
  const int F = C + 3;

  float wgt[VT], dd[VT];
  float vsum = 0.f;
#pragma unroll
  for (int v = 0; v < VT; ++v) {
    wgt[v] = v < V ? vis_in[(size_t)v * N + n] : 0.f;
    dd[v] = v < V ? dd_in[(size_t)v * N + n] : 0.f;
    vsum += wgt[v];
  }
  float visraw[VT];
#pragma unroll
  for (int v = 0; v < VT; ++v) { visraw[v] = wgt[v]; wgt[v] = wgt[v] / (vsum + 1e-8f); }

  float bwr[8], bbias = 0.f;   // this lane's row (j = lane < 32) of the small blend-layer weights
#pragma unroll
  for (int i = 0; i < 8; ++i) bwr[i] = (bl1 && lane < 32) ? blw[lane * 8 + i] : 0.f;
  if (bl1 && lane < 32) bbias = blw[256 + lane];
  float xv[VT][4];
  int cnt1 = 0;
  // query-camera unit ray (ibrnet.py:157-158)
  float tq[3] = {vw.qcam[0] - X, vw.qcam[1] - Y, vw.qcam[2] - Z};
  {
    float nq = sqrtf(tq[0] * tq[0] + tq[1] * tq[1] + tq[2] * tq[2]) + 1e-6f;
    tq[0] /= nq; tq[1] /= nq; tq[2] /= nq;
  }
#pragma unroll
  for (int v = 0; v < VT; ++v) {
    xv[v][0] = xv[v][1] = xv[v][2] = xv[v][3] = 0.f;
    if (v < V) {
      const float* P = vw.P1[v];
      const float cx = fmaf(P[2], Z, fmaf(P[1], Y, P[0] * X)) + P[3];
      const float cy = fmaf(P[6], Z, fmaf(P[5], Y, P[4] * X)) + P[7];
      const float cz = fmaf(P[10], Z, fmaf(P[9], Y, P[8] * X)) + P[11];
      const float zc = fmaxf(cz, 1e-8f);
      float px = cx / zc, py = cy / zc;
      px = fminf(fmaxf(px, -1e6f), 1e6f);
      py = fminf(fmaxf(py, -1e6f), 1e6f);
      const bool m1 = (px <= (float)vw.Wimg - 1.f) && (px >= 0.f) && (py <= (float)vw.H - 1.f) && (py >= 0.f) && (cz > 0.f);
      cnt1 += m1 ? 1 : 0;
      const float xn = 2.f * px / (float)(vw.Wimg - 1) - 1.f;
      const float yn = 2.f * py / (float)(vw.H - 1) - 1.f;
      // feature map taps (align_corners=True, zeros)
      float pv = 0.f;   // tap of the blend-projected map, channel = lane (< 32)
      {
        const Taps t = make_taps<true, false>(xn, yn, vw.w, vw.h);
        const float* base = feat + (size_t)v * vw.h * vw.w * C;
        const size_t o_nw = ((size_t)t.y0 * vw.w + t.x0) * C, o_ne = o_nw + C, o_sw = o_nw + (size_t)vw.w * C, o_se = o_sw + C;
        const bool a = t.mn && t.mw, b = t.mn && t.me, c = t.ms && t.mw, d = t.ms && t.me;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int ch = lane + 64 * j;
          if (ch < C) {
            float va = a ? base[o_nw + ch] : 0.f, vb = b ? base[o_ne + ch] : 0.f;
            float vc = c ? base[o_sw + ch] : 0.f, vd = d ? base[o_se + ch] : 0.f;
            xv[v][j] = va * t.nw + vb * t.ne + vc * t.sw + vd * t.se;
          }
        }
        if (bl1 && lane < 32) {
          const float* pb = pfeat + (size_t)v * vw.h * vw.w * 32 + lane;
          const size_t p_nw = ((size_t)t.y0 * vw.w + t.x0) * 32, p_sw = p_nw + (size_t)vw.w * 32;
          float va = a ? pb[p_nw] : 0.f, vb = b ? pb[p_nw + 32] : 0.f, vc = c ? pb[p_sw] : 0.f, vd = d ? pb[p_sw + 32] : 0.f;
          pv = va * t.nw + vb * t.ne + vc * t.sw + vd * t.se;
        }
      }
      // image taps: lanes 0..2 own the rgb planes
      if (lane < 3) {
        const Taps t = make_taps<true, false>(xn, yn, vw.Wimg, vw.H);
        const float* base = images + ((size_t)v * 3 + lane) * vw.H * vw.Wimg;
        const size_t o_nw = (size_t)t.y0 * vw.Wimg + t.x0;
        float va = (t.mn && t.mw) ? base[o_nw] : 0.f, vb = (t.mn && t.me) ? base[o_nw + 1] : 0.f;
        float vc = (t.ms && t.mw) ? base[o_nw + vw.Wimg] : 0.f, vd = (t.ms && t.me) ? base[o_nw + vw.Wimg + 1] : 0.f;
        xv[v][3] = va * t.nw + vb * t.ne + vc * t.sw + vd * t.se;
      }
      // view-angle features (ibrnet.py:144-167), wave-uniform
      float tt[3] = {vw.cam[v][0] - X, vw.cam[v][1] - Y, vw.cam[v][2] - Z};
      const float nt = sqrtf(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]) + 1e-6f;
      tt[0] /= nt; tt[1] /= nt; tt[2] /= nt;
      const float df[3] = {tq[0] - tt[0], tq[1] - tt[1], tq[2] - tt[2]};
      const float nd = fmaxf(sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]), 1e-6f);
      const float ang[4] = {df[0] / nd, df[1] / nd, df[2] / nd, tq[0] * tt[0] + tq[1] * tt[1] + tq[2] * tt[2]};
      if (bl1) {
        // colour-blend layer 1, per-(sample, view) part, by linearity of the bilinear tap (model.py:532-535):
        //   W[:, feat] . bilinear(featmap) == bilinear(W[:, feat] . featmap) = pv ; plus rgb / visibility / angle columns + bias
        const float r = __shfl(xv[v][3], 0, 64), g = __shfl(xv[v][3], 1, 64), bb = __shfl(xv[v][3], 2, 64);
        if (lane < 32) {
          float o = pv + bbias;
          o = fmaf(bwr[0], r, o); o = fmaf(bwr[1], g, o); o = fmaf(bwr[2], bb, o);
          o = fmaf(bwr[3], visraw[v], o);
          o = fmaf(bwr[4], ang[0], o); o = fmaf(bwr[5], ang[1], o); o = fmaf(bwr[6], ang[2], o); o = fmaf(bwr[7], ang[3], o);
          bl1[((size_t)n * V + v) * 32 + lane] = o;
        }
        if (lane < 4) rgbv[((size_t)n * V + v) * 4 + lane] = lane < 3 ? xv[v][3] : visraw[v];
      }
      if (rgb_feat) {   // stage API only: materialise the raw multi-view projection and [vis, angle]
        float* row = rgb_feat + ((size_t)n * V + v) * NL_FPAD;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int ch = lane + 64 * j;
          if (ch < C) row[3 + ch] = xv[v][j];
        }
        if (lane < 3) row[lane] = xv[v][3];
        if (lane == 3) row[F] = 0.f;
        if (lane == 0 && vis_ang) {
          float* va = vis_ang + ((size_t)n * V + v) * 8;
          *(float4*)va = make_float4(visraw[v], ang[0], ang[1], ang[2]);
          *(float4*)(va + 4) = make_float4(ang[3], 0.f, 0.f, 0.f);
        }
      }
    }
  }
  // visibility-weighted mean / variance over views (ibrnet.py:8-12)
  float* g = g393 + (size_t)n * ldg;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float mean = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) mean += xv[v][j] * wgt[v];
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { float d = xv[v][j] - mean; var += wgt[v] * (d * d); }
    int pos = -1;
    if (j < 3) { int ch = lane + 64 * j; if (ch < C) pos = 3 + ch; }
    else if (lane < 3) pos = lane;
    if (pos >= 0) { g[pos] = mean; g[F + pos] = var; }
  }
  if (lane == 0) {
    float mean = 0.f, wsum = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { mean += dd[v] * wgt[v]; wsum += wgt[v]; }
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { float d = dd[v] - mean; var += wgt[v] * (d * d); }
    g[2 * F] = mean;
    g[2 * F + 1] = var;
    g[2 * F + 2] = wsum / (float)V;
    for (int p = 2 * F + 3; p < ldg; ++p) g[p] = 0.f;
    valid_s[n] = cnt1 > 1 ? 1 : 0;
  }
}

}  // namespace

int nl_launch_chw_to_hwc(const float* src, float* dst, int V, int Cc, int HW, hipStream_t st) {
  size_t total = (size_t)V * Cc * HW;
  hipLaunchKernelGGL(chw_to_hwc_kernel, dim3((unsigned)nl_cdiv(total, 256)), dim3(256), 0, st, src, dst, V, Cc, HW);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_mv_vis(const NlViews& vw, const float* visf_hwc, const float* dec_w, const float* xyz, int64_t N,
                     float* vis_out, float* dd_out, hipStream_t st) {
  if (N <= 0) return NL_OK;
  dim3 grid((unsigned)nl_cdiv(N, 256), (unsigned)vw.V);
  hipLaunchKernelGGL(mv_vis_kernel, grid, dim3(256), 0, st, vw, visf_hwc, dec_w, xyz, (int)N, vis_out, dd_out);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_mv_stats(const NlViews& vw, const float* images, const float* feat, int C, const float* xyz, int64_t N,
                       const float* vis_in, const float* dd_in, float* g393, int ldg, float* rgb_feat, float* vis_ang,
                       int* valid_s, const float* pfeat, const float* blw, float* bl1, float* rgbv, hipStream_t st) {
  if (N <= 0) return NL_OK;
  if (C > 192) return NL_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nl_cdiv(N, 4));
  if (vw.V <= 4)
    hipLaunchKernelGGL(mv_stats_kernel<4>, grid, dim3(256), 0, st, vw, images, feat, C, xyz, (int)N, vis_in, dd_in, g393, ldg, rgb_feat, vis_ang, valid_s, pfeat, blw, bl1, rgbv);
  else if (vw.V <= 8)
    hipLaunchKernelGGL(mv_stats_kernel<8>, grid, dim3(256), 0, st, vw, images, feat, C, xyz, (int)N, vis_in, dd_in, g393, ldg, rgb_feat, vis_ang, valid_s, pfeat, blw, bl1, rgbv);
  else if (vw.V <= 10)
    hipLaunchKernelGGL(mv_stats_kernel<10>, grid, dim3(256), 0, st, vw, images, feat, C, xyz, (int)N, vis_in, dd_in, g393, ldg, rgb_feat, vis_ang, valid_s, pfeat, blw, bl1, rgbv);
  else
    hipLaunchKernelGGL(mv_stats_kernel<16>, grid, dim3(256), 0, st, vw, images, feat, C, xyz, (int)N, vis_in, dd_in, g393, ldg, rgb_feat, vis_ang, valid_s, pfeat, blw, bl1, rgbv);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
