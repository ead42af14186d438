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

// ---------------------------------------------------------------------------------------------------------------
// MFMA version of mv_vis (bf16x3 / bf16 modes): rows = (view, sample) pairs, 32 rows per wave; the decoders are mvd_decode_tile
// (mvdec.h: three-term split-fp16 in the parity mode).  Weights live in LDS in fragment order (copied once per persistent workgroup).
template <bool X3>
__global__ __launch_bounds__(256) void mv_vis_mfma_kernel(const NlViews vw, const float* __restrict__ visf /*(V,vh,vw,32)*/,
                                                          const uint4* __restrict__ dpack, const float* __restrict__ xyz, int N,
                                                          int tiles_per_view, int total_tiles, float* __restrict__ vis_out,
                                                          float* __restrict__ dd_out) {
  __shared__ uint4 sw[MVD_LDS_UINT4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, j = lane & 31;
  if (X3) __builtin_amdgcn_s_setreg(1473, 1);   // MODE.FP16_OVFL: the split-fp16 conversions saturate instead of overflowing to inf
  mvd_load_lds<X3>(sw, dpack, tid, 256);
  __syncthreads();

  for (int tile = blockIdx.x * 4 + wave; tile < total_tiles; tile += gridDim.x * 4) {
    const int v = __builtin_amdgcn_readfirstlane(tile / tiles_per_view);
    const int n = (tile - v * tiles_per_view) * 32 + j;
    const bool live = n < N;
    const int nn = live ? n : N - 1;
    const float X = xyz[3 * (size_t)nn], Y = xyz[3 * (size_t)nn + 1], Z = xyz[3 * (size_t)nn + 2];
    float px, py, depth;
    const bool valid = project_neuray(vw.P2[v], X, Y, Z, vw.Wimg, vw.H, px, py, depth);
    if (__ballot(valid && live) == 0ull) {
      // none of the tile's 32 samples projects into this view: visibility is exactly 0 for all of them and the depth difference
      // is only ever used multiplied by that weight (mv_stats) — skip the decoders (wave-uniform branch)
      if (live && hh == 0) {
        vis_out[(size_t)v * N + n] = 0.f;
        dd_out[(size_t)v * N + n] = 0.f;
      }
      continue;
    }
    float x0[8], x1[8];
    mvd_tap16(visf + (size_t)v * vw.vh * vw.vw * 32 + 8 * hh, vw.vh, vw.vw, vw.Wimg, vw.H, px, py, valid, x0, x1);
    float m0, m1, v0, v1, aw, vs;
    mvd_decode_tile<X3>(sw, lane, x0, x1, m0, m1, v0, v1, vs, aw);
    const float ni = -1.f / vw.near_, fi = -1.f / vw.far_;
    float refd = -1.f / (m0 * (fi - ni) + ni);
    refd = fminf(fmaxf(refd, vw.near_), vw.far_);
    const float dd = fabsf(depth - refd) / (vw.far_ - vw.near_);
    const float dn = (-1.f / fmaxf(depth, 1e-5f) - ni) / (fi - ni);
    const float c0 = (0.5f + 0.5f * tanhf((dn - m0) * v0)) * vs;
    const float c1 = (0.5f + 0.5f * tanhf((dn - m1) * v1)) * vs;
    float vis = (1.f - c0) * aw + (1.f - c1) * (1.f - aw);
    vis = valid ? vis : 0.f;
    if (live && hh == 0) {
      vis_out[(size_t)v * N + n] = vis;
      dd_out[(size_t)v * N + n] = dd;
    }
  }
}

// dpack layout (mvdec.h, MVD_*): fragment-ordered bf16 AND fp16 hi / lo weights of decoder layers 1-2, fp32 biases, lane-ordered layer-3 rows
__global__ void pack_mv_decoder_kernel(const float* __restrict__ dec /* packed VALU layout, 4 x DEC_STRIDE */, uint4* __restrict__ out) {
  unsigned short* o16 = reinterpret_cast<unsigned short*>(out);
  unsigned short* h16 = reinterpret_cast<unsigned short*>(out + MVD_F16);
  float* of = reinterpret_cast<float*>(out + MVD_F32);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  auto f2bf = [](float x) { unsigned int u = __float_as_uint(x); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
  auto f2h = [](float x) { const _Float16 h = (_Float16)x; return __builtin_bit_cast(unsigned short, h); };
  auto h2f = [](unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); };
  if (i < 4096) {          // layer 1: element (q, d, lane, t)
    const int t = i & 7, lane = (i >> 3) & 63, d = (i >> 9) & 3, q = i >> 11;
    const float v = dec[d * DEC_STRIDE + (lane & 31) * 32 + 16 * q + 8 * (lane >> 5) + t];
    const unsigned short h = f2bf(v);
    o16[(size_t)MVD_W1 * 8 + i] = h;
    o16[(size_t)(MVD_W1 + 512) * 8 + i] = f2bf(v - __uint_as_float(((unsigned)h) << 16));
    const unsigned short g = f2h(v);
    h16[(size_t)MVD_W1 * 8 + i] = g;
    h16[(size_t)(MVD_W1 + 512) * 8 + i] = f2h(v - h2f(g));
  } else if (i < 8192) {   // layer 2: element (d, s, lane, t), K permuted to accumulator order
    const int e = i - 4096;
    const int t = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 1, d = e >> 10;
    const int fin = 16 * s + (t & 3) + 8 * (t >> 2) + 4 * (lane >> 5);
    const float v = dec[d * DEC_STRIDE + 1056 + (lane & 31) * 32 + fin];
    const unsigned short h = f2bf(v);
    o16[(size_t)MVD_W2 * 8 + e] = h;
    o16[(size_t)(MVD_W2 + 512) * 8 + e] = f2bf(v - __uint_as_float(((unsigned)h) << 16));
    const unsigned short g = f2h(v);
    h16[(size_t)MVD_W2 * 8 + e] = g;
    h16[(size_t)(MVD_W2 + 512) * 8 + e] = f2h(v - h2f(g));
  } else if (i < 8192 + 128) { const int e = i - 8192; of[e] = dec[(e >> 5) * DEC_STRIDE + 1024 + (e & 31)]; }            // b1
  else if (i < 8192 + 256) { const int e = i - 8192 - 128; of[128 + e] = dec[(e >> 5) * DEC_STRIDE + 2080 + (e & 31)]; }   // b2
  else if (i < 8192 + 512) {                                                                                               // w4p[d][u][hh][r]
    const int e = i - 8192 - 256;
    const int r = e & 15, hh = (e >> 4) & 1, u = (e >> 5) & 1, d = e >> 6;
    const int idx = (r & 3) + 8 * (r >> 2) + 4 * hh;
    of[256 + e] = dec[d * DEC_STRIDE + 2112 + u * 32 + idx];
  } else if (i < 8192 + 520) { const int e = i - 8192 - 512; of[512 + e] = dec[(e >> 1) * DEC_STRIDE + 2176 + (e & 1)]; }   // b4
  // transposed fragments for the input gradient (bf16 hi / lo): element (which, d, s, lane, t), which 0 = W2T, 1 = W1T
  if (i < 8192) {
    unsigned short* t16 = reinterpret_cast<unsigned short*>(out + MVD_T);
    const int which = i >> 12, e = i & 4095;
    const int t = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 1, d = e >> 10;
    const int u = 16 * s + (t & 3) + 8 * (t >> 2) + 4 * (lane >> 5);   // K slot -> unit of the layer's OUTPUT side (accumulator order)
    const int row = lane & 31;
    float v;
    if (which == 0) v = dec[d * DEC_STRIDE + 1056 + u * 32 + row];     // W2[u][row]
    else {
      const int hh = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);   // the register / half whose accumulator row is `row`
      const int chan = r < 8 ? 8 * hh + r : 16 + 8 * hh + (r - 8);
      v = dec[d * DEC_STRIDE + u * 32 + chan];                          // W1[u][chan]
    }
    const unsigned short h = f2bf(v);
    t16[(size_t)which * 1024 * 8 + e] = h;
    t16[(size_t)which * 1024 * 8 + 512 * 8 + e] = f2bf(v - __uint_as_float(((unsigned)h) << 16));
  }
}

__device__ __forceinline__ float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int rli(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// One wave per sample.  Phase A: lane v (< V) does view v's scalar work once — IBRNet projection, tap offsets / weights for
// the feature map and the image (validity folded into zero weights + clamped offsets, so phase-B loads are unconditional),
// view-angle features, visibility weight.  Phase B: unrolled loop over views; the per-view scalars come from
// v_readlane with a constant lane (-> SGPRs), lanes span channels (64 lanes x 3 floats = one 768-B texel row per tap).
template <int VT, bool V4, bool EXACT>   // EXACT: the frame has exactly VT views (no per-view guards)
__global__ __launch_bounds__(256, VT <= 4 ? 4 : (VT <= 10 ? 3 : 2)) void mv_stats_kernel(const NlViews vw, const float* __restrict__ viewsdev /*[16][12] P1, then [16][3] cam*/,
                                                       const float* __restrict__ images /*(V,3,H,W)*/,
                                                       const float* __restrict__ feat /*(V,h,w,C)*/, int C,
                                                       const float* __restrict__ xyz, int N,
                                                       const float* __restrict__ vis_in, const float* __restrict__ dd_in,
                                                       float* __restrict__ g393, int ldg, float* __restrict__ rgb_feat /*(N*V,196) or null*/,
                                                       float* __restrict__ vis_ang /*(N*V,8) or null*/, int* __restrict__ valid_s,
                                                       const float* __restrict__ pfeat /*(V,h,w,32) blend-projected feature map*/,
                                                       const float* __restrict__ blw /*[32][8] = W[rgb3|vis1|ang4], then bias[32]*/,
                                                       float* __restrict__ bl1 /*(N*V,32) or null*/, float* __restrict__ rgbv /*(N*V,4) or null*/) {
  const int lane = threadIdx.x & 63;
  const int n = nl_xcd_block() * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int V = EXACT ? VT : vw.V;
  const float X = xyz[3 * (size_t)n], Y = xyz[3 * (size_t)n + 1], Z = xyz[3 * (size_t)n + 2];
  const int F = C + 3;

  // ---------------------------------------------------------------- phase A (lane = view)
  const int vl = lane < V ? lane : 0;
  const bool vact = lane < V;
  float a_fw[4], a_iw[4];   // tap weights (nw, ne, sw, se), zero where the tap is outside
  unsigned a_fo, a_io;      // packed clamped tap offsets (pack_taps): feature map in texels, image in pixels
  float a_ang[4], a_vis, a_dd, a_wgt;
  int cnt1;
  {
    const float4 p0 = *(const float4*)(viewsdev + 12 * vl), p1 = *(const float4*)(viewsdev + 12 * vl + 4), p2 = *(const float4*)(viewsdev + 12 * vl + 8);
    const float cx = fmaf(p0.z, Z, fmaf(p0.y, Y, p0.x * X)) + p0.w;
    const float cy = fmaf(p1.z, Z, fmaf(p1.y, Y, p1.x * X)) + p1.w;
    const float cz = fmaf(p2.z, Z, fmaf(p2.y, Y, p2.x * X)) + p2.w;
    const float zc = fmaxf(cz, 1e-8f);
    float px = cx / zc, py = cy / zc;
    px = fminf(fmaxf(px, -1e6f), 1e6f);
    py = fminf(fmaxf(py, -1e6f), 1e6f);
    const bool m1 = vact && (px <= (float)vw.Wimg - 1.f) && (px >= 0.f) && (py <= (float)vw.H - 1.f) && (py >= 0.f) && (cz > 0.f);
    cnt1 = __popcll(__ballot(m1));
    const float xn = 2.f * px / (float)(vw.Wimg - 1) - 1.f;
    const float yn = 2.f * py / (float)(vw.H - 1) - 1.f;
    {
      const Taps t = make_taps<true, false>(xn, yn, vw.w, vw.h);
      a_fo = pack_taps(t, vw.w, vw.h);
      a_fw[0] = (t.mn && t.mw) ? t.nw : 0.f; a_fw[1] = (t.mn && t.me) ? t.ne : 0.f;
      a_fw[2] = (t.ms && t.mw) ? t.sw : 0.f; a_fw[3] = (t.ms && t.me) ? t.se : 0.f;
    }
    {
      const Taps t = make_taps<true, false>(xn, yn, vw.Wimg, vw.H);
      a_io = pack_taps(t, vw.Wimg, vw.H);
      a_iw[0] = (t.mn && t.mw) ? t.nw : 0.f; a_iw[1] = (t.mn && t.me) ? t.ne : 0.f;
      a_iw[2] = (t.ms && t.mw) ? t.sw : 0.f; a_iw[3] = (t.ms && t.me) ? t.se : 0.f;
    }
    // view-angle features (ibrnet.py:144-167)
    float qc0 = vw.qcam[0], qc1 = vw.qcam[1], qc2 = vw.qcam[2];
    if (vw.qrows) { const float* qr = vw.qrows + 3 * (size_t)(n / vw.qS); qc0 = qr[0]; qc1 = qr[1]; qc2 = qr[2]; }
    float tq[3] = {qc0 - X, qc1 - Y, qc2 - Z};
    const float nq = sqrtf(tq[0] * tq[0] + tq[1] * tq[1] + tq[2] * tq[2]) + 1e-6f;
    tq[0] /= nq; tq[1] /= nq; tq[2] /= nq;
    float tt[3] = {viewsdev[192 + 3 * vl] - X, viewsdev[192 + 3 * vl + 1] - Y, viewsdev[192 + 3 * vl + 2] - Z};
    const float nt = sqrtf(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]) + 1e-6f;
    tt[0] /= nt; tt[1] /= nt; tt[2] /= nt;
    const float df[3] = {tq[0] - tt[0], tq[1] - tt[1], tq[2] - tt[2]};
    const float nd = fmaxf(sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]), 1e-6f);
    a_ang[0] = df[0] / nd; a_ang[1] = df[1] / nd; a_ang[2] = df[2] / nd;
    a_ang[3] = tq[0] * tt[0] + tq[1] * tt[1] + tq[2] * tt[2];
    a_vis = vact ? vis_in[(size_t)vl * N + n] : 0.f;
    a_dd = vact ? dd_in[(size_t)vl * N + n] : 0.f;
    // weight = vis / (sum_v vis + 1e-8): sequential sum over views like the reference's reduction
    float vsum = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) vsum += v < V ? rl(a_vis, v) : 0.f;
    a_wgt = a_vis / (vsum + 1e-8f);
  }

  // ---------------------------------------------------------------- phase B (lanes = channels)
  float bwr[8], bbias = 0.f;   // this lane's row (j = lane < 32) of the small blend-layer weights
#pragma unroll
  for (int i = 0; i < 8; ++i) bwr[i] = (bl1 && lane < 32) ? blw[lane * 8 + i] : 0.f;
  if (bl1 && lane < 32) bbias = blw[256 + lane];
  float xv[VT][5];   // [0..3]: feature channels (V4: 4*lane+j; else lane+64j for j<3), [4]: rgb plane `lane` (lanes 0..2)
  const int lch = lane < 3 ? lane : 0;
  // The view loop is software-pipelined by hand: all 12 taps of view v+1 (feature map, image, blend-projected map) are
  // issued before view v is reduced, so a wave has one view of loads in flight while it computes (left to itself the
  // compiler emits load -> wait -> use three times per view: 30 dependent memory round trips per sample).
  struct ViewTaps { float4 f[4]; float fs[3][4]; float im[4]; float pf[4]; };
  // lanes past the last channel group re-read the last one (their results are never stored): no exec-masked loads, no zero fills
  const unsigned lane4 = V4 ? (unsigned)min(4 * lane, C - 4) : 0u, lane31 = (unsigned)lane & 31u, lplane = (unsigned)lch * (unsigned)(vw.H * vw.Wimg);
  auto issue = [&](int v) __attribute__((always_inline)) {
    ViewTaps t;
    int o[4], oi[4];
    unpack_taps((unsigned)rli((int)a_fo, v), vw.w, o);
    unpack_taps((unsigned)rli((int)a_io, v), vw.Wimg, oi);
    const float* fb = feat + (size_t)v * vw.h * vw.w * C;
    if constexpr (V4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) t.f[k] = *(const float4*)((fb + (size_t)o[k] * C) + lane4);   // uniform base + 32-bit lane offset
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int ch = lane + 64 * j;
#pragma unroll
        for (int k = 0; k < 4; ++k) t.fs[j][k] = ch < C ? fb[(size_t)o[k] * C + ch] : 0.f;
      }
    }
    const float* ib = images + (size_t)v * 3 * vw.H * vw.Wimg;
#pragma unroll
    for (int k = 0; k < 4; ++k) t.im[k] = (ib + oi[k])[lplane];
    if (bl1) {
      const float* pb = pfeat + (size_t)v * vw.h * vw.w * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) t.pf[k] = (pb + (size_t)o[k] * 32)[lane31];
    }
    return t;
  };
  auto finish = [&](int v, const ViewTaps& t) __attribute__((always_inline)) {
    const float w0 = rl(a_fw[0], v), w1 = rl(a_fw[1], v), w2 = rl(a_fw[2], v), w3 = rl(a_fw[3], v);
    if constexpr (V4) {
      // explicit fma chains: the file is built with -ffp-contract=off (the bit-exact kernels need it), which would cost 7
      // instructions per bilinear tap sum instead of 4
      xv[v][0] = fmaf(t.f[3].x, w3, fmaf(t.f[2].x, w2, fmaf(t.f[1].x, w1, t.f[0].x * w0)));
      xv[v][1] = fmaf(t.f[3].y, w3, fmaf(t.f[2].y, w2, fmaf(t.f[1].y, w1, t.f[0].y * w0)));
      xv[v][2] = fmaf(t.f[3].z, w3, fmaf(t.f[2].z, w2, fmaf(t.f[1].z, w1, t.f[0].z * w0)));
      xv[v][3] = fmaf(t.f[3].w, w3, fmaf(t.f[2].w, w2, fmaf(t.f[1].w, w1, t.f[0].w * w0)));
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) xv[v][j] = fmaf(t.fs[j][3], w3, fmaf(t.fs[j][2], w2, fmaf(t.fs[j][1], w1, t.fs[j][0] * w0)));
    }
    {
      const float i0 = rl(a_iw[0], v), i1 = rl(a_iw[1], v), i2 = rl(a_iw[2], v), i3 = rl(a_iw[3], v);
      const float val = fmaf(t.im[3], i3, fmaf(t.im[2], i2, fmaf(t.im[1], i1, t.im[0] * i0)));
      xv[v][4] = lane < 3 ? val : 0.f;
    }
    const float s_vis = rl(a_vis, v);
    if (bl1) {
      // colour-blend layer 1, per-(sample, view) part, by linearity of the bilinear tap (model.py:532-535):
      //   W[:, feat] . bilinear(featmap) == bilinear(W[:, feat] . featmap); plus rgb / visibility / angle columns + bias
      const float pv = fmaf(t.pf[3], w3, fmaf(t.pf[2], w2, fmaf(t.pf[1], w1, t.pf[0] * w0)));
      const float r = rl(xv[v][4], 0), g = rl(xv[v][4], 1), bb = rl(xv[v][4], 2);
      float o = pv + bbias;
      o = fmaf(bwr[0], r, o); o = fmaf(bwr[1], g, o); o = fmaf(bwr[2], bb, o);
      o = fmaf(bwr[3], s_vis, o);
      o = fmaf(bwr[4], rl(a_ang[0], v), o); o = fmaf(bwr[5], rl(a_ang[1], v), o);
      o = fmaf(bwr[6], rl(a_ang[2], v), o); o = fmaf(bwr[7], rl(a_ang[3], v), o);
      if (lane < 32) bl1[((size_t)n * V + v) * 32 + lane] = o;
      if (lane < 4) rgbv[((size_t)n * V + v) * 4 + lane] = lane < 3 ? xv[v][4] : s_vis;
    }
    if (rgb_feat) {   // stage API only: materialise the raw multi-view projection and [vis, angle]
      float* row = rgb_feat + ((size_t)n * V + v) * NL_FPAD;
#pragma unroll
      for (int j = 0; j < (V4 ? 4 : 3); ++j) {
        const int ch = V4 ? 4 * lane + j : lane + 64 * j;
        if (ch < C) row[3 + ch] = xv[v][j];
      }
      if (lane < 3) row[lane] = xv[v][4];
      if (lane == 3) row[F] = 0.f;
      if (lane == 0 && vis_ang) {
        float* va = vis_ang + ((size_t)n * V + v) * 8;
        *(float4*)va = make_float4(s_vis, rl(a_ang[0], v), rl(a_ang[1], v), rl(a_ang[2], v));
        *(float4*)(va + 4) = make_float4(rl(a_ang[3], v), 0.f, 0.f, 0.f);
      }
    }
  };
#pragma unroll
  for (int v = 0; v < VT; ++v) xv[v][0] = xv[v][1] = xv[v][2] = xv[v][3] = xv[v][4] = 0.f;
  {
    // two views of taps in flight ahead of the one being reduced (the kernel waits on memory, not on the vector unit)
    ViewTaps cur = issue(0);
    ViewTaps nxt = cur;
    if (1 < VT && 1 < V) nxt = issue(1);
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (v < V) {
        ViewTaps nx2 = nxt;
        if (v + 2 < VT && v + 2 < V) nx2 = issue(v + 2);
        finish(v, cur);
        cur = nxt;
        nxt = nx2;
      }
    }
  }
  // visibility-weighted mean / variance over views (ibrnet.py:8-12)
  float wg[VT];
#pragma unroll
  for (int v = 0; v < VT; ++v) wg[v] = v < V ? rl(a_wgt, v) : 0.f;
  float* g = g393 + (size_t)n * ldg;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (!V4 && j == 3) continue;   // the scalar layout has three feature slots
    float mean = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) mean = fmaf(xv[v][j], wg[v], mean);
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { float d = xv[v][j] - mean; var = fmaf(wg[v] * d, d, var); }
    int pos = -1;
    if (j < 4) { const int ch = V4 ? 4 * lane + j : lane + 64 * j; if (ch < C) pos = 3 + ch; }
    else if (lane < 3) pos = lane;
    if (pos >= 0) { g[pos] = mean; g[F + pos] = var; }
  }
  if (lane == 0) {
    float mean = 0.f, wsum = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { if (v < V) { mean += rl(a_dd, v) * wg[v]; wsum += wg[v]; } }
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { if (v < V) { float d = rl(a_dd, v) - mean; var += wg[v] * (d * d); } }
    g[2 * F] = mean;
    g[2 * F + 1] = var;
    g[2 * F + 2] = wsum / (float)V;
    for (int p = 2 * F + 3; p < ldg; ++p) g[p] = 0.f;
    valid_s[n] = cnt1 > 1 ? 1 : 0;
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// mv_stats8_kernel: the same computation as mv_stats_kernel with EIGHT CONSECUTIVE SAMPLES PER WAVE (lane = 8 * sample + j).
// Why: mv_stats_kernel gives a whole wave to one sample (lanes = channels), so every bilinear tap is its own 768-byte row fetch —
// 30 KB through the vector L1 per sample, 16 GB per batch, and knock-outs show that this data path (not vector instructions, not
// latency alone) is what the kernel waits on.  Consecutive samples of a ray project onto almost the same texels of a support view
// (the feature map has a quarter of the image resolution), so here one load instruction serves the same 128-byte channel chunk of
// the taps of 8 neighbouring samples: lanes that hit the same texel share its cache line, and the per-(sample, view) scalar work
// (projection, tap weights, angle features) is done by one lane instead of being replicated over a wave.
//   phase A   lane (s, j) handles view j (and j + 8): projection, packed tap offsets + weights for the feature map and the image,
//             view-angle features, visibility -> one 20-dword slot per (sample, view) in LDS (wave-private, no block barrier)
//   phase B   for every 32-channel chunk: lane (s, j) owns channels 32 i + 4 j .. + 3 of sample s: unrolled loop over the views
//             (taps of two views in flight), then the visibility-weighted mean / variance over the views
//   image     lanes j < 3 own one colour plane each; the tapped colours go to the LDS slot for the blend layer
//   blend     (model.py:532-535, per-(sample, view) part) lane (s, j) owns 4 of the 32 hidden units
constexpr int MS8_SLOT = 20;   // 0 feature taps (packed), 1 image taps (packed), 2-5 feature tap weights, 6-9 image tap weights,
                               // 10-13 angle features, 14 visibility, 15 depth difference, 16 weight, 17-19 tapped rgb

// (The view loops keep their run-time `v < V` guards on purpose: a specialisation for V == VT lets the compiler hoist ~40 more
// registers of loop-invariant addresses, which costs the fourth wave per SIMD and 0.7 ms — measured.)
template <int VT>
__global__ __launch_bounds__(256) void mv_stats8_kernel(const NlViews vw, const float* __restrict__ viewsdev, const float* __restrict__ images,
                                                        const float* __restrict__ feat, int C, const float* __restrict__ xyz, int N,
                                                        const float* __restrict__ vis_in, const float* __restrict__ dd_in,
                                                        float* __restrict__ g393, int ldg, int* __restrict__ valid_s,
                                                        const float* __restrict__ pfeat, const float* __restrict__ blw,
                                                        float* __restrict__ bl1, float* __restrict__ rgbv) {
  extern __shared__ float ms8_lds[];   // [4 waves][8 samples][VT][MS8_SLOT]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane >> 3, j = lane & 7;
  const int V = vw.V;
  const int n0 = (nl_xcd_block() * 4 + wave) * 8;
  if (n0 >= N) return;
  const int n = n0 + s;
  const bool live = n < N;
  const int nn = live ? n : N - 1;
  float* slot0 = ms8_lds + (size_t)((wave * 8 + s) * VT) * MS8_SLOT;
  const int F = C + 3;

  // ---------------------------------------------------------------- phase A (lane = (sample, view))
  int cnt1 = 0;
  {
    const float X = xyz[3 * (size_t)nn], Y = xyz[3 * (size_t)nn + 1], Z = xyz[3 * (size_t)nn + 2];
    float qc0 = vw.qcam[0], qc1 = vw.qcam[1], qc2 = vw.qcam[2];
    if (vw.qrows) { const float* qr = vw.qrows + 3 * (size_t)(nn / vw.qS); qc0 = qr[0]; qc1 = qr[1]; qc2 = qr[2]; }
    float tq[3] = {qc0 - X, qc1 - Y, qc2 - Z};
    const float rq = 1.f / (sqrtf(tq[0] * tq[0] + tq[1] * tq[1] + tq[2] * tq[2]) + 1e-6f);   // one division per vector (the angle features are not bit-compared)
    tq[0] *= rq; tq[1] *= rq; tq[2] *= rq;
#pragma unroll
    for (int vv = 0; vv < VT; vv += 8) {
      const int v = vv + j;
      const bool vact = v < V;
      const int vl = vact ? v : 0;
      const float4 p0 = *(const float4*)(viewsdev + 12 * vl), p1 = *(const float4*)(viewsdev + 12 * vl + 4), p2 = *(const float4*)(viewsdev + 12 * vl + 8);
      const float cx = fmaf(p0.z, Z, fmaf(p0.y, Y, p0.x * X)) + p0.w;
      const float cy = fmaf(p1.z, Z, fmaf(p1.y, Y, p1.x * X)) + p1.w;
      const float cz = fmaf(p2.z, Z, fmaf(p2.y, Y, p2.x * X)) + p2.w;
      const float zc = fmaxf(cz, 1e-8f);
      float px = cx / zc, py = cy / zc;
      px = fminf(fmaxf(px, -1e6f), 1e6f);
      py = fminf(fmaxf(py, -1e6f), 1e6f);
      const bool m1 = vact && (px <= (float)vw.Wimg - 1.f) && (px >= 0.f) && (py <= (float)vw.H - 1.f) && (py >= 0.f) && (cz > 0.f);
      const unsigned long long bm = __ballot(m1);
      cnt1 += __popc((unsigned)(bm >> (8 * s)) & 0xffu);
      const float xn = 2.f * px / (float)(vw.Wimg - 1) - 1.f;
      const float yn = 2.f * py / (float)(vw.H - 1) - 1.f;
      const Taps tf = make_taps<true, false>(xn, yn, vw.w, vw.h);
      const Taps ti = make_taps<true, false>(xn, yn, vw.Wimg, vw.H);
      // view-angle features (ibrnet.py:144-167)
      float tt[3] = {viewsdev[192 + 3 * vl] - X, viewsdev[192 + 3 * vl + 1] - Y, viewsdev[192 + 3 * vl + 2] - Z};
      const float rt = 1.f / (sqrtf(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]) + 1e-6f);
      tt[0] *= rt; tt[1] *= rt; tt[2] *= rt;
      const float df[3] = {tq[0] - tt[0], tq[1] - tt[1], tq[2] - tt[2]};
      const float rd = 1.f / fmaxf(sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]), 1e-6f);
      if (v < VT) {
        float* sl = slot0 + v * MS8_SLOT;
        sl[0] = __uint_as_float(pack_taps(tf, vw.w, vw.h));
        sl[1] = __uint_as_float(pack_taps(ti, vw.Wimg, vw.H));
        *(float4*)(sl + 2) = make_float4((tf.mn && tf.mw) ? tf.nw : 0.f, (tf.mn && tf.me) ? tf.ne : 0.f, (tf.ms && tf.mw) ? tf.sw : 0.f,
                                         (tf.ms && tf.me) ? tf.se : 0.f);
        *(float4*)(sl + 6) = make_float4((ti.mn && ti.mw) ? ti.nw : 0.f, (ti.mn && ti.me) ? ti.ne : 0.f, (ti.ms && ti.mw) ? ti.sw : 0.f,
                                         (ti.ms && ti.me) ? ti.se : 0.f);
        *(float4*)(sl + 10) = make_float4(df[0] * rd, df[1] * rd, df[2] * rd, tq[0] * tt[0] + tq[1] * tt[1] + tq[2] * tt[2]);
        sl[14] = vact ? vis_in[(size_t)vl * N + nn] : 0.f;
        sl[15] = vact ? dd_in[(size_t)vl * N + nn] : 0.f;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // weight = vis / (sum_v vis + 1e-8): sequential sum over views like the reference's reduction
  float wg[VT];
  {
    float vsum = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) vsum += v < V ? slot0[v * MS8_SLOT + 14] : 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) wg[v] = v < V ? slot0[v * MS8_SLOT + 14] / (vsum + 1e-8f) : 0.f;
  }
  // Views in which none of the wave's 8 samples is visible (weight exactly 0 for all of them) contribute nothing to any statistic
  // and their blend logits are masked out (vis == 0): their feature-map taps are not fetched at all (wave-uniform branches).  The
  // colour taps stay: a sample that no view sees is blended as the plain mean of the tapped colours (softmax of equal logits).
  unsigned vmask = 0;
#pragma unroll
  for (int v = 0; v < VT; ++v)
    if (v < V && __ballot(slot0[v * MS8_SLOT + 14] != 0.f) != 0ull) vmask |= 1u << v;
  auto active = [&](int v) { return ((vmask >> v) & 1u) != 0u; };
  float* g = g393 + (size_t)nn * ldg;
  struct Tap4 { float4 t[4]; };
  const size_t fmap = (size_t)vw.h * vw.w;

  // ---------------------------------------------------------------- phase B: feature chunks of 32 channels
  const int nchunk = (C + 31) >> 5;
  for (int i = 0; i < nchunk; ++i) {
    const int ch = 32 * i + 4 * j;
    const bool chv = ch < C;
    const unsigned cho = chv ? (unsigned)ch : 0u;
    auto issue = [&](int v) __attribute__((always_inline)) {
      Tap4 r;
      if (!active(v)) { r.t[0] = r.t[1] = r.t[2] = r.t[3] = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
      int o[4];
      unpack_taps(__float_as_uint(slot0[v * MS8_SLOT]), vw.w, o);
      const float* fb = feat + (size_t)v * fmap * C;   // uniform base + 32-bit lane offset
#pragma unroll
      for (int k = 0; k < 4; ++k) r.t[k] = *(const float4*)(fb + ((unsigned)o[k] * (unsigned)C + cho));
      return r;
    };
    float xv[VT][4];
#pragma unroll
    for (int v = 0; v < VT; ++v) xv[v][0] = xv[v][1] = xv[v][2] = xv[v][3] = 0.f;
    Tap4 cur = issue(0), nxt = cur;
    if (1 < VT && 1 < V) nxt = issue(1);
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (v < V) {
        Tap4 nx2 = nxt;
        if (v + 2 < VT && v + 2 < V) nx2 = issue(v + 2);
        if (active(v)) {
          const float4 w = *(const float4*)(slot0 + v * MS8_SLOT + 2);
          xv[v][0] = fmaf(cur.t[3].x, w.w, fmaf(cur.t[2].x, w.z, fmaf(cur.t[1].x, w.y, cur.t[0].x * w.x)));
          xv[v][1] = fmaf(cur.t[3].y, w.w, fmaf(cur.t[2].y, w.z, fmaf(cur.t[1].y, w.y, cur.t[0].y * w.x)));
          xv[v][2] = fmaf(cur.t[3].z, w.w, fmaf(cur.t[2].z, w.z, fmaf(cur.t[1].z, w.y, cur.t[0].z * w.x)));
          xv[v][3] = fmaf(cur.t[3].w, w.w, fmaf(cur.t[2].w, w.z, fmaf(cur.t[1].w, w.y, cur.t[0].w * w.x)));
        }
        cur = nxt;
        nxt = nx2;
      }
    }
    // visibility-weighted mean / variance over views (ibrnet.py:8-12)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float mean = 0.f;
#pragma unroll
      for (int v = 0; v < VT; ++v) mean = fmaf(xv[v][c], wg[v], mean);
      float var = 0.f;
#pragma unroll
      for (int v = 0; v < VT; ++v) { const float d = xv[v][c] - mean; var = fmaf(wg[v] * d, d, var); }
      if (live && chv) { g[3 + ch + c] = mean; g[F + 3 + ch + c] = var; }
    }
  }

  // ---------------------------------------------------------------- image taps (lanes j < 3: one colour plane each)
  {
    const unsigned plane = (unsigned)(j < 3 ? j : 0) * (unsigned)(vw.H * vw.Wimg);
    float xi[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      xi[v] = 0.f;
      if (v < V) {   // also for invisible views: when no view at all sees a sample the blend falls back to the plain mean of the colours
        int o[4];
        unpack_taps(__float_as_uint(slot0[v * MS8_SLOT + 1]), vw.Wimg, o);
        const float* ib = images + (size_t)v * 3 * vw.H * vw.Wimg;
        const float4 w = *(const float4*)(slot0 + v * MS8_SLOT + 6);
        const float a = ib[plane + (unsigned)o[0]], b = ib[plane + (unsigned)o[1]], c = ib[plane + (unsigned)o[2]], d = ib[plane + (unsigned)o[3]];
        xi[v] = fmaf(d, w.w, fmaf(c, w.z, fmaf(b, w.y, a * w.x)));
        if (j < 3) slot0[v * MS8_SLOT + 17 + j] = xi[v];
      }
    }
    float mean = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) mean = fmaf(xi[v], wg[v], mean);
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { const float d = xi[v] - mean; var = fmaf(wg[v] * d, d, var); }
    if (live && j < 3) { g[j] = mean; g[F + j] = var; }
  }
  __builtin_amdgcn_wave_barrier();

  // ---------------------------------------------------------------- colour-blend layer 1, per-(sample, view) part
  if (bl1) {
    float bw[4][8], bb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 lo = *(const float4*)(blw + (4 * j + c) * 8), hi = *(const float4*)(blw + (4 * j + c) * 8 + 4);
      bw[c][0] = lo.x; bw[c][1] = lo.y; bw[c][2] = lo.z; bw[c][3] = lo.w; bw[c][4] = hi.x; bw[c][5] = hi.y; bw[c][6] = hi.z; bw[c][7] = hi.w;
      bb[c] = blw[256 + 4 * j + c];
    }
    auto issue = [&](int v) __attribute__((always_inline)) {
      Tap4 r;
      if (!active(v)) { r.t[0] = r.t[1] = r.t[2] = r.t[3] = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
      int o[4];
      unpack_taps(__float_as_uint(slot0[v * MS8_SLOT]), vw.w, o);
      const float* pb = pfeat + (size_t)v * fmap * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) r.t[k] = *(const float4*)(pb + ((unsigned)o[k] * 32u + 4u * (unsigned)j));
      return r;
    };
    Tap4 cur = issue(0);
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (v < V) {
        Tap4 nxt = cur;
        if (v + 1 < VT && v + 1 < V) nxt = issue(v + 1);
        if (!active(v)) {   // logit masked out downstream (vis == 0 for all 8 samples): finite placeholder, no taps, no arithmetic
          if (live) {
            *(float4*)(bl1 + ((size_t)n * V + v) * 32 + 4 * j) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j == 0) *(float4*)(rgbv + ((size_t)n * V + v) * 4) = make_float4(slot0[v * MS8_SLOT + 17], slot0[v * MS8_SLOT + 18], slot0[v * MS8_SLOT + 19], 0.f);
          }
          cur = nxt;
          continue;
        }
        const float* sl = slot0 + v * MS8_SLOT;
        const float4 w = *(const float4*)(sl + 2), ang = *(const float4*)(sl + 10);
        const float s_vis = sl[14], r = sl[17], gg = sl[18], b = sl[19];
        const float pv[4] = {fmaf(cur.t[3].x, w.w, fmaf(cur.t[2].x, w.z, fmaf(cur.t[1].x, w.y, cur.t[0].x * w.x))),
                             fmaf(cur.t[3].y, w.w, fmaf(cur.t[2].y, w.z, fmaf(cur.t[1].y, w.y, cur.t[0].y * w.x))),
                             fmaf(cur.t[3].z, w.w, fmaf(cur.t[2].z, w.z, fmaf(cur.t[1].z, w.y, cur.t[0].z * w.x))),
                             fmaf(cur.t[3].w, w.w, fmaf(cur.t[2].w, w.z, fmaf(cur.t[1].w, w.y, cur.t[0].w * w.x)))};
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float a = pv[c] + bb[c];
          a = fmaf(bw[c][0], r, a); a = fmaf(bw[c][1], gg, a); a = fmaf(bw[c][2], b, a);
          a = fmaf(bw[c][3], s_vis, a);
          a = fmaf(bw[c][4], ang.x, a); a = fmaf(bw[c][5], ang.y, a); a = fmaf(bw[c][6], ang.z, a); a = fmaf(bw[c][7], ang.w, a);
          o[c] = a;
        }
        if (live) {
          *(float4*)(bl1 + ((size_t)n * V + v) * 32 + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
          if (j == 0) *(float4*)(rgbv + ((size_t)n * V + v) * 4) = make_float4(r, gg, b, s_vis);
        }
        cur = nxt;
      }
    }
  }

  // ---------------------------------------------------------------- depth-difference statistics, padding, valid flag
  if (live && j == 0) {
    float mean = 0.f, wsum = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { if (v < V) { mean += slot0[v * MS8_SLOT + 15] * wg[v]; wsum += wg[v]; } }
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < VT; ++v) { if (v < V) { const float d = slot0[v * MS8_SLOT + 15] - mean; var += wg[v] * (d * d); } }
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

size_t nl_mv_decoder_pack_bytes() { return (size_t)MVD_PACK_UINT4 * 16; }

int nl_pack_mv_decoder(const float* dec_valu_layout, void* out, hipStream_t st) {
  hipLaunchKernelGGL(pack_mv_decoder_kernel, dim3((8192 + 520 + 255) / 256), dim3(256), 0, st, dec_valu_layout, (uint4*)out);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_mv_vis_mfma(const NlViews& vw, const float* visf_hwc, const void* dpack, const float* xyz, int64_t N, float* vis_out,
                          float* dd_out, bool x3, hipStream_t st) {
  if (N <= 0) return NL_OK;
  const int tpv = (int)nl_cdiv(N, 32), total = tpv * vw.V;
#ifndef MV_VIS_MAX_BLOCKS
#define MV_VIS_MAX_BLOCKS 2048
#endif
  const int blocks = (int)(nl_cdiv(total, 4) < MV_VIS_MAX_BLOCKS ? nl_cdiv(total, 4) : MV_VIS_MAX_BLOCKS);
  if (x3) hipLaunchKernelGGL(mv_vis_mfma_kernel<true>, dim3(blocks), dim3(256), 0, st, vw, visf_hwc, (const uint4*)dpack, xyz, (int)N, tpv, total, vis_out, dd_out);
  else hipLaunchKernelGGL(mv_vis_mfma_kernel<false>, dim3(blocks), dim3(256), 0, st, vw, visf_hwc, (const uint4*)dpack, xyz, (int)N, tpv, total, vis_out, dd_out);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_mv_stats(const NlViews& vw, const float* viewsdev, const float* images, const float* feat, int C, const float* xyz, int64_t N,
                       const float* vis_in, const float* dd_in, float* g393, int ldg, float* rgb_feat, float* vis_ang,
                       int* valid_s, const float* pfeat, const float* blw, float* bl1, float* rgbv, hipStream_t st) {
  if (N <= 0) return NL_OK;
  if (C > 192) return NL_ERR_UNSUPPORTED;
  const bool v4 = (C % 4 == 0) && ((((size_t)feat) & 15) == 0);   // 16-B channel groups
#ifdef NERFLOC_DEBUG_SWITCHES
  static const bool force_wave = getenv("NERFLOC_MVSTATS_WAVE") != nullptr;   // A/B switch (debug builds only): one sample per wave everywhere
#else
  const bool force_wave = false;
#endif
  if (v4 && !rgb_feat && !vis_ang && !force_wave) {   // everything but the stage API: eight samples per wave
    dim3 grid8(nl_xcd_grid(nl_cdiv(N, 32)));
#define NL_MS8(VT) hipLaunchKernelGGL((mv_stats8_kernel<VT>), grid8, dim3(256), sizeof(float) * 4 * 8 * VT * MS8_SLOT, st, vw, viewsdev, images, feat, C, \
                                      xyz, (int)N, vis_in, dd_in, g393, ldg, valid_s, pfeat, blw, bl1, rgbv)
    if (vw.V <= 4) NL_MS8(4); else if (vw.V <= 8) NL_MS8(8); else if (vw.V <= 10) NL_MS8(10); else NL_MS8(16);
#undef NL_MS8
    NL_LAUNCH_CHECK();
    return NL_OK;
  }
  dim3 grid(nl_xcd_grid(nl_cdiv(N, 4)));
  const bool ex = vw.V == 4 || vw.V == 8 || vw.V == 10 || vw.V == 16;   // the frame has exactly the bucket's view count
#define NL_MS1(VT, V4, EX)                                                                                                              \
  hipLaunchKernelGGL((mv_stats_kernel<VT, V4, EX>), grid, dim3(256), 0, st, vw, viewsdev, images, feat, C, xyz, (int)N, vis_in, dd_in, \
                     g393, ldg, rgb_feat, vis_ang, valid_s, pfeat, blw, bl1, rgbv)
#define NL_MS1_BUCKET(VT)                 \
  do {                                    \
    if (v4 && ex) NL_MS1(VT, true, true); \
    else if (v4) NL_MS1(VT, true, false); \
    else NL_MS1(VT, false, false);        \
  } while (0)
  if (vw.V <= 4) NL_MS1_BUCKET(4);
  else if (vw.V <= 8) NL_MS1_BUCKET(8);
  else if (vw.V <= 10) NL_MS1_BUCKET(10);
  else NL_MS1_BUCKET(16);
#undef NL_MS1_BUCKET
#undef NL_MS1
  NL_LAUNCH_CHECK();
  return NL_OK;
}
