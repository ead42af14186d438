// Device helpers shared by mvagg.hip and hier.hip: grid_sample tap arithmetic and the NeuRay mixture-of-logistics
// decoders (visibility_decoder.py:64-107) evaluated per lane with wave-uniform (scalar-loaded) weights.
#pragma once
#include "common.h"

namespace nlmv {

struct Taps {
  int x0, y0;
  float nw, ne, sw, se;
  bool mw, me, mn, ms;  // tap validity (west/east columns, north/south rows)
};

// ATen vectorised CPU grid_sample arithmetic (GridSamplerKernel.cpp: ComputeLocation + compute_interp_params)
template <bool ALIGN, bool BORDER>
__device__ __forceinline__ Taps make_taps(float xn, float yn, int Wm, int Hm) {
  float ix, iy;
  if (ALIGN) {
    ix = (xn + 1.f) * ((float)(Wm - 1) / 2.f);
    iy = (yn + 1.f) * ((float)(Hm - 1) / 2.f);
  } else {
    ix = (xn + 1.f) * ((float)Wm / 2.f) - 0.5f;
    iy = (yn + 1.f) * ((float)Hm / 2.f) - 0.5f;
  }
  if (BORDER) {
    ix = fminf((float)(Wm - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(Hm - 1), fmaxf(iy, 0.f));
  }
  float xw = floorf(ix), yn0 = floorf(iy);
  float w = ix - xw, e = 1.f - w, n = iy - yn0, s = 1.f - n;
  Taps t;
  t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
  // keep the int conversion safe for the +-1e6 clamped coordinates
  xw = fminf(fmaxf(xw, -2.f), (float)Wm + 1.f);
  yn0 = fminf(fmaxf(yn0, -2.f), (float)Hm + 1.f);
  t.x0 = (int)xw; t.y0 = (int)yn0;
  t.mw = t.x0 >= 0 && t.x0 < Wm;
  t.me = t.x0 + 1 >= 0 && t.x0 + 1 < Wm;
  t.mn = t.y0 >= 0 && t.y0 < Hm;
  t.ms = t.y0 + 1 >= 0 && t.y0 + 1 < Hm;
  return t;
}

// packed decoder weights: 4 x { W0[32][32], b0[32], W2[32][32], b2[32], W4[2][32] (row 1 zero if nout=1), b4[2] }
constexpr int DEC_STRIDE = 1024 + 32 + 1024 + 32 + 64 + 2;

__device__ __forceinline__ void decoder(const float* __restrict__ w, const float (&x)[32], float& o0, float& o1) {
  float h1[32], h2[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    float a = w[1024 + j];  // b0[j]
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
  float a0 = w4[64], a1 = w4[65];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    a0 = fmaf(w4[i], h2[i], a0);
    a1 = fmaf(w4[32 + i], h2[i], a1);
  }
  o0 = a0;
  o1 = a1;
}


// NeuRay-convention projection of one point into one view (depth_fusion.py:78-126): returns validity, pixel, depth
__device__ __forceinline__ bool project_neuray(const float* P, float X, float Y, float Z, int Wimg, int H, float& px, float& py, float& depth) {
  const float cx = fmaf(P[2], Z, fmaf(P[1], Y, P[0] * X)) + P[3];
  const float cy = fmaf(P[6], Z, fmaf(P[5], Y, P[4] * X)) + P[7];
  depth = fmaf(P[10], Z, fmaf(P[9], Y, P[8] * X)) + P[11];
  const bool bad = fabsf(depth) < 1e-4f;
  if (bad) depth = 1e-3f;
  px = cx / depth; py = cy / depth;
  const bool outside = (px < -0.5f) | (px >= (float)Wimg - 0.5f) | (py < -0.5f) | (py >= (float)H - 0.5f);
  return !bad && !outside;
}

// bilinear (border, align_corners=False) tap of the channels-last 32-channel visibility map; zeros when !valid
__device__ __forceinline__ void sample_visf(const float* __restrict__ base, int h, int w, int Wimg, int H, float px, float py, bool valid, float (&x)[32]) {
  if (valid) {
    const float xn = px / (float)(Wimg - 1) * 2.f - 1.f;
    const float yn = py / (float)(H - 1) * 2.f - 1.f;
    const Taps t = make_taps<false, true>(xn, yn, w, h);
    const int xe = t.x0 + 1, ys = t.y0 + 1;
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      float4 a = make_float4(0, 0, 0, 0), b = a, c = a, d = a;
      if (t.mn && t.mw) a = *(const float4*)(base + ((size_t)t.y0 * w + t.x0) * 32 + c4 * 4);
      if (t.mn && t.me) b = *(const float4*)(base + ((size_t)t.y0 * w + xe) * 32 + c4 * 4);
      if (t.ms && t.mw) c = *(const float4*)(base + ((size_t)ys * w + t.x0) * 32 + c4 * 4);
      if (t.ms && t.me) d = *(const float4*)(base + ((size_t)ys * w + xe) * 32 + c4 * 4);
      x[c4 * 4 + 0] = a.x * t.nw + b.x * t.ne + c.x * t.sw + d.x * t.se;
      x[c4 * 4 + 1] = a.y * t.nw + b.y * t.ne + c.y * t.sw + d.y * t.se;
      x[c4 * 4 + 2] = a.z * t.nw + b.z * t.ne + c.z * t.sw + d.z * t.se;
      x[c4 * 4 + 3] = a.w * t.nw + b.w * t.ne + c.w * t.sw + d.w * t.se;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = 0.f;
  }
}

// the four decoders -> mean(2), var(2)+0.05, vis, aw
__device__ __forceinline__ void decode_all(const float* __restrict__ dw, const float (&x)[32], float& m0, float& m1, float& v0, float& v1, float& vs, float& aw) {
  float dummy;
  decoder(dw + 0 * DEC_STRIDE, x, m0, m1);
  decoder(dw + 1 * DEC_STRIDE, x, v0, v1);
  decoder(dw + 2 * DEC_STRIDE, x, aw, dummy);
  decoder(dw + 3 * DEC_STRIDE, x, vs, dummy);
  m0 = nl_softplus(m0); m1 = nl_softplus(m1);
  v0 = nl_softplus(v0) + 0.05f; v1 = nl_softplus(v1) + 0.05f;
  aw = nl_sigmoid(aw); vs = nl_sigmoid(vs);
}

}  // namespace nlmv
