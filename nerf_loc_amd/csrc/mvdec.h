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

// The four tap offsets of one bilinear sample in one register: offset of the (clamped) north-west texel in bits 0-29, whether the
// east column / south row is a different texel in bits 30 / 31.  Taps outside the map are clamped onto a valid texel; their
// weights are zero.  unpack_taps runs on the scalar unit (the packed word comes from v_readlane).
__device__ __forceinline__ unsigned pack_taps(const Taps& t, int w, int h) {
  const int x0 = min(max(t.x0, 0), w - 1), x1 = min(max(t.x0 + 1, 0), w - 1);
  const int y0 = min(max(t.y0, 0), h - 1), y1 = min(max(t.y0 + 1, 0), h - 1);
  return (unsigned)(y0 * w + x0) | ((unsigned)(x1 - x0) << 30) | ((unsigned)(y1 - y0) << 31);
}
__device__ __forceinline__ void unpack_taps(unsigned pk, int w, int (&o)[4]) {
  const int base = (int)(pk & 0x3fffffffu), dx = (int)((pk >> 30) & 1u), dy = (pk >> 31) ? w : 0;
  o[0] = base; o[1] = base + dx; o[2] = base + dy; o[3] = base + dy + dx;
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

// ---------------------------------------------------------------------------------------------------------------
// MFMA evaluation of the four decoders for a tile of 32 rows (row = one (view, point) pair): used by mv_vis_mfma_kernel (mvagg.hip)
// and coarse_view_mfma_kernel (hier.hip).  Transposed MFMA: weights = A operand (LDS, fragment order), the rows' activations = B
// operand in registers; lane (j = lane & 31, hh = lane >> 5) holds channels {8hh..8hh+7} and {16+8hh..16+8hh+7} of row j.  The four
// decoders' first layers are 32 -> 32 products (2 k-steps), the second layers 32 -> 32 with K in the accumulator's register order, so
// the hidden activations go C/D registers -> ELU -> 16-bit split -> B fragments without leaving the lane; the 6 output units are VALU
// dots over the lane's 16 hidden values + one cross-half add.
// Arithmetic: X3 = THREE-TERM SPLIT-FP16 (hi.hi + lo.hi + hi.lo, fp32 accumulate): fp16 carries 11 significant bits per part, so the
// products are good to ~2^-22 — the decoders feed exp / tanh / a division by the summed visibility and are the one place of the
// path where split-bf16's 2^-17 showed (a sample whose views are all almost invisible turned 7e-6 of visibility error into 2e-4 of
// compositing weight, tools/precision_budget.py); the visibility features are O(1) outputs of the per-frame CNN, far inside fp16's
// range.  !X3 = one bf16 MFMA per product (throughput mode).
typedef __bf16 mvd_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 mvd_f16x8 __attribute__((ext_vector_type(8)));
typedef float mvd_f32x16 __attribute__((ext_vector_type(16)));

// dpack (uint4 units): [0, 2048) bf16 fragments: W1 hi [q 2][d 4][lane 64] at 0, lo at +512; W2 hi [d 4][s 2][lane 64] at 1024, lo at
// +512.  [2048, 2048 + 130) floats: b1[128] b2[128] w4p[4][2][2][16] b4[8].  [2178, 2178 + 2048) the same fragments in fp16 hi / lo.
constexpr int MVD_W1 = 0;
constexpr int MVD_W2 = 1024;
constexpr int MVD_F32 = 2048;
constexpr int MVD_NF32 = (128 + 128 + 256 + 8) / 4;
constexpr int MVD_F16 = MVD_F32 + MVD_NF32;
// [MVD_T, MVD_T + 2048): TRANSPOSED layer-2 / layer-1 fragments in bf16 hi / lo for the decoders' input gradient (backward.hip):
//   W2T [part 2][d 4][s 2][lane 64] at +0 (1024 uint4): A row = hidden-1 unit, K = hidden-2 units in accumulator order
//   W1T [part 2][d 4][s 2][lane 64] at +1024: A row i = the INPUT CHANNEL that the tap code keeps in register r of half hh with m(r, hh) = i
//       (channels 8 hh + r for r < 8, 16 + 8 hh + r - 8 otherwise), K = hidden-1 units in accumulator order
constexpr int MVD_T = MVD_F16 + 2048;
constexpr int MVD_PACK_UINT4 = MVD_T + 2048;   // global image
constexpr int MVD_LDS_UINT4 = 2048 + MVD_NF32;   // what a kernel keeps in LDS: ONE set of fragments + the floats

// copies the fragments the kernel's mode needs + the float block into LDS (all threads of the block; caller syncs)
template <bool X3>
__device__ __forceinline__ void mvd_load_lds(uint4* sw, const uint4* __restrict__ dpack, int tid, int nthreads) {
  for (int i = tid; i < 2048; i += nthreads) sw[i] = dpack[(X3 ? MVD_F16 : 0) + i];
  for (int i = tid; i < MVD_NF32; i += nthreads) sw[2048 + i] = dpack[MVD_F32 + i];
}

template <bool X3> struct MvdOps;
typedef unsigned mvd_u32x4 __attribute__((ext_vector_type(4)));
template <> struct MvdOps<true> {
  typedef mvd_f16x8 v8;
  // hi = f16(v), lo = f16(v - hi): per pair v_cvt_pk_f16_f32 | 2 x v_fma_mix_f32 (reads the f16 half directly: no v_cvt_f32_f16, no separate subtraction) |
  // v_cvt_pk_f16_f32 = 2 vector instructions per value instead of 3.5 (round 4: the kernel is bound by its vector instruction count, profiles/r4_pmc_sq.csv)
  static __device__ __forceinline__ void split(const float (&v)[8], v8& hi, v8& lo) {
    mvd_u32x4 h, l;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned hp, lp; float l0, l1;
      asm("v_cvt_pk_f16_f32 %0, %3, %4\n\tv_fma_mix_f32 %1, %0, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
          : "=&v"(hp), "=&v"(l0), "=&v"(l1) : "v"(v[2 * t]), "v"(v[2 * t + 1]));
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(l0), "v"(l1));
      h[t] = hp; l[t] = lp;
    }
    hi = __builtin_bit_cast(v8, h); lo = __builtin_bit_cast(v8, l);
  }
  static __device__ __forceinline__ mvd_f32x16 mfma(v8 a, v8 b, mvd_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct MvdOps<false> {
  typedef mvd_bf16x8 v8;
  static __device__ __forceinline__ void split(const float (&v)[8], v8& hi, v8& lo) {
#pragma unroll
    for (int t = 0; t < 8; ++t) hi[t] = (__bf16)v[t];
    lo = hi;
  }
  static __device__ __forceinline__ mvd_f32x16 mfma(v8 a, v8 b, mvd_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ void mvd_split_bf16(const float (&v)[8], mvd_bf16x8& hi, mvd_bf16x8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) { const __bf16 h = (__bf16)v[t]; hi[t] = h; lo[t] = (__bf16)(v[t] - (float)h); }
}

// ELU of two values at once (x > 0 ? x : exp(x) - 1 with the exponential on the hardware exp2 unit, as nl_elu_fast): the scale by log2(e) and the -1 are ONE packed
// instruction each for the pair (v_pk_mul_f32, v_pk_add_f32), the select is a median — 6 vector instructions per pair (round 4: 8; scalar form: 10)
typedef float mvd_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mvd_elu2(float& a, float& b) {
  const mvd_f32x2 x = {a, b};
  const mvd_f32x2 y = x * mvd_f32x2{1.4426950408889634f, 1.4426950408889634f};
  mvd_f32x2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
  e = e - mvd_f32x2{1.f, 1.f};
  a = __builtin_amdgcn_fmed3f(x[0], e[0], 0.f);   // = x > 0 ? x : e up to the rounding of e near 0 (e^x - 1 >= x: the median of {x, e, 0} is ELU(x); common.h: nl_elu_fast)
  b = __builtin_amdgcn_fmed3f(x[1], e[1], 0.f);
}

// this lane's 16 channels of the bilinear (border, align_corners = False) tap of the channels-last 32-channel visibility map at
// pixel (px, py) of view `base`; zero when !valid (depth_fusion.py:60-76, neuray_ops.py:14-36)
__device__ __forceinline__ void mvd_tap16(const float* __restrict__ base /* view's map + 8 * hh */, int vh, int vw_, int Wimg, int H, float px, float py,
                                          bool valid, float (&x0)[8], float (&x1)[8]) {
  const float xn = px / (float)(Wimg - 1) * 2.f - 1.f;
  const float yn = py / (float)(H - 1) * 2.f - 1.f;
  const Taps t = make_taps<false, true>(xn, yn, vw_, vh);
  const size_t o00 = ((size_t)(t.mn ? t.y0 : 0) * vw_ + (t.mw ? t.x0 : 0)) * 32;
  const size_t o01 = ((size_t)(t.mn ? t.y0 : 0) * vw_ + (t.me ? t.x0 + 1 : 0)) * 32;
  const size_t o10 = ((size_t)(t.ms ? t.y0 + 1 : 0) * vw_ + (t.mw ? t.x0 : 0)) * 32;
  const size_t o11 = ((size_t)(t.ms ? t.y0 + 1 : 0) * vw_ + (t.me ? t.x0 + 1 : 0)) * 32;
  const float w00 = (valid && t.mn && t.mw) ? t.nw : 0.f, w01 = (valid && t.mn && t.me) ? t.ne : 0.f;
  const float w10 = (valid && t.ms && t.mw) ? t.sw : 0.f, w11 = (valid && t.ms && t.me) ? t.se : 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int c4 = 0; c4 < 2; ++c4) {
      const int co = 16 * g + 4 * c4;
      const float4 a = *(const float4*)(base + o00 + co), b = *(const float4*)(base + o01 + co);
      const float4 c = *(const float4*)(base + o10 + co), d = *(const float4*)(base + o11 + co);
      float* dst = g ? x1 : x0;
      // (explicit fma chains: the file is built with -ffp-contract=off, which costs 7 instructions per tap sum instead of 4)
      dst[4 * c4 + 0] = fmaf(d.x, w11, fmaf(c.x, w10, fmaf(b.x, w01, a.x * w00)));
      dst[4 * c4 + 1] = fmaf(d.y, w11, fmaf(c.y, w10, fmaf(b.y, w01, a.y * w00)));
      dst[4 * c4 + 2] = fmaf(d.z, w11, fmaf(c.z, w10, fmaf(b.z, w01, a.z * w00)));
      dst[4 * c4 + 3] = fmaf(d.w, w11, fmaf(c.w, w10, fmaf(b.w, w01, a.w * w00)));
    }
  }
}

// the four decoders of a 32-row tile, one after the other (only one decoder's 2 x 16 accumulators are live at a time: that is what
// lets four waves share a SIMD) -> mean(2), var(2) + 0.05, vis, aw of this lane's row (both halves of the wave hold the result)
template <bool X3>
__device__ __forceinline__ void mvd_decode_tile(const uint4* sw, int lane, const float (&x0)[8], const float (&x1)[8], float& m0, float& m1,
                                                float& v0, float& v1, float& vs, float& aw) {
  typedef MvdOps<X3> OP;
  typedef typename OP::v8 v8;
  const int hh = lane >> 5;
  const float* sf = reinterpret_cast<const float*>(sw + 2048);
  const float* b1 = sf, *b2 = sf + 128, *w4p = sf + 256, *b4 = sf + 512;
  v8 xh[2], xl[2];
  OP::split(x0, xh[0], xl[0]);
  OP::split(x1, xh[1], xl[1]);
  float o[4][2];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    mvd_f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *(const float4*)(b1 + 32 * d + 8 * g + 4 * hh);
      acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const v8 ah = __builtin_bit_cast(v8, sw[MVD_W1 + (q * 4 + d) * 64 + lane]);
      if (X3) {
        const v8 al = __builtin_bit_cast(v8, sw[MVD_W1 + 512 + (q * 4 + d) * 64 + lane]);
        acc = OP::mfma(al, xh[q], acc);
        acc = OP::mfma(ah, xl[q], acc);
      }
      acc = OP::mfma(ah, xh[q], acc);
    }
    v8 gh[2], gl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float vv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) vv[t] = acc[8 * s + t];
#pragma unroll
      for (int t = 0; t < 8; t += 2) mvd_elu2(vv[t], vv[t + 1]);
      OP::split(vv, gh[s], gl[s]);
    }
    mvd_f32x16 acc2;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *(const float4*)(b2 + 32 * d + 8 * g + 4 * hh);
      acc2[4 * g] = b.x; acc2[4 * g + 1] = b.y; acc2[4 * g + 2] = b.z; acc2[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const v8 ah = __builtin_bit_cast(v8, sw[MVD_W2 + (d * 2 + s) * 64 + lane]);
      if (X3) {
        const v8 al = __builtin_bit_cast(v8, sw[MVD_W2 + 512 + (d * 2 + s) * 64 + lane]);
        acc2 = OP::mfma(al, gh[s], acc2);
        acc2 = OP::mfma(ah, gl[s], acc2);
      }
      acc2 = OP::mfma(ah, gh[s], acc2);
    }
    float h2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) h2[r] = acc2[r];
#pragma unroll
    for (int r = 0; r < 16; r += 2) mvd_elu2(h2[r], h2[r + 1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float p = 0.f;
      if (u == 0 || d < 2) {
        const float* w = w4p + ((d * 2 + u) * 2 + hh) * 16;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 ww = *(const float4*)(w + 4 * r4);
          p = fmaf(ww.x, h2[4 * r4], p); p = fmaf(ww.y, h2[4 * r4 + 1], p);
          p = fmaf(ww.z, h2[4 * r4 + 2], p); p = fmaf(ww.w, h2[4 * r4 + 3], p);
        }
        p += __shfl_xor(p, 32, 64);
        p += b4[d * 2 + u];
      }
      o[d][u] = p;
    }
  }
  m0 = nl_softplus(o[0][0]); m1 = nl_softplus(o[0][1]);
  v0 = nl_softplus(o[1][0]) + 0.05f; v1 = nl_softplus(o[1][1]) + 0.05f;
  aw = nl_sigmoid(o[2][0]); vs = nl_sigmoid(o[3][0]);
}

}  // namespace nlmv
