// Shared host/device definitions for libnerfloc_render.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nerfloc_render.h"

#define NL_WAVE 64
#define NL_F (195)        // C+3 for C=192; kernels take F at run time, this is the padded-row default
#define NL_FPAD (196)

#define NL_CHECK_HIP(expr)                      \
  do {                                          \
    hipError_t _e = (expr);                     \
    if (_e != hipSuccess) return NL_ERR_HIP;    \
  } while (0)

#define NL_LAUNCH_CHECK()                                   \
  do {                                                      \
    if (hipPeekAtLastError() != hipSuccess) return NL_ERR_HIP; \
  } while (0)

static inline size_t nl_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// Workgroups of a persistent kernel on the CURRENT device: its CU count rounded down to a multiple of the 8 XCDs (>= 8); cached per device id
// (abi.hip) — a process may drive several GPUs through several HipRenderers.  < 0: NL_ERR_HIP
int nl_persistent_cus();
static inline int64_t nl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ activations (match torch fp32 CPU ops)
enum { NL_ACT_NONE = 0, NL_ACT_LRELU = 1, NL_ACT_ELU = 2,
       NL_ACT_LRELU_MASK = 3 };   // backward passes (streaming GEMM only): out = acc * LeakyReLU'(.) taken from the sign of ep_res[m][n] (the layer's forward output)

__device__ __forceinline__ float nl_lrelu(float x) { return x > 0.f ? x : x * 0.01f; }
__device__ __forceinline__ float nl_elu(float x) { return x > 0.f ? x : expm1f(x); }
// ELU with the negative branch on the hardware exp2 unit (abs error <= ~2e-7 on a quantity in (-1, 0]); used by the MFMA
// decoder kernel where 192 ELUs per row would otherwise dominate
// (written as a median: e^x - 1 >= x everywhere, so for x > 0 the order is 0 < x <= e^x - 1 and for x < 0 it is x <= e^x - 1 < 0 — the middle value is ELU(x)
// in both cases: ONE instruction, v_med3_f32, instead of compare + select.  Where the rounded e^x - 1 lands a rounding error below x (|x| < ~1e-3) the median
// returns the other of two values that differ by <= 1.2e-7 — the absolute error the e^x - 1 branch has there anyway; round 5)
// NaN: v_med3_f32 with a NaN operand falls back to min3 — ELU(NaN) comes out as 0, not NaN (the compare + select it replaced handed the NaN on).  Its callers are the
// NeuRay decoders (mvdec.h), whose inputs are bilinear taps of the per-frame CNN's maps; a NaN there is a NaN of the frame's inputs, which the decoders' first-layer
// products still carry into the hidden units of the OTHER rows of the tap (the outputs stay non-finite in practice: tests feed finite maps only), but this function
// by itself does not propagate it — use nl_elu where that matters.
__device__ __forceinline__ float nl_elu_fast(float x) { return __builtin_amdgcn_fmed3f(x, __expf(x) - 1.f, 0.f); }
// max(v, the value of lane ^ 1) as ONE DPP-modified v_max_f32 (MaxPool over neighbouring positions held in neighbouring lanes).  `fmaxf(v, nl_dpp<0xB1>(v, v))` costs four:
// the DPP move plus a canonicalising v_max_f32 x, x per operand in front of the maximum (llvm.maxnum under IEEE mode).  The hardware instruction quiets NaNs by itself; for
// everything else the value is the same.  (s_nop 1: the two wait states a DPP read needs behind a VALU write of its source — the compiler does not see into the statement.)
__device__ __forceinline__ float nl_max_lane_xor1(float v) {
  float r;
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
  return r;
}
// Split-bf16 of a PAIR: hi = the pair rounded to bf16 (one v_cvt_pk_bf16_f32), lo = bf16(v - float(hi)) with the subtraction as one packed instruction — per element the
// same two roundings as `h = (__bf16)v; l = (__bf16)(v - (float)h)`, which hipcc lowers to a conversion per ELEMENT for the subtraction and a second, packed one for the
// store (32 vector instructions per 8 values; this form: 20).  Results as raw 32-bit words: element 0 in the low half.
typedef float nl_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 nl_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned nl_bf16_pair(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(nl_f32x2{a, b}, nl_bf16x2)); }
__device__ __forceinline__ void nl_split_bf16_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const nl_f32x2 v = {a, b};
  const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, nl_bf16x2));
  const nl_f32x2 f = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  hi = u;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - f, nl_bf16x2));
}
// Element-wise epilogue arithmetic on PAIRS (v_pk_add_f32 / v_pk_mul_f32; round 6): per element the same operations in the same order as the scalar forms they replace.
// ELU of a pair (nl_elu_fast per element): 6 vector instructions per pair instead of 8
__device__ __forceinline__ nl_f32x2 nl_elu_fast2(nl_f32x2 x) {
  const nl_f32x2 y = x * nl_f32x2{1.4426950408889634f, 1.4426950408889634f};
  nl_f32x2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
  e = e - nl_f32x2{1.f, 1.f};
  return nl_f32x2{__builtin_amdgcn_fmed3f(x[0], e[0], 0.f), __builtin_amdgcn_fmed3f(x[1], e[1], 0.f)};
}
// ELU((x - mean) * rstd * g + b) of four values
__device__ __forceinline__ float4 nl_ln_elu4(float x0, float x1, float x2, float x3, float mean, float rstd, const float4& g, const float4& b) {
  const nl_f32x2 mm = {mean, mean}, rr = {rstd, rstd};
  const nl_f32x2 p0 = nl_elu_fast2((nl_f32x2{x0, x1} - mm) * rr * nl_f32x2{g.x, g.y} + nl_f32x2{b.x, b.y});
  const nl_f32x2 p1 = nl_elu_fast2((nl_f32x2{x2, x3} - mm) * rr * nl_f32x2{g.z, g.w} + nl_f32x2{b.z, b.w});
  return make_float4(p0[0], p0[1], p1[0], p1[1]);
}
// LeakyReLU(0.01) of a pair: max(x, 0.01 x) — the value of `x > 0 ? x : 0.01 x` for every input (both orderings agree on +-0 and NaN); one packed multiply + two maxima
__device__ __forceinline__ nl_f32x2 nl_lrelu2(nl_f32x2 x) {
  const nl_f32x2 z = x * nl_f32x2{0.01f, 0.01f};
  return nl_f32x2{fmaxf(x[0], z[0]), fmaxf(x[1], z[1])};
}
// (x0..x3) + b as two pairs; their sum in `sum` (two packed adds + the horizontal ones)
__device__ __forceinline__ void nl_bias_sum4(float x0, float x1, float x2, float x3, const float4& b, nl_f32x2& p0, nl_f32x2& p1, float& sum) {
  p0 = nl_f32x2{x0, x1} + nl_f32x2{b.x, b.y};
  p1 = nl_f32x2{x2, x3} + nl_f32x2{b.z, b.w};
  const nl_f32x2 q = p0 + p1;
  sum += q[0] + q[1];
}
// sp += (x - mean)^2 of a pair (two running sums; the caller adds the halves at the end)
__device__ __forceinline__ void nl_sumsq_dev2(nl_f32x2& sp, float x0, float x1, float mean) {
  const nl_f32x2 d = nl_f32x2{x0, x1} - nl_f32x2{mean, mean};
  sp += d * d;
}
__device__ __forceinline__ float nl_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float nl_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float nl_act(float x, int act) {
  return act == NL_ACT_LRELU ? nl_lrelu(x) : (act == NL_ACT_ELU ? nl_elu(x) : x);
}

// ------------------------------------------------------------------ wave helpers (wave64)
// Cross-lane reductions are DPP-modified VALU instructions (one v_add_f32 / v_max_f32 each), not __shfl: a shuffle is a
// ds_bpermute round trip through the LDS crossbar plus address arithmetic, ~10x the cost.
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float nl_dpp(float v, float old = 0.f) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
// sum / max over aligned groups of 8 lanes, result in all 8: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
__device__ __forceinline__ float nl_sum8(float v) {
  v += nl_dpp<0xB1>(v); v += nl_dpp<0x4E>(v); v += nl_dpp<0x141>(v);
  return v;
}
__device__ __forceinline__ float nl_max8(float v) {
  v = fmaxf(v, nl_dpp<0xB1>(v, v)); v = fmaxf(v, nl_dpp<0x4E>(v, v)); v = fmaxf(v, nl_dpp<0x141>(v, v));
  return v;
}
// whole-wave sum / max, result wave-uniform (row_shr 1/2/4/8 scan inside each row of 16, row_bcast:15 / :31 carry the row
// totals up, lane 63 holds the total)
__device__ __forceinline__ float wave_sum(float v) {
  v += nl_dpp<0x111>(v); v += nl_dpp<0x112>(v); v += nl_dpp<0x114>(v); v += nl_dpp<0x118>(v);
  v += nl_dpp<0x142, 0xa>(v); v += nl_dpp<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -3.4028235e38f;
  v = fmaxf(v, nl_dpp<0x111>(v, ninf)); v = fmaxf(v, nl_dpp<0x112>(v, ninf)); v = fmaxf(v, nl_dpp<0x114>(v, ninf)); v = fmaxf(v, nl_dpp<0x118>(v, ninf));
  v = fmaxf(v, nl_dpp<0x142, 0xa>(v, ninf)); v = fmaxf(v, nl_dpp<0x143, 0xc>(v, ninf));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ------------------------------------------------------------------ XCD-aware block order
// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  nl_xcd_block() turns the hardware block id into a
// logical one such that every XCD works through ONE contiguous eighth of the index space: neighbouring samples of a ray
// (which tap neighbouring texels / share KNN leaves and table rows) then meet in the same L2.  Launch nl_xcd_grid(G)
// blocks (a multiple of 8); logical ids >= G exit through the kernel's ordinary bounds check.
__device__ __forceinline__ unsigned nl_xcd_block() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }
static inline unsigned nl_xcd_grid(int64_t G) { return (unsigned)((G + 7) / 8 * 8); }

// ------------------------------------------------------------------ camera block passed by value (kernarg -> SGPRs)
struct NlViews {
  float P1[NL_MAX_VIEWS][12];   // Projector rows 0..2 (ibrnet.py:183)
  float P2[NL_MAX_VIEWS][12];   // NeuRay K@Rt (depth_fusion.py:90)
  float cam[NL_MAX_VIEWS][3];   // support camera centres
  float qcam[3];                // query camera centre
  const float* qrows;           // optional per-ray query centres (R, 3): sample n belongs to ray n / qS (several query frames per launch)
  int qS;
  int V, H, Wimg, h, w;   // h,w: feature map; vh,vw: visibility map
  int vh, vw;
  float near_, far_;
};

// ------------------------------------------------------------------ KNN grid (device-resident parameters)
struct NlGridParams {
  float origin[3];
  float inv_cell;
  float cell;
  int dims[3];
  float bmax[3];
  int ncells;
};

// ------------------------------------------------------------------ fused neural-point kernels (point_fused.hip, point_fused2.hip)
struct NlPointFusedArgs {
  const float* xyz; const float* dir; int dir_stride, dir_div;
  const int* idx;          // (N,8) neighbour indices
  const float* Q;          // (N,128) query projection (w_qs . mv_feat)
  float* O;                // (N,128) attention output
  const float* ptt;        // [(M+1)][W] per-frame table T = sp_feature . W1[:, :F]^T + b1 in accumulator order; row M = b1 only
  const float* sp_xyz; const float* sp_dir;
  const uint4* wstream;    // packed weight stream (v1: pack_point_stream_kernel)
  const float* bias;       // [3][W] base_mlp biases
  const float* rd_w;       // ray_diff_fc: W0[16][4], b0[16], W2[27][16], b2[27]
  int N, M;
  float inv_span;
  const uint4* wstream2;   // v2 stream (pack_point_stream2_kernel): row-tile-major chunks + resident block
  unsigned* logit_amax = nullptr;   // optional (v1 kernel; the v2 launcher takes it as a parameter): running max |attention logit| as float bits (nl_frame_diagnostics)
};

// ------------------------------------------------------------------ the five inner ray-U-Net layers as one kernel (unet_inner.hip)
struct NlUnetInnerArgs {
  const float* c1;            // (R, 64, 64) fp32: conv1's pooled block output
  float* c2;                  // (R, 32, 128) fp32: conv2's pooled block output — written (it is also parked here while its LDS region holds x0) and read back
  float* x2;                  // (R, 128, 32) fp32: trans_conv1's block output (conv_out's second source)
  const char* w[5];           // weight streams in tgemm_kernel's chunk layout: conv2, conv3, trans_conv3 / 2 / 1 (merged phases)
  const float* bias[5];
  const float* gl[5];         // LayerNorm tables in accumulator-lane order (abi.hip: ln_lane_major_kernel)
  const float* bl[5];
  int R;
  float eps;
};

// ------------------------------------------------------------------ generic segment GEMM (gemm.hip)
#define NL_GEMM_MAX_SEG 6
struct NlGemmSeg {
  const float* ptr;  // source rows
  int ld;            // row stride (floats)
  int k;             // columns taken from this source
  int ioff;          // conv tap offset along the ray (row-mapped modes)
  int rdiv;          // plain mode: source row = m / rdiv (row broadcast), >=1
  int vec;           // 1: rows are 16-B aligned (ptr, ld) so 4 consecutive k may be fetched with one float4 load
  int ntap;          // 1, or T > 1 (k % 32 == 0): conv taps interleaved per 32-channel block — the segment spans T*k slots of
                     // K-space ordered [block k/32][tap 0..T-1][32]; tap t reads row offset ioff + t - T/2.  Neighbouring chunks then
                     // re-read the same rows (+-1), which L1 still holds; tap-major order streams the source T times through L2
  int frag;          // 1 (tgemm.hip only; k % 32 == 0): `ptr` is a FRAGMENT-NATIVE split-bf16 image instead of fp32 rows — [32-row tile][k-step of 16][hi | lo][lane][8 bf16],
                     // lane = (row & 31) + 32 hh holding the k-slots 8 hh .. 8 hh + 7 of the k-step, i.e. exactly the B fragments the consumer's MFMAs take: what
                     // sample_chain_kernel<.., FRAGOUT> writes for feature_agg (channels of every 32-block in ACCUMULATOR order: the layer's weights are packed to
                     // match, G_CONV1F / G_CONVOUTF).  The consumer loads 16-byte pieces that are contiguous across the wave and converts nothing (round 5).
                     // 2: fp32 rows as ever, but the layer's K order inside every 32-block is that accumulator order (the same packed weights serve both source
                     // formats, so a batch renders to the same bits whichever format its chunk took)
                     // 3 (late round 6; tgemm_conv1_kernel / tgemm_mx_kernel / feat_comp_mx_kernel only): the fragment image of 1 in split-FP16 — hi = f16(v), lo = f16(v - hi)
                     // (sample_chain_kernel<.., F16FRAG>): the same bytes in the same places, taken as fp16 operands
};
struct NlGemmArgs {
  NlGemmSeg seg[NL_GEMM_MAX_SEG];
  int nseg;
  int M, K, N;          // output rows (compute index space), total PADDED K (each segment rounded up to 4), output columns
  int Kpad, Npad;       // padded sizes of B
  const void* B;        // packed weights: f32 [Kpad][Npad]  or bf16 hi/lo [Npad][Kpad] (see pack.hip)
  const void* Blo;      // bf16x3 only
  const void* Bst;      // bf16 hi/lo weight stream in A-fragment chunk order (tgemm.hip), or null
  const void* Bsh16;    // the layer's fp16 hi / lo stream for a launch whose input is the split-FP16 fragment image (NlGemmSeg::frag == 3: tgemm_conv1_kernel<true, true>), or null
  const void* Bsh_mx;   // NL_PREC_F16MX (round 6, tgemm_mx_kernel): the layer's fp16 hi/lo stream (its hi fragments are read) ...
  const void* Bmx;      // ... and its fp6 images + block scales (pack_tgemm_mx6_kernel), or null
  const float* zeros;   // >= 512 zero floats (source row of conv halos / rows beyond M in tgemm.hip)
  int kstart[NL_GEMM_MAX_SEG];   // first k of each segment in the padded K space; INT_MAX for unused slots
  const float* bias;    // [N] or null
  float* C;
  int ldc;
  int act;
  // row mapping: So==0 plain (out row = m); else r=m/So, t=m%So, in row = r*Li + t + ioff (valid 0<=t+ioff<Li),
  // out row = r*Lo + t*ostride + ooff
  int So, Li, Lo, ostride, ooff;
  // optional fused epilogues (tgemm.hip only)
  //   NL_EPI_LNROW : out = (LayerNorm_row(acc + res[m]; eps) * gamma[n] + beta[n]) * scale[m]
  //   NL_EPI_LNSLAB: out = ELU(LayerNorm over the whole (So x N) slab of one ray * gamma[t][n] + beta[t][n]); ep_pool: MaxPool(2) along the ray
  int epi; int ep_pool;
  const float* ep_res; int ep_ldres;
  const float* ep_gamma; const float* ep_beta; const float* ep_scale;
  float ep_eps;
  // NL_EPI_LNSLAB only, optional: sigma[m] = softplus(ep_sig_w . out_row + ep_sig_b[0]) (model.py:525) from the values in registers
  const float* ep_sig_w; const float* ep_sig_b; float* ep_sig_out;
  // optional (tgemm.hip, plain row mapping, no fused epilogue): only the 32-row tiles listed in tile_map[0 .. *tile_count) are computed
  // (early termination: rows of dead samples are neither read nor written)
  const int* tile_map; const int* tile_count;
  // optional (tgemm.hip, plain epilogue): the SIGN of every output (post-activation) as bits, 16 bytes per lane of each 32-row tile ([tile][lane][4 dwords],
  // bit 16 (rt & 1) + 4 gq + e of dword rt >> 1 <-> column 32 rt + 8 gq + 4 hh + e) — written by a forward layer (ep_maskout), read back by the
  // NL_ACT_LRELU_MASK product of the backward pass (ep_maskin) instead of 1 KB of activations per row
  unsigned* ep_maskout; const unsigned* ep_maskin;
  // plain epilogue, optional: + ep_tab[row_of(m)][slot(n)] (columns in accumulator order, as the fused kernel's table) before the activation, row_of(m) = ep_tabidx[m] if (m % ep_tabK) < ep_tabM else ep_tabM — the neural-point
  // branch's first layer on the per-frame table (support features x their weight columns + bias; row ep_tabM = bias only: zero-filled neighbours)
  const float* ep_tab; const int* ep_tabidx; int ep_ldtab, ep_tabK, ep_tabM;
};
enum { NL_EPI_NONE = 0, NL_EPI_LNROW = 1, NL_EPI_LNSLAB = 2 };
// internal arithmetic of the segment GEMMs beyond the public nl_precision values: three-term split-FP16 (tgemm.hip), used by the backward passes for
// their recomputed forward; falls back to exact fp32 where the streaming kernel does not apply
#define NL_PREC_F16X3_INTERNAL 16

int nl_gemm_launch(const NlGemmArgs& a, int precision, hipStream_t stream);
// streaming transposed GEMM (tgemm.hip): bf16 modes, N <= 256, 16-B aligned segments
int nl_tgemm_nrt(int N);
size_t nl_tgemm_stream_bytes(int Kpad, int N);
bool nl_tgemm_supported(const NlGemmArgs& a, int precision);
int nl_tgemm_launch(const NlGemmArgs& a, int precision, hipStream_t stream);
// feat_mlp.0 + LeakyReLU + the compositing of its rows along the ray in the f16mx arithmetic (tgemm.hip: feat_comp_mx_kernel): hc (N / S, 256) from feature_agg's
// fragment image, the samples' compositing weights, G_FEAT0P's fp16 stream and its fp6 images
bool nl_feat_comp_mx_supported(int W, int S, int64_t N);
// w2 != null: feat_mlp.2 too (G_FEAT2's packed fp32 matrix [k][npad], row 256 = the bias that meets the weight sum): feat (N / S, C) is written, hc is not
int nl_launch_feat_comp_mx(const float* fa_frag, const float* wts, int64_t N, int S, const void* bsh, const void* bmx, const float* bias, float* hc, hipStream_t st,
                           const float* w2 = nullptr, int npad = 0, int C = 0, const float* wsum = nullptr, float* feat = nullptr, bool frag_f16 = false);
