// Fused neural-point branch, second generation — the MFMA roofline kernel.
//
//   per (sample, neighbour) row:  [feature(195) | posenc(63) | ray_diff_fc(27)]            model.py:394-409
//       -> base_mlp 285->W->W->W (LeakyReLU)                                                model.py:63-71
//       -> k/v projections W->128+128                                                       ibrnet.py:98-99
//   per sample: 4-head attention of the (precomputed) query over its 8 neighbours -> O (N,128)  ibrnet.py:28-45,104
//
// What changed against point_fused.hip (kept for W = 64) and why — measured with tools/ubench/mfma_fill.hip / mfma_rowtile.hip:
// one wave per SIMD hides <= 5 plain VALU instructions (or 2-3 v_cvt_pk_bf16_f32, 8 cycles each, or 2 ds_read_b128, 16 cycles
// of LDS bandwidth each) in the 32-cycle issue shadow of every v_mfma_f32_32x32x16_bf16, but ONLY if they sit between the MFMAs
// in program order.  The first kernel ran K-outer / row-tile-inner, so all 128 accumulators of a layer finished together and
// the LeakyReLU + bf16 hi/lo split of the whole layer (and the operand assembly, the LDS-DMA bursts, the attention) ran with the
// matrix pipe idle: 38 % MFMA-busy.  Here every layer is OUTPUT-STATIONARY:
//   * a chunk of the weight stream is one 32-row output tile x all K (32 KB in bf16x3) instead of 2 k-steps x all row tiles;
//     the wave accumulates ONE tile (16 registers) over the whole K of the layer;
//   * while tile rt accumulates, the epilogue of tile rt-1 (+bias / +table row, LeakyReLU, hi/lo split -> B fragments of the next
//     layer), the LDS-DMA pieces of the chunk three ahead and the A-fragment reads two k-steps ahead are issued a few
//     instructions at a time after each MFMA (`fill`), fenced with sched_barrier so the compiler keeps that order;
//   * the k / v projections finish one attention head (= one row tile) at a time, so the softmax over the 8 neighbours and the
//     weighted sums run in the shadow of the next head's MFMAs; 128 accumulators are never live;
//   * activations of layer L (128 VGPRs as hi/lo bf16 fragments) and of layer L+1 (128, being produced) are both register
//     resident: ~2 x 128 + 2 x 16 accumulators + fragments, one wave per SIMD (512-register file);
//   * the workgroups are persistent (one per CU, XCD-contiguous tile order): the weight ring keeps streaming across tiles,
//     so only the first tile of a workgroup waits for LDS-DMA latency;
//   * ray_diff_fc (4 -> 16 -> 27) runs on the matrix pipe too (6 MFMAs) and each half-wave computes only the positional-encoding
//     octaves of its own k-slots (half 0: octaves 0-4, half 1: 5-9; three fp64 sin/cos + four fp64 double-angle steps per axis).
#include <string.h>
#include <utility>
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));

namespace {

#ifndef PF2_KO
#define PF2_KO 0   // knock-out bits for timing experiments (results are garbage): 1 no in-loop prologue, 2 no layer epilogues, 4 no LDS-DMA in the loop, 8 no attention, 256 no barriers in the tile loop (what the waves' skew costs), 512 the barrier of every odd region only (what a ring with one slot of slack could return at most), 16 the second cross-term matrix instruction of every slab skipped (its reads and fills stay): the matrix-pipe time fp6 cross terms would take
#endif
constexpr int KO = PF2_KO;
#ifndef PF2_MX_FP6
#define PF2_MX_FP6 1   // the cross terms of NL_PREC_F16MX: 1 = MX-FP6 (e2m3, 8 passes per K = 64: 1.5 matrix-instruction equivalents per product), 0 = MX-FP8 (e4m3, 16 passes: 2.0)
#endif
#ifndef PF2_L1_MX
#define PF2_L1_MX 1   // MX-FP6: layer 1 (K = 96: positional encoding + ray_diff_fc outputs) also as fp16 hi.hi + two fp6 cross terms (6 + 4 matrix instructions of 8 passes instead of 18)
#endif
constexpr bool L1MX = PF2_L1_MX;
constexpr int MXK = PF2_MX_FP6 ? 2 : 1;   // the one MX instance this library carries (pack and launch agree by construction)
#ifndef PF2_PE_F32
#define PF2_PE_F32 1   // the positional encoding's sin / cos recurrence of the f16mx instance in fp32 (round 6): 4.5e-6 on the encoding against the 3e-5 of its fp6
                       // cross-term image; 0 = fp64 like the other precisions (whose operands carry 1e-7)
#endif
#ifdef PF2_TRACE
__device__ unsigned long long pf2_trace[256];   // debug: cycle counter at every region start of one tile (block 0, wave 0)
#endif
constexpr int NBUF = 4;   // LDS ring slots; chunk c lives in slot c % 4, chunks c+1 .. c+3 are in flight while c is consumed

template <int NRT, bool X3, bool KEEP = false, bool MX6 = false>
struct Geo {
  static constexpr int W = 32 * NRT, PARTS = X3 ? 2 : 1, MPK = X3 ? 3 : 1;   // MPK: MFMAs per k-step
  static constexpr int KSL = 2 * NRT;        // k-steps of the wide layers (K = W)
  static constexpr int KS1 = 6;              // layer 1: 4 positional-encoding + 2 ray_diff_fc k-steps (K = 96)
  static constexpr int KS1I = X3 ? 6 : 8;    // k-steps COPIED per part for a layer-1 chunk (bf16: 8 so that the pieces divide by 4 waves)
  static constexpr int NC = 3 * NRT + 8;     // chunks (= output row tiles) per tile: L1 | L2 | L3 | K heads 0-3, V heads 0-3
  static constexpr int SLOT = PARTS * KSL * 64;   // ring slot in uint4
  static constexpr int cm(int g) { return ((g % NC) + NC) % NC; }
  static constexpr int layer(int g) { return cm(g) < NRT ? 0 : cm(g) < 2 * NRT ? 1 : cm(g) < 3 * NRT ? 2 : 3; }
  static constexpr int rt(int g) { return cm(g) < 3 * NRT ? cm(g) % NRT : cm(g) - 3 * NRT; }
  static constexpr int nks(int g) { return layer(g) == 0 ? KS1 : KSL; }
  static constexpr int ksi(int g) { return layer(g) == 0 ? KS1I : KSL; }       // k-steps per part in the LDS image
  // LDS-DMA pieces (1 KB) per wave.  MX-FP6, W = 256: a wide chunk has 28 of its 32 KB in use (16 K of f16 fragments + 4 slabs x 3 K of fp6 images; the weights' scale
  // dwords are resident) — 7 pieces per wave instead of 8 (W = 128: 14 of 16 KB do not divide by four waves: the whole chunk is copied)
  static constexpr int ppw(int g) { return layer(g) > 0 && MX6 && NRT == 8 ? 7 : PARTS * ksi(g) / 4; }
  static constexpr int gkb(int g) {   // offset of chunk g in the global stream, in KB; the stream always holds both parts
    int o = 0;
    for (int i = 0; i < cm(g); ++i) o += 2 * nks(i);
    return o;
  }
  static constexpr int cumks(int g) {
    int o = 0;
    for (int i = 0; i < g; ++i) o += nks(i);
    return o;
  }
  static constexpr int STREAM_KB = 2 * (NRT * KS1 + (2 * NRT + 8) * KSL);
  static constexpr int RES_RD = NBUF * SLOT;                 // resident: ray_diff_fc A fragments [layer][part][64] uint4
  static constexpr int RES_BIAS = RES_RD + 2 * PARTS * 64;   // resident: biases in accumulator order, floats [rd1 32 | rd2 32 | L2 W | L3 W]
  static constexpr int RES_ATT = RES_BIAS + (64 + 2 * W) / 4;   // attention weights [wave][head][32 rows] floats (wave-private)
  static constexpr int RES_SC = RES_ATT + 4 * 4 * 32 / 4;   // MX mode: E8M0 scale bytes of the chunks' fp8 weight images, ints [NC][2] = {w_hi8, w_lo8}
  static constexpr int RES_BND = RES_SC + (2 * NC + 3) / 4;   // MX mode: 8 floats — the bounds the activation block scales are derived from (pf2_mx_bounds_kernel)
  static constexpr int RES_SC6 = RES_BND + 2;   // MX-FP6: the wide chunks' weight-scale dwords [chunk 2 NRT + 8][lane 64] x {w_hi6, w_lo6} (pack_point_mx6_kernel)
  static constexpr int LDS_U4 = RES_SC6 + (MX6 ? (L1MX ? NC : 2 * NRT + 8) * 32 : 0);
  // micro-steps of a finished chunk's epilogue: layers: 8 pairs x (LeakyReLU + hi | lo); k head: 9; v head: 4 x (4 sums + store)
  // (KEEP: + 4 row stores of a k head / 4 x 4 dword stores of a v head: the rows nl_attn_backward reads)
  static constexpr int epi_steps(int c) { return layer(c) < 3 ? 8 * (X3 ? 2 : 1) + (MX6 && (rt(c) & 1) ? 2 : 0) : (rt(c) < 4 ? 10 : 10) + (KEEP ? 4 : 0); }
  static constexpr int RL = cumks(NC) % 3 == 0 ? 3 : 4;   // A-fragment register ring (3 k-steps are live)
  static constexpr int rpos(int runks) { return runks % RL; }
  static_assert(NC % NBUF == 0 && cumks(NC) % RL == 0, "ring positions must be tile-periodic");
  static_assert((PARTS * KS1I) % 4 == 0 && (PARTS * KSL) % 4 == 0, "pieces per wave");
};

template <int OFF>
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, OFF, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {   // compile-time loop: every index is a constant expression
  static_for_impl(std::make_integer_sequence<int, (N > 0 ? N : 0)>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // low half = bf16(a), high half = bf16(b), round to nearest even
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {   // bare v_max_f32 (fmaxf adds a canonicalising v_max x,x)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// LeakyReLU of a pair + its bf16 hi word, one statement (no compiler-inserted boundary nops between the five instructions)
__device__ __forceinline__ unsigned lrelu_hi2(float& v0, float& v1) {
  unsigned hi; float t0, t1;
  asm("v_mul_f32 %3, 0x3c23d70a, %1\n\tv_mul_f32 %4, 0x3c23d70a, %2\n\tv_max_f32 %1, %1, %3\n\tv_max_f32 %2, %2, %4\n\tv_cvt_pk_bf16_f32 %0, %1, %2"
      : "=&v"(hi), "+v"(v0), "+v"(v1), "=&v"(t0), "=&v"(t1));
  return hi;
}
// lo word of a pair: bf16(v - float(hi))
__device__ __forceinline__ unsigned lo2(float v0, float v1, unsigned hi) {
  unsigned lo; float t0, t1;
  asm("v_lshlrev_b32 %1, 16, %5\n\tv_and_b32 %2, 0xffff0000, %5\n\tv_sub_f32 %1, %3, %1\n\tv_sub_f32 %2, %4, %2\n\tv_cvt_pk_bf16_f32 %0, %1, %2"
      : "=&v"(lo), "=&v"(t0), "=&v"(t1) : "v"(v0), "v"(v1), "v"(hi));
  return lo;
}
// ---- "fp16 hi.hi + two MX-FP8 cross terms" (MX mode): hi = f16(v) (v_cvt_pk_f16_f32, saturating under MODE.FP16_OVFL),
// lo = v - hi exactly (v_fma_mix_f32 reads the f16 half directly), fp8 images hi8 = e4m3(hi), lo8 = e4m3(lo * 2^11) (the matrix instruction's
// block scale 2^-11 undoes the factor); two bytes land in the low (HALF = 0) or high (HALF = 1) half of their dword.
__device__ __forceinline__ unsigned lrelu_hi2_f16(float& v0, float& v1) {
  unsigned hi; float t0, t1;
  asm("v_mul_f32 %3, 0x3c23d70a, %1\n\tv_mul_f32 %4, 0x3c23d70a, %2\n\tv_max_f32 %1, %1, %3\n\tv_max_f32 %2, %2, %4\n\tv_cvt_pk_f16_f32 %0, %1, %2"
      : "=&v"(hi), "+v"(v0), "+v"(v1), "=&v"(t0), "=&v"(t1));
  return hi;
}
// sc_hi / sc_lo: the lane's (= the row's) power-of-two block scales of the two images (both conversions DIVIDE by their scale operand)
// ... + the running row maximum of |value| in the same statement (an asm boundary costs a compiler-inserted s_nop: 120 per tile as a separate v_max3)
__device__ __forceinline__ unsigned lrelu_hi2_f16_amax(float& v0, float& v1, float& m) {
  unsigned hi; float t0, t1;
  asm("v_mul_f32 %3, 0x3c23d70a, %1\n\tv_mul_f32 %4, 0x3c23d70a, %2\n\tv_max_f32 %1, %1, %3\n\tv_max_f32 %2, %2, %4\n\tv_max3_f32 %5, |%1|, |%2|, %5\n\tv_cvt_pk_f16_f32 %0, %1, %2"
      : "=&v"(hi), "+v"(v0), "+v"(v1), "=&v"(t0), "=&v"(t1), "+v"(m));
  return hi;
}
template <int HALF>
__device__ __forceinline__ void mx_bytes2(float v0, float v1, unsigned hi, unsigned& h8, unsigned& l8, float sc_hi, float sc_lo) {
  float t0, t1;
  if (HALF == 0)
    asm("v_fma_mix_f32 %2, %6, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %6, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_scalef32_pk_fp8_f16 %0, %6, %8\n\tv_cvt_scalef32_pk_fp8_f32 %1, %2, %3, %7"
        : "+v"(h8), "+v"(l8), "=&v"(t0), "=&v"(t1) : "v"(v0), "v"(v1), "v"(hi), "v"(sc_lo), "v"(sc_hi));
  else
    asm("v_fma_mix_f32 %2, %6, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %6, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_scalef32_pk_fp8_f16 %0, %6, %8 op_sel:[0,0,1]\n\tv_cvt_scalef32_pk_fp8_f32 %1, %2, %3, %7 op_sel:[0,0,0,1]"
        : "+v"(h8), "+v"(l8), "=&v"(t0), "=&v"(t1) : "v"(v0), "v"(v1), "v"(hi), "v"(sc_lo), "v"(sc_hi));
}
// running row maximum of |activation| (MX mode: what the next layer's block scale is bounded from)
__device__ __forceinline__ void amax2(float& m, float v0, float v1) {
  asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(v0), "v"(v1));
}

// hi/lo split of a pair: hi = bf16(v), lo = bf16(v - float(hi))
template <bool X3>
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
  hi = cvt_pk_bf16(v0, v1);
  if (X3) {
    const float f0 = __uint_as_float(hi << 16), f1 = __uint_as_float(hi & 0xffff0000u);
    lo = cvt_pk_bf16(v0 - f0, v1 - f1);
  }
}

// split-FP16 variants (F16 mode: the gradient path's forward, 2^-22 products): hi = f16(v), lo = f16(v - hi)
__device__ __forceinline__ unsigned lo2_f16(float v0, float v1, unsigned hi) {
  unsigned lo; float t0, t1;
  asm("v_fma_mix_f32 %1, %5, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %5, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_cvt_pk_f16_f32 %0, %1, %2"
      : "=&v"(lo), "=&v"(t0), "=&v"(t1) : "v"(v0), "v"(v1), "v"(hi));
  return lo;
}
// f16 pair of two values + the running maximum of their magnitudes (layer-1 operands of the MX-FP6 path: no activation in between)
__device__ __forceinline__ unsigned hi2_f16_amax(float v0, float v1, float& m) {
  unsigned hi;
  asm("v_max3_f32 %1, |%2|, |%3|, %1\n\tv_cvt_pk_f16_f32 %0, %2, %3" : "=&v"(hi), "+v"(m) : "v"(v0), "v"(v1));
  return hi;
}
// residuals of a pair as floats: v - float(hi half) (exact)
__device__ __forceinline__ void lo2_f32(float v0, float v1, unsigned hi, float& l0, float& l1) {
  asm("v_fma_mix_f32 %0, %4, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %4, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l0), "=&v"(l1) : "v"(v0), "v"(v1), "v"(hi));
}
// The two fp6 packing conversions as asm statements with EARLY-CLOBBER results: hipcc 7.2 lets the builtins' 6-register result overlap the scale operand (seen:
// v_cvt_scalef32_2xpk16_fp6_f32 v[206:211], v[122:137], v[138:153], v206), and the multi-pass instruction then reads a scale it has already overwritten — one slab of
// one layer came out with garbage residuals (found with tools/mx6_debug.py: only K slab 2 of base_mlp.4 was off).
__device__ __forceinline__ u32x6 cvt_pk32_fp6_f16(u32x16 h, float sc) {
  u32x6 r;
  asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(r) : "v"(h), "v"(sc));
  return r;
}
__device__ __forceinline__ u32x6 cvt_2xpk16_fp6_f32(f32x16 a, f32x16 b, float sc) {
  u32x6 r;
  asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(sc));
  return r;
}
template <bool X3, bool F16>
__device__ __forceinline__ void split2f(float v0, float v1, unsigned& hi, unsigned& lo) {
  if constexpr (F16) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    lo = lo2_f16(v0, v1, hi);
  } else split2<X3>(v0, v1, hi, lo);
}

// branch-free sin/cos in fp64 (|x| up to ~1e5): Cody-Waite reduction to [-pi/4, pi/4] + Taylor (error < 1e-11)
__device__ __forceinline__ void sincos_d(double x, double& s, double& c) {
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

struct Pf2Scalars { int dir_stride, dir_div; unsigned dir_magic; int dir_shift; unsigned dir_one; int N, M; float inv_span; int ntiles; unsigned t_bytes;
                    const float* tmax;      // MX mode: max |T| over the frame's table (one float, written by nl_table_absmax), or null (= 0)
                    unsigned* logit_amax;
                    unsigned long long* clk; };   // optional: [shader cycles, 100-MHz reference ticks] workgroup 0 spent in this launch (bench.py: the clock under load)   // optional: running maximum of |attention logit| (q.k / sqrt d_k) as float bits, atomicMax'ed once per wave (nl_frame_diagnostics)

// MX (with X3) = 1, round 4: layer 1 stays three-term split-bf16 (K = 96, issue-bound anyway); layers 2, 3 and the k / v projections multiply as
// fp16 hi.hi + fp8(lo).fp8(hi) + fp8(hi).fp8(lo): per K = 64 slab 4 x v_mfma_f32_32x32x16_f16 + 2 x v_mfma_scale_f32_32x32x64_f8f6f4 instead of 12 bf16 MFMAs.
// MX = 2, round 5 (the instance the library carries, PF2_MX_FP6): the same two cross terms on fp6 (e2m3) operands — 8 passes instead of 16 — with a power-of-two scale per
// 32-value block taken from the values themselves; every layer, layer 1 included (PF2_L1_MX).  DESIGN.md 2.4.
// F16 (with X3, without MX): every layer in three-term split-FP16 (2^-22 products: what the gradient path's forward needs so that its LeakyReLU sign decisions are the
// fp32 function's).  KEEP: the kernel also leaves what the frozen-weight way back reads — the k / v rows (N x 8, 256) and the SIGN of the three
// layers' outputs as bits in the streaming GEMM's ep_maskin layout ([32-row tile][lane][4 dwords]) — so that the staged forward of the branch (an encode kernel,
// four (N x 8)-row GEMMs through HBM, an attention kernel) is one launch.
struct Pf2Keep { float* kv; unsigned* mk[3]; unsigned kv_bytes, mk_bytes; };
template <int NRT, bool X3, int MX, bool F16 = false, bool KEEP = false>
__global__ __launch_bounds__(256, 1) void point_fused2_kernel(
    const float* __restrict__ p_xyz, const float* __restrict__ p_dir, const int* __restrict__ p_idx, const float* __restrict__ p_Q,
    float* __restrict__ p_O, const float* __restrict__ p_ptt, const float* __restrict__ p_sp_xyz,
    const float* __restrict__ p_sp_dir, const uint4* __restrict__ p_wstream, const Pf2Scalars sc, const Pf2Keep keep) {
  static_assert(!MX || X3, "the MX mode extends the three-term mode");
  static_assert(!F16 || (X3 && !MX), "split-FP16 is a three-term mode");
  static_assert(!KEEP || F16, "the kept masks must come from the split-FP16 forward");
  constexpr bool MX6 = MX == 2, MX8 = MX == 1;   // cross terms on MX-FP6 (e2m3) / MX-FP8 (e4m3)
  using GG = Geo<NRT, X3, KEEP, MX6>;
  constexpr int W = GG::W, PARTS = GG::PARTS, MPK = GG::MPK, NC = GG::NC, SLOT = GG::SLOT;
  if (MX || F16) __builtin_amdgcn_s_setreg(1473, 1);   // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: f16 / fp8 conversions saturate instead of producing inf / NaN
  // ONE __shared__ object, read through ONE native vector type with compile-time slot indices: hipcc then keeps the alias
  // information that lets SIInsertWaitcnts leave LDS reads alone while LDS-DMA writes are in flight
  __shared__ uint4 lds_all[GG::LDS_U4];
  // upper 64 KB of the ring through ONE opaque base (every offset fits ds_read's 16-bit immediate): without it the compiler keeps ~60 separate "base + constant"
  // addresses in AGPRs and re-reads one before every second fragment read (344 of the tile's 1 018 v_accvgpr_read_b32; 8 068 -> 7 565 instructions per tile, kernel -2.3 %).
  // (Still the same __shared__ object for the alias analysis: the waits around the LDS-DMA writes stay as they were — checked in the ISA.)
  typedef __attribute__((address_space(3))) uint4 lds_u4;
  lds_u4* lds_hi = (lds_u4*)lds_all + 4096 + (threadIdx.x & 63);
  asm volatile("" : "+v"(lds_hi));
  // (MX-FP6: the 8-byte tails of the fp6 images — lane stride 8 — through a second opaque base of the same kind)
  typedef __attribute__((address_space(3))) u32x2 lds_u2;
  lds_u2* lds_hi8 = (lds_u2*)((lds_u4*)lds_all + 4096) + (threadIdx.x & 63);
  if constexpr (MX6) asm volatile("" : "+v"(lds_hi8));
  lds_u2* lds_res8 = (lds_u2*)((lds_u4*)lds_all + NBUF * SLOT) + (threadIdx.x & 63);   // ... and the resident block behind the ring
  if constexpr (MX6) asm volatile("" : "+v"(lds_res8));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31, kk = j & 7;
  const unsigned nwg = gridDim.x;
  int tile = (int)nl_xcd_block();
  if (tile >= sc.ntiles) return;
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();   // (s_memtime: shader clock; s_memrealtime: 100 MHz)

  // ---------------------------------------------------------------- resident block: ray_diff_fc fragments + bias tables
  {
    const uint4* src = p_wstream + (size_t)GG::STREAM_KB * 64;
    for (int i = tid; i < 2 * PARTS * 64; i += 256) lds_all[GG::RES_RD + i] = src[(i / (PARTS * 64)) * 128 + i % (PARTS * 64)];
    for (int i = tid; i < (64 + 2 * W) / 4; i += 256) lds_all[GG::RES_BIAS + i] = src[256 + i];
    if (MX8) for (int i = tid; i < (2 * NC + 3) / 4 + 2; i += 256) lds_all[GG::RES_SC + i] = src[256 + (64 + 2 * W) / 4 + i];   // + the bounds block
    if (MX6) for (int i = tid; i < (L1MX ? NC : 2 * NRT + 8) * 32; i += 256) lds_all[GG::RES_SC6 + i] = src[256 + (64 + 2 * W) / 4 + (2 * NC + 3) / 4 + 2 + i];
  }
  __syncthreads();

  // ---------------------------------------------------------------- weight stream: piece p = 4 i + wave of a chunk
  // All memory traffic of the tile loop goes through BUFFER instructions.  (a) LDS-DMA issued as global_load_lds is a FLAT
  // instruction that touches two address spaces; while one is pending SIInsertWaitcnts turns every wait for an ordinary load
  // into vmcnt(0) — eight full drains of the weight ring per tile.  buffer_load ... lds is counted like any other load, so
  // the compiler's waits stay exact.  (b) voffset is one VGPR for all pieces, the piece offset is a scalar: no per-piece
  // 64-bit addresses to hoist out of the loop (LICM did: 450 registers / SGPR spills).  (c) out-of-range offsets are dropped
  // by the bounds check: predicated stores without EXEC juggling or branches.
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p_wstream, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)p_ptt, 0, (int)sc.t_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void*)p_Q, 0, sc.N * 512, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)p_O, 0, sc.N * 512, 0x00020000);
  // KEEP: k / v rows and the three layers' sign bits (rows / tiles past the end are dropped by the bounds check)
  const __amdgpu_buffer_rsrc_t rKV = __builtin_amdgcn_make_buffer_rsrc((void*)keep.kv, 0, KEEP ? (int)keep.kv_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM0 = __builtin_amdgcn_make_buffer_rsrc((void*)keep.mk[0], 0, KEEP ? (int)keep.mk_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM1 = __builtin_amdgcn_make_buffer_rsrc((void*)keep.mk[1], 0, KEEP ? (int)keep.mk_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM2 = __builtin_amdgcn_make_buffer_rsrc((void*)keep.mk[2], 0, KEEP ? (int)keep.mk_bytes : 0, 0x00020000);
  unsigned kmask[4] = {0u, 0u, 0u, 0u};
  int tile_prev = 0;
  // byte offsets into the k / v rows: the wave's first row of the tile, this lane's row.  (The first tile's region 0 runs the "previous tile's" last v head on
  // whatever the accumulator holds: out of range = dropped, like ooff_prev)
  unsigned kvtile = 0, kvtile_prev = 0x80000000u, kvrow = 0;
  // Wave w loads the ppw(c) CONSECUTIVE 1-KB pieces [w ppw, (w + 1) ppw) of a chunk, in groups of four: ONE scalar stream offset and ONE M0 (LDS base) per group, the
  // piece inside its group through the instruction's 12-bit offset, which the hardware adds to the memory address AND to the LDS address (round 5: pieces p = 4 i + w lay
  // 4 KB apart, every one of a tile's 216 needed its own s_mov m0 + s_addk: 432 of the tile loop's ~6 800 issue slots; now ~110)
  const unsigned wvoff = lane * 16;
  constexpr int PPW_L1 = GG::ppw(0), PPW_W = GG::ppw(NRT);   // pieces per wave: layer-1 chunks, wide chunks
  typedef __attribute__((address_space(3))) uint4 lds_u4q;
  lds_u4q* const lwL1 = (lds_u4q*)lds_all + wave * (PPW_L1 * 64);
  lds_u4q* const lwW = (lds_u4q*)lds_all + wave * (PPW_W * 64);
  const unsigned wdelta = (unsigned)wave * (unsigned)((PPW_W - PPW_L1) * 1024);   // a wave's offset inside a wide chunk minus its offset inside a layer-1 chunk
  unsigned soff = 0;                                // running stream offset of the current group (scalar), the wave's share included
  auto dma_piece = [&](auto Cc, auto Ic) __attribute__((always_inline)) {
    constexpr int c = GG::cm(decltype(Cc)::value), i = decltype(Ic)::value, g = i / 4, r = i % 4;
    if constexpr (r == 0) {
      constexpr int cp = g == 0 ? GG::cm(c - 1) : c, gp = g == 0 ? (GG::ppw(c - 1) - 1) / 4 : g - 1;   // the group issued before this one (gkb wraps: chunk -1 = NC-1)
      constexpr int want = GG::gkb(c) * 1024 + g * 4096, prev = GG::gkb(cp) * 1024 + gp * 4096;
      soff += (unsigned)(want - prev);   // groups are issued in stream order: one scalar add per group
      if constexpr (g == 0 && GG::ppw(c) > GG::ppw(c - 1)) soff += wdelta;
      if constexpr (g == 0 && GG::ppw(c) < GG::ppw(c - 1)) soff -= wdelta;
      asm volatile("" : "+s"(soff));     // opaque, so that the offsets are not re-materialised (and hoisted) as constants
    }
    lds_u4q* const base = GG::ppw(c) == PPW_L1 ? lwL1 : lwW;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + (c % NBUF) * SLOT + g * 256), 16, wvoff, soff, r * 1024, 0);
  };

  // ---------------------------------------------------------------- register state
  // Activations as B fragments (hi / lo), ping-pong between layers, and the layer-1 operands of the current tile.  Kept as SCALAR
  // dwords (assembled into a 4-dword operand at the MFMA): a fragment is written one dword at a time, and with vector-typed
  // storage every such insert keeps the whole old vector alive in the compiler's eyes — both ping-pong halves then stay live
  // around the tile loop (512 registers + 290 spills instead of ~400)
  unsigned Xh[2][2 * NRT][4], Xl[2][2 * NRT][4];
  // MX mode: Xh holds f16 pairs; the fp8 images of a slab (4 k-steps) as the matrix instruction wants them: [buffer][slab][0 = hi8, 1 = lo8][8 dwords], byte
  // u = 8 s + t of a lane <-> element t of k-step 4 q + s (the weight images use the same map: any bijection works as long as both operands share it)
  unsigned X8[2][NRT / 2][2][8];
  u32x4 w8[2][4];   // fp8 A operands of a slab, double-buffered by slab parity: [0..1] = w_hi8 (32 bytes per lane), [2..3] = w_lo8
  // ---- MX-FP6 (MX == 2, round 5): the two cross terms on e2m3 elements, 8 passes per K = 64 instead of 16 (profiles/ubench_mx6_rowtile.txt: conversions, layouts, timing).
  // A lane's 32 values of a slab are ONE MX block with its own power-of-two scale — 2^(floor(log2 max) - 2), from the values themselves: the largest lands in [4, 8) of e2m3's
  // 7.5 — for the activations (per row, slab and image; the residual image takes the hi image's scale x 2^-11) and for the weights (per output row and half-slab, from the
  // packing kernel: byte q of the chunk's two scale dwords).  The f16 B fragments of a slab live in ONE 16-register vector (Xh16): v_cvt_scalef32_pk32_fp6_f16 packs the hi
  // image from it in one instruction, the matrix instructions read its 4-register quarters; v_cvt_scalef32_2xpk16_fp6_f32 packs the residuals of two row tiles (interleaved:
  // position 2 i <- first operand, 2 i + 1 <- second; the weights' hi image is stored in that order).
  u32x16 Xh16[2][NRT / 2];              // [buffer][slab]: dword 4 s + d = fragment dword d of k-step 4 q + s
  unsigned X6[2][NRT / 2][2][6];        // [buffer][slab][0 = hi6, 1 = lo6][6 dwords]: position P = bits 6 P .. 6 P + 5
  u32x4 w6a[2][2]; u32x2 w6b[2][2];     // fp6 A operands of a slab, [slab parity][0 = w_hi6, 1 = w_lo6]: dwords 0-3 | 4-5
  u32x2 wsc6 = {0u, 0u};                // the chunk's weight-scale dwords {w_hi6, w_lo6}: byte q = E8M0 of slab q for this lane's row and K half
  unsigned xs6h[2] = {0u, 0u}, xs6l[2] = {0u, 0u};   // activation scale bytes [buffer]: byte q = slab q (hi image | residual image = hi - 11)
  unsigned hp6[2][8];                   // f16 pairs of the two row tiles a slab is made of (until the slab is complete)
  float lo6t[2][16];                    // their residuals
  float scf6 = 1.f;
  // layer 1 in the same arithmetic (L1MX): its B operands of the tile — slab 0 = k-steps 0-3 (positional encoding + raw offsets), slab 1 = k-steps 4, 5 (ray_diff_fc outputs) + two
  // empty k-steps — as f16 fragments (P16), fp6 images (P6[slab][0 = hi, 1 = residual]) and the two scale-byte registers; pp* / plo*: pairs and residuals until a slab is complete
  u32x16 P16[2];
  unsigned P6[2][2][6];
  unsigned pxs6h = 0u, pxs6l = 0u;
  unsigned pp0[16], pp1[8];
  float plo0[32], plo1[16];
  float pam0 = 0.f, pam1 = 0.f;
  const int* ssc = reinterpret_cast<const int*>(lds_all + GG::RES_SC);
  // MX mode, round 5: the activations' fp8 images carry a block scale PER ROW AND LAYER instead of the constants 1 / 2^-11 (whose window was |a| = 2^-6 ... 448:
  // beyond it the cross terms saturated and the product fell back to single-fp16 accuracy — tools/scale_sweep.py found it with the feature maps x 8).  The scale of
  // layer L's output rows is fixed BEFORE the layer runs, from a bound that cannot be exceeded: |out| <= max_c ||w_c||_1 . max|in| + max|b| with max|in| the row's
  // running maximum over the previous layer's outputs (one v_max3 per finished pair), layer 1 from max|T| over the frame's table + the L1 norms of its
  // positional-encoding / ray-difference columns.  The bound is ~50x loose, which e4m3's 15 binades absorb (values land at <= 256 of 448; what falls below
  // the normal range is < 0.4 % of the row's largest value).  xs_hi / xs_lo: the floats the conversions divide by, xs_e8h / xs_e8l: the E8M0 bytes the matrix
  // instruction reads for the lane's row (both halves of a row hold the same values), indexed by the ping-pong buffer the rows live in.
  float xs_hi[2] = {1.f, 1.f}, xs_lo[2] = {0.00048828125f, 0.00048828125f};
  int xs_e8h[2] = {127, 127}, xs_e8l[2] = {116, 116};
  float amax = 0.f, cur_b1 = 0.f, pn_b1 = 0.f;
  float lmax = 0.f;   // largest |attention logit| this lane has scored: the softmax over nearly tied neighbours turns a logit error e into a weight error ~e, and the
                      // logit error is (relative product error) x |logit| — the conditioning indicator nl_frame_diagnostics reports
  float bnd_c1a = 0.f, bnd_c1x = 0.f, bnd_B2 = 0.f, bnd_bm2 = 0.f, bnd_B3 = 0.f, bnd_bm3 = 0.f, bnd_t = 0.f;
  if constexpr (MX8) {
    const float* bf = reinterpret_cast<const float*>(lds_all + GG::RES_BND);
    auto uni = [](float v) __attribute__((always_inline)) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    bnd_c1a = uni(bf[0]); bnd_c1x = uni(bf[1]); bnd_B2 = uni(bf[2]); bnd_bm2 = uni(bf[3]); bnd_B3 = uni(bf[4]); bnd_bm3 = uni(bf[5]);
    bnd_t = sc.tmax ? uni(*sc.tmax) : 0.f;
  }
  // bound (>= the largest |value| of the lane's row) -> the scales of buffer `buf`: 2^(e - 8) with bound < 2^e, i.e. bound / scale < 256
  auto set_scale = [&](auto Bc, float bound) __attribute__((always_inline)) {
    constexpr int buf = decltype(Bc)::value;
    int eb = __builtin_amdgcn_frexp_expf(bound) + 119;
    eb = eb < 12 ? 12 : (eb > 254 ? 254 : eb);
    xs_hi[buf] = __builtin_bit_cast(float, eb << 23); xs_lo[buf] = __builtin_bit_cast(float, (eb - 11) << 23);
    xs_e8h[buf] = eb; xs_e8l[buf] = eb - 11;
  };
  auto row_amax = [&]() __attribute__((always_inline)) {   // both halves of a row: lane ^ 32 holds the other 16 channels of every row tile
    const float o = __shfl_xor(amax, 32, 64);
    return fmaxf(amax, o);
  };
  unsigned Ph[GG::KS1][4], Pl[GG::KS1][4];
  u32x4 frh[GG::RL], frl[GG::RL];         // A-fragment ring, position = (running k-step) % RL
  // accumulator of chunk c = acc[c & 3].  Four, because the accumulator is INITIALISED by loads that must be in flight early:
  // while region G accumulates into acc[G & 3] and the epilogue of G-1 drains acc[(G-1) & 3], the bias slice of chunk G+1
  // (LDS) and the table-row slice of layer-1 chunk G+2 (a gather from the per-frame table T) land in the other two
  f32x16 acc[4];
  f32x4 Qr[4];                            // query of the head being scored
  float att[4] = {0.f, 0.f, 0.f, 0.f};
  float ev0 = 0.f, ev1 = 0.f, ap = 0.f, amx = 0.f, aee = 0.f, ase = 0.f;
  unsigned ehi = 0;
  float ov[4];
  f32x4 aw[4];
  unsigned ooff_cur = 0x80000000u, ooff_prev = 0x80000000u;   // byte offset of the lane's slice of O; out of range (= dropped) for lanes that do not store
  unsigned toff = 0, qoff = 0;                                // byte offsets of the lane's table row / query slice
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* sres = reinterpret_cast<const float*>(lds_all + GG::RES_BIAS);
  float* satt = reinterpret_cast<float*>(lds_all + GG::RES_ATT) + wave * 128;

  auto mfma = [](const u32x4& a, const u32x4& b, const f32x16& c) __attribute__((always_inline)) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  auto mfma_h = [](const u32x4& a, const u32x4& b, const f32x16& c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  };
  // fp8 (e4m3) x fp8, K = 64; sa / sb: E8M0 exponent of the operand's block scale (one value for every block here)
  auto mfma_8 = [](const i32x8& a, const i32x8& b, const f32x16& c, int sa, int sb) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  };
  auto cat8 = [](const u32x4& a, const u32x4& b) __attribute__((always_inline)) {
    return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
  auto frag8 = [](const unsigned (&d)[8]) __attribute__((always_inline)) {
    return i32x8{(int)d[0], (int)d[1], (int)d[2], (int)d[3], (int)d[4], (int)d[5], (int)d[6], (int)d[7]};
  };

  // LeakyReLU + split of a finished pair -> dword `d` of the destination fragments
  auto finish_pair = [&](float v0, float v1, unsigned& dh, unsigned& dl) __attribute__((always_inline)) {
    v0 = vmax(v0, v0 * 0.01f); v1 = vmax(v1, v1 * 0.01f);
    unsigned h = 0, l = 0;
    split2f<X3, F16>(v0, v1, h, l);
    dh = h;
    if (X3) dl = l;
  };
  auto frag4 = [](const unsigned (&d)[4]) __attribute__((always_inline)) { return u32x4{d[0], d[1], d[2], d[3]}; };

  // ---------------------------------------------------------------- tile prologue: layer-1 operands (model.py:394-409)
  // Written as a list of micro-steps so that the NEXT tile's operands are produced in the MFMA shadow of the current tile's k / v
  // regions (the other ping-pong half of X is dead there, so the registers are free); the first tile runs the list back to back.
  // All loads are unconditional: rows that must read as zero (k >= M: knn_gather zero-fill, knn_utils.py:211-220; rows past N)
  // carry an out-of-range buffer offset.
  const __amdgpu_buffer_rsrc_t rSX = __builtin_amdgcn_make_buffer_rsrc((void*)p_sp_xyz, 0, sc.M * 12, 0x00020000);
  const __amdgpu_buffer_rsrc_t rSD = __builtin_amdgcn_make_buffer_rsrc((void*)p_sp_dir, 0, sc.M * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)p_dir, 0, p_dir ? 0x7fffffff : 0, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  int pn_tile = tile, pn_nn = 0, pn_id = 0;
  unsigned pn_ooff = OOB, pn_toff = 0, pn_qoff = 0;
  bool pn_have = false, pn_live = false;
  float pq[3], pdv[3], pdv0[3], pp[3], pnd[3], poff[3], prd[4];
  unsigned rdbh[2] = {0, 0}, rdbl[2] = {0, 0}, hidh[4] = {0, 0, 0, 0}, hidl[4] = {0, 0, 0, 0};
  f32x16 pa = zero16;
  f32x4 pb[4];
  using pe_t = std::conditional_t<(MX6 && L1MX && PF2_PE_F32), float, double>;   // fp64 FMAs issue at the fp32 rate on this part: what fp32 returns is registers (24 instead of 48)
  constexpr pe_t PE_PIO2_HI = sizeof(pe_t) == 4 ? (pe_t)1.57079637050628662 : (pe_t)1.5707963267948966;
  constexpr pe_t PE_PIO2_LO = sizeof(pe_t) == 4 ? (pe_t)-4.37113882867379e-8 : (pe_t)6.123233995736766e-17;
  pe_t sx[3], skd[3], sr[3], sr2[3], su[3], sw[3], ss[3], scs[3];
  int skq[3];
  constexpr int NPL = 2, NPC = (MX6 && L1MX) ? 57 : 55, NPRO = NPL + NPC;   // load steps, compute steps (L1MX: + the two slabs' packing conversions)
  auto pro_step = [&](auto Ic) __attribute__((always_inline)) {
    constexpr int I = decltype(Ic)::value;
    if constexpr (I == 0) {   // ---- loads that depend on the tile number only
      const int n = pn_tile * 16 + wave * 4 + (j >> 3);
      const bool live = n < sc.N;
      pn_nn = live ? n : sc.N - 1;
      // O slice of the transposed v heads: lane = output dim (lane & 31) of the wave's four samples, half 0 stores
      pn_ooff = hh ? OOB : (unsigned)(pn_tile * 16 + wave * 4) * 512u + 4u * j;
      pn_qoff = (unsigned)pn_nn * 512u + 16u * hh;
      pn_live = live;
      pn_have = live && kk < sc.M && sc.M > 0;
      pn_id = p_idx[(size_t)pn_nn * 8 + kk];
      pq[0] = p_xyz[3 * (size_t)pn_nn]; pq[1] = p_xyz[3 * (size_t)pn_nn + 1]; pq[2] = p_xyz[3 * (size_t)pn_nn + 2];
      const unsigned ray = (__umulhi((unsigned)pn_nn, sc.dir_magic) >> sc.dir_shift) + (unsigned)pn_nn * sc.dir_one;   // branch-free n / dir_div
      const unsigned dro = ray * (unsigned)sc.dir_stride * 4u;   // out of range (reads 0) when there is no direction array
      pdv[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, dro, 0, 0));
      pdv[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, dro + 4, 0, 0));
      pdv[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, dro + 8, 0, 0));
    } else if constexpr (I == 1) {   // ---- gathers that depend on the neighbour index
      pn_toff = (unsigned)(pn_have ? pn_id : sc.M) * (unsigned)(W * 4) + 64u * hh;   // row M holds the bias alone
      const unsigned so = pn_have ? (unsigned)pn_id : OOB / 16;
      pp[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rSX, so * 12, 0, 0));
      pp[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rSX, so * 12 + 4, 0, 0));
      pp[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rSX, so * 12 + 8, 0, 0));
      const f32x4 nd = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rSD, so * 16, 0, 0));
      pnd[0] = nd[0]; pnd[1] = nd[1]; pnd[2] = nd[2];
      // model.py:391-392: without a direction array the nearest neighbour's direction is used (first lane of the sample's 8)
      const int i0 = __shfl(pn_id, lane & ~7, 64);
      const bool ok0 = !p_dir && pn_live && sc.M > 0;
      const unsigned so0 = ok0 ? (unsigned)i0 * 16u : OOB;
      const f32x4 d0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rSD, so0, 0, 0));
      pdv0[0] = d0[0]; pdv0[1] = d0[1]; pdv0[2] = d0[2];
    } else {
      constexpr int C = I - NPL;
      if constexpr (C == 0) {
        pdv[0] += pdv0[0]; pdv[1] += pdv0[1]; pdv[2] += pdv0[2];   // exactly one of the two sources is non-zero
        poff[0] = (pq[0] - pp[0]) * sc.inv_span; poff[1] = (pq[1] - pp[1]) * sc.inv_span; poff[2] = (pq[2] - pp[2]) * sc.inv_span;
        prd[0] = pdv[0] - pnd[0]; prd[1] = pdv[1] - pnd[1]; prd[2] = pdv[2] - pnd[2];
        prd[3] = pdv[0] * pnd[0] + pdv[1] * pnd[1] + pdv[2] * pnd[2];
        if constexpr (MX8) {   // |base_mlp.0 row| <= max|T| + sum |w| over the sin / cos / ray-difference columns + sum |w| over the raw-offset columns x max |offset|
          float mo = fabsf(poff[0]);
          amax2(mo, poff[1], poff[2]);
          pn_b1 = fmaf(bnd_c1x, mo, bnd_t + bnd_c1a);
        }
      } else if constexpr (C == 1) {
        pq[0] = sqrtf(prd[0] * prd[0] + prd[1] * prd[1] + prd[2] * prd[2]) + 1e-8f;   // pq[0] is free now: the norm
      } else if constexpr (C == 2) { prd[0] /= pq[0]; prd[1] /= pq[0]; }
      else if constexpr (C == 3) {
        prd[2] /= pq[0];
        unsigned h0 = 0, l0 = 0, h1 = 0, l1 = 0;
        split2f<X3, F16>(prd[0], prd[1], h0, l0); split2f<X3, F16>(prd[2], prd[3], h1, l1);
        rdbh[0] = hh ? 0u : h0; rdbh[1] = hh ? 0u : h1;   // k-slots 0..3 of half 0
        if (X3) { rdbl[0] = hh ? 0u : l0; rdbl[1] = hh ? 0u : l1; }
      } else if constexpr (C == 4 || C == 7) {   // ray_diff_fc layers on the matrix pipe (model.py:36-39)
        constexpr int l = C == 4 ? 0 : 1;
        const u32x4 xh = l == 0 ? u32x4{rdbh[0], rdbh[1], 0u, 0u} : frag4(hidh);
        const u32x4 xl = l == 0 ? u32x4{rdbl[0], rdbl[1], 0u, 0u} : frag4(hidl);
        const u32x4 ah = __builtin_bit_cast(u32x4, lds_all[GG::RES_RD + (l * PARTS + 0) * 64 + lane]);
        if (X3) {
          const u32x4 al = __builtin_bit_cast(u32x4, lds_all[GG::RES_RD + (l * PARTS + (PARTS - 1)) * 64 + lane]);
          pa = mfma(al, xh, zero16);
          pa = mfma(ah, xl, pa);
          pa = mfma(ah, xh, pa);
        } else pa = mfma(ah, xh, zero16);
        static_for<l == 0 ? 2 : 4>([&](auto Gc) __attribute__((always_inline)) {
          constexpr int g = decltype(Gc)::value;
          pb[g] = *reinterpret_cast<const f32x4*>(sres + 32 * l + 16 * hh + 4 * g);
        });
      } else if constexpr (C == 5 || C == 6) {   // hidden layer: registers 0..7 = units m(r, hh) < 16
        constexpr int g = C - 5;
        finish_pair(pa[4 * g] + pb[g][0], pa[4 * g + 1] + pb[g][1], hidh[2 * g], hidl[2 * g]);
        finish_pair(pa[4 * g + 2] + pb[g][2], pa[4 * g + 3] + pb[g][3], hidh[2 * g + 1], hidl[2 * g + 1]);
      } else if constexpr (C >= 8 && C <= 11) {   // output layer -> k-steps 4, 5
        constexpr int g = C - 8;
        if constexpr (MX6 && L1MX) {   // k-steps 4, 5 = dwords 2 g, 2 g + 1 of slab 1
          if constexpr (g == 0) pam1 = 0.f;
          float v0 = pa[4 * g] + pb[g][0], v1 = pa[4 * g + 1] + pb[g][1], v2 = pa[4 * g + 2] + pb[g][2], v3 = pa[4 * g + 3] + pb[g][3];
          v0 = vmax(v0, v0 * 0.01f); v1 = vmax(v1, v1 * 0.01f); v2 = vmax(v2, v2 * 0.01f); v3 = vmax(v3, v3 * 0.01f);
          pp1[2 * g] = hi2_f16_amax(v0, v1, pam1); pp1[2 * g + 1] = hi2_f16_amax(v2, v3, pam1);
          lo2_f32(v0, v1, pp1[2 * g], plo1[4 * g], plo1[4 * g + 1]); lo2_f32(v2, v3, pp1[2 * g + 1], plo1[4 * g + 2], plo1[4 * g + 3]);
        } else {
        finish_pair(pa[4 * g] + pb[g][0], pa[4 * g + 1] + pb[g][1], Ph[4 + g / 2][(2 * g) & 3], Pl[4 + g / 2][(2 * g) & 3]);
        finish_pair(pa[4 * g + 2] + pb[g][2], pa[4 * g + 3] + pb[g][3], Ph[4 + g / 2][(2 * g + 1) & 3], Pl[4 + g / 2][(2 * g + 1) & 3]);
        }
      } else if constexpr (C >= 12 && C < 39) {
        // ---- positional encoding (utils.py:5-35).  Half hh owns octaves 5 hh .. 5 hh + 4 of every axis: value e = 10 a + 2 f' + comp
        // (comp 0 = sin, 1 = cos), then e = 30, 31 = raw (x, y) for half 0 and (z, 0) for half 1.  One accurate fp64 evaluation per
        // axis at the half's first octave (Cody-Waite to [-pi/4, pi/4] + Taylor, error < 1e-11) + the double-angle recurrence in
        // fp64 (abs error < 1e-12: correctly rounded in fp32).  Nine sub-steps per axis, the three axes interleaved.
        constexpr int x = (C - 12) / 3, a = (C - 12) % 3;
        if constexpr (x == 0) { sx[a] = (pe_t)poff[a] * (hh ? (pe_t)(32.0) : (pe_t)(1.0)); skd[a] = rint(sx[a] * (pe_t)(0.63661977236758134308)); }
        else if constexpr (x == 1) { sr[a] = fma(-skd[a], PE_PIO2_HI, sx[a]); sr[a] = fma(-skd[a], PE_PIO2_LO, sr[a]); }
        else if constexpr (x == 2) { sr2[a] = sr[a] * sr[a]; skq[a] = (int)skd[a]; su[a] = fma(sr2[a], (pe_t)(-1.0 / 39916800), (pe_t)(1.0 / 362880)); }
        else if constexpr (x == 3) { su[a] = fma(sr2[a], su[a], (pe_t)(-1.0 / 5040)); su[a] = fma(sr2[a], su[a], (pe_t)(1.0 / 120)); }
        else if constexpr (x == 4) { su[a] = fma(sr2[a], su[a], (pe_t)(-1.0 / 6)); sw[a] = fma(sr2[a], (pe_t)(1.0 / 479001600), (pe_t)(-1.0 / 3628800)); }
        else if constexpr (x == 5) { sw[a] = fma(sr2[a], sw[a], (pe_t)(1.0 / 40320)); sw[a] = fma(sr2[a], sw[a], (pe_t)(-1.0 / 720)); }
        else if constexpr (x == 6) { sw[a] = fma(sr2[a], sw[a], (pe_t)(1.0 / 24)); sw[a] = fma(sr2[a], sw[a], (pe_t)(-0.5)); }
        else if constexpr (x == 7) { su[a] = fma(sr[a] * sr2[a], su[a], sr[a]); sw[a] = fma(sr2[a], sw[a], (pe_t)(1.0)); }
        else {
          const bool swp = skq[a] & 1;
          const pe_t s1 = swp ? sw[a] : su[a], c1 = swp ? su[a] : sw[a];
          ss[a] = (skq[a] & 2) ? -s1 : s1;
          scs[a] = ((skq[a] + 1) & 2) ? -c1 : c1;
        }
      } else if constexpr (C >= 39 && C < 54) {
        constexpr int f = (C - 39) / 3, a = (C - 39) % 3, p = 5 * a + f;
        if constexpr (MX6 && L1MX) {   // pair p = dword p of slab 0
          if constexpr (C == 39) pam0 = 0.f;
          const float v0 = (float)ss[a], v1 = (float)scs[a];
          pp0[p] = hi2_f16_amax(v0, v1, pam0);
          lo2_f32(v0, v1, pp0[p], plo0[2 * p], plo0[2 * p + 1]);
        } else {
        unsigned h = 0, l = 0;
        split2f<X3, F16>((float)ss[a], (float)scs[a], h, l);
        Ph[p / 4][p & 3] = h;
        if (X3) Pl[p / 4][p & 3] = l;
        }
        if constexpr (f < 4) {
          const pe_t s2 = (pe_t)(2.0) * ss[a] * scs[a];
          scs[a] = fma((pe_t)(-2.0) * ss[a], ss[a], (pe_t)(1.0));
          ss[a] = s2;
        }
      } else if constexpr (C == 54) {
        if constexpr (MX6 && L1MX) {
          const float v0 = hh ? poff[2] : poff[0], v1 = hh ? 0.f : poff[1];
          pp0[15] = hi2_f16_amax(v0, v1, pam0);
          lo2_f32(v0, v1, pp0[15], plo0[30], plo0[31]);
        } else {
        unsigned h = 0, l = 0;
        split2f<X3, F16>(hh ? poff[2] : poff[0], hh ? 0.f : poff[1], h, l);
        Ph[3][3] = h;
        if (X3) Pl[3][3] = l;
        }
      } else {   // L1MX: a slab of the layer-1 operands is complete — block scale, the f16 fragments as one vector, the two fp6 images (as in the layers' epilogues)
        constexpr int q = C - 55;
        int eb = __builtin_amdgcn_frexp_expf(q == 0 ? pam0 : pam1) + 124;
        eb = eb < 12 ? 12 : (eb > 254 ? 254 : eb);
        const float sf = __builtin_bit_cast(float, eb << 23);
        if constexpr (q == 0) { pxs6h = (unsigned)eb; pxs6l = (unsigned)(eb - 11); }
        else { pxs6h |= (unsigned)eb << 8; pxs6l |= (unsigned)(eb - 11) << 8; }
        u32x16 H;
        f32x16 l0, l1;
        if constexpr (q == 0) {
          H = u32x16{pp0[0], pp0[1], pp0[2], pp0[3], pp0[4], pp0[5], pp0[6], pp0[7], pp0[8], pp0[9], pp0[10], pp0[11], pp0[12], pp0[13], pp0[14], pp0[15]};
          l0 = f32x16{plo0[0], plo0[1], plo0[2], plo0[3], plo0[4], plo0[5], plo0[6], plo0[7], plo0[8], plo0[9], plo0[10], plo0[11], plo0[12], plo0[13], plo0[14], plo0[15]};
          l1 = f32x16{plo0[16], plo0[17], plo0[18], plo0[19], plo0[20], plo0[21], plo0[22], plo0[23], plo0[24], plo0[25], plo0[26], plo0[27], plo0[28], plo0[29], plo0[30], plo0[31]};
        } else {
          H = u32x16{pp1[0], pp1[1], pp1[2], pp1[3], pp1[4], pp1[5], pp1[6], pp1[7], 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
          l0 = f32x16{plo1[0], plo1[1], plo1[2], plo1[3], plo1[4], plo1[5], plo1[6], plo1[7], plo1[8], plo1[9], plo1[10], plo1[11], plo1[12], plo1[13], plo1[14], plo1[15]};
          l1 = zero16;
        }
        P16[q] = H;
        const u32x6 rh = cvt_pk32_fp6_f16(H, sf), rl = cvt_2xpk16_fp6_f32(l0, l1, sf * 0.00048828125f);
        P6[q][0][0] = rh[0]; P6[q][0][1] = rh[1]; P6[q][0][2] = rh[2]; P6[q][0][3] = rh[3]; P6[q][0][4] = rh[4]; P6[q][0][5] = rh[5];
        P6[q][1][0] = rl[0]; P6[q][1][1] = rl[1]; P6[q][1][2] = rl[2]; P6[q][1][3] = rl[3]; P6[q][1][4] = rl[4]; P6[q][1][5] = rl[5];
      }
    }
  };
  // prologue compute steps of region G: the six k / v regions K0 .. V1 share them evenly
  auto pro_lo = [](int g) constexpr { return g < 3 * NRT ? 0 : g >= 3 * NRT + 6 ? NPC : (g - 3 * NRT) * NPC / 6; };
  // table-row slice of layer-1 chunk c = initial value of its accumulator: four 16-byte gathers (lane = row: each costs the CU's address
  // unit ~64 cycles, and the four waves issue in step), one per quarter of the region two ahead rather than back to back
  auto load_Tg = [&](auto Cc, auto Gc, unsigned tof) __attribute__((always_inline)) {
    constexpr int c = decltype(Cc)::value, g = decltype(Gc)::value;
    const f32x4 t4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rT, tof + (32 * GG::rt(c) + 4 * g) * 4, 0, 0));
    acc[c & 3][4 * g] = t4[0]; acc[c & 3][4 * g + 1] = t4[1]; acc[c & 3][4 * g + 2] = t4[2]; acc[c & 3][4 * g + 3] = t4[3];
  };
  auto load_T = [&](auto Cc, unsigned tof) __attribute__((always_inline)) {
    static_for<4>([&](auto Gc) __attribute__((always_inline)) { load_Tg(Cc, Gc, tof); });
  };
  auto load_bias = [&](auto Cc) __attribute__((always_inline)) {   // bias slice of layer-2/3 chunk c = initial value of its accumulator
    constexpr int c = GG::cm(decltype(Cc)::value);
    static_for<4>([&](auto Gc) __attribute__((always_inline)) {
      constexpr int g = decltype(Gc)::value;
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(sres + 64 + (GG::layer(c) - 1) * W + 32 * GG::rt(c) + 16 * hh + 4 * g);
      acc[c & 3][4 * g] = t4[0]; acc[c & 3][4 * g + 1] = t4[1]; acc[c & 3][4 * g + 2] = t4[2]; acc[c & 3][4 * g + 3] = t4[3];
    });
  };

  // A fragments of running k-step t of region G (t >= nks(G): chunk G+1) -> ring position
  auto read_frag = [&](auto Gc, auto Tc, auto Pc) __attribute__((always_inline)) {
    constexpr int G = decltype(Gc)::value, t = decltype(Tc)::value, part = decltype(Pc)::value;
    constexpr int c = t >= GG::nks(G) ? GG::cm(G + 1) : GG::cm(G), ks = t >= GG::nks(G) ? t - GG::nks(G) : t;
    constexpr int pos = GG::rpos(GG::cumks(GG::cm(G)) + t);
    constexpr int li = (c % NBUF) * SLOT + (part * GG::ksi(c) + ks) * 64;
    u32x4 v;
    if constexpr (li >= 4096 && li < 8192) v = __builtin_bit_cast(u32x4, lds_hi[li - 4096]);
    else v = __builtin_bit_cast(u32x4, lds_all[li + lane]);
    if (part == 0) frh[pos] = v; else frl[pos] = v;
  };

  // MX mode: 16 of the 32 fp8 bytes per lane of slab q of chunk c (part 1 of the slot: [slab][w_hi8 lo16 | w_hi8 hi16 | w_lo8 lo16 | w_lo8 hi16][lane]) -> w8[parity][i]
  auto read_w8 = [&](auto Cc, auto Qc, auto Ic) __attribute__((always_inline)) {
    constexpr int c = GG::cm(decltype(Cc)::value), q = decltype(Qc)::value, i = decltype(Ic)::value;
    constexpr int li = (c % NBUF) * SLOT + (GG::ksi(c) + 4 * q + i) * 64;
    if constexpr (li >= 4096 && li < 8192) w8[q & 1][i] = __builtin_bit_cast(u32x4, lds_hi[li - 4096]);
    else w8[q & 1][i] = __builtin_bit_cast(u32x4, lds_all[li + lane]);
  };

  // MX-FP6: the images of slab q of chunk c: part 1 of the slot = per slab [w_hi6 dwords 0-3 | w_hi6 dwords 4-5 | w_lo6 dwords 0-3 | w_lo6 dwords 4-5][lane] = 1 K + 512 + 1 K + 512 bytes
  auto read_w6 = [&](auto Cc, auto Qc, auto Ic) __attribute__((always_inline)) {
    constexpr int c = GG::cm(decltype(Cc)::value), q = decltype(Qc)::value, i = decltype(Ic)::value;
    constexpr int lb = ((c % NBUF) * SLOT + GG::ksi(c) * 64) * 16 + q * 3072 + (i >> 1) * 1536 + (i & 1) * 1024;   // byte offset in the LDS object
    if constexpr ((i & 1) == 0) {
      constexpr int li = lb / 16;
      if constexpr (li >= 4096 && li < 8192) w6a[q & 1][i >> 1] = __builtin_bit_cast(u32x4, lds_hi[li - 4096]);
      else w6a[q & 1][i >> 1] = __builtin_bit_cast(u32x4, lds_all[li + lane]);
    } else {
      constexpr int l8 = lb / 8;
      if constexpr (l8 >= 8192 && l8 < 16384) w6b[q & 1][i >> 1] = lds_hi8[l8 - 8192];
      else w6b[q & 1][i >> 1] = ((const lds_u2*)(lds_u4*)lds_all)[l8 + lane];
    }
  };
  // ... and the chunk's two scale dwords: resident (8 bytes per lane and wide chunk)
  auto read_wsc6 = [&](auto Cc) __attribute__((always_inline)) {
    constexpr int c = GG::cm(decltype(Cc)::value);
    wsc6 = lds_res8[(GG::RES_SC6 - NBUF * SLOT) * 2 + (L1MX ? c : c - NRT) * 64];
  };

  // ---------------------------------------------------------------- epilogue micro-steps of chunk C, run inside region C+1
  auto epi_step = [&](auto Cc, auto Ec, auto PrevC) __attribute__((always_inline)) {
    constexpr int C = GG::cm(decltype(Cc)::value), E = decltype(Ec)::value;
    constexpr bool PREV = decltype(PrevC)::value;   // chunk of the previous tile (its V head 3 finishes inside the next tile's first region)
    constexpr int L = GG::layer(C), RT = GG::rt(C), AB = C & 3;
    if constexpr (L < 3) {
      constexpr int SPP = X3 ? 2 : 1;
      constexpr int p = E / SPP, sub = E % SPP;
      constexpr int out = L & 1;   // L1 -> X[0], L2 -> X[1], L3 -> X[0]
      constexpr int fo = 2 * RT + (p >> 2), d = p & 3;
      if constexpr (MX6) {
        constexpr int par = RT & 1, q = RT >> 1;
        if constexpr (E < 16 && sub == 0) {
          if constexpr (par == 0 && p == 0) amax = 0.f;   // a slab's block: this tile's and the next one's 16 values of the lane
          ev0 = acc[AB][2 * p]; ev1 = acc[AB][2 * p + 1];
          ehi = lrelu_hi2_f16_amax(ev0, ev1, amax);
          hp6[par][p] = ehi;
        } else if constexpr (E < 16) {
          lo2_f32(ev0, ev1, ehi, lo6t[par][2 * p], lo6t[par][2 * p + 1]);
        } else if constexpr (E == 16) {   // the slab is complete: block scale 2^(ex - 3) for amax = m 2^ex, m in [0.5, 1); the hi image from the f16 fragments
          int eb = __builtin_amdgcn_frexp_expf(amax) + 124;
          eb = eb < 12 ? 12 : (eb > 254 ? 254 : eb);   // (12: the residual image's byte eb - 11 stays positive; an all-zero block takes any scale)
          scf6 = __builtin_bit_cast(float, eb << 23);
          if constexpr (q == 0) { xs6h[out] = (unsigned)eb; xs6l[out] = (unsigned)(eb - 11); }
          else { xs6h[out] |= (unsigned)eb << (8 * q); xs6l[out] |= (unsigned)(eb - 11) << (8 * q); }
          const u32x16 H = {hp6[0][0], hp6[0][1], hp6[0][2], hp6[0][3], hp6[0][4], hp6[0][5], hp6[0][6], hp6[0][7],
                            hp6[1][0], hp6[1][1], hp6[1][2], hp6[1][3], hp6[1][4], hp6[1][5], hp6[1][6], hp6[1][7]};
          Xh16[out][q] = H;
          const u32x6 r = cvt_pk32_fp6_f16(H, scf6);
          X6[out][q][0][0] = r[0]; X6[out][q][0][1] = r[1]; X6[out][q][0][2] = r[2]; X6[out][q][0][3] = r[3]; X6[out][q][0][4] = r[4]; X6[out][q][0][5] = r[5];
        } else {   // the residual image: two row tiles interleaved, scale x 2^-11
          const f32x16 l0 = {lo6t[0][0], lo6t[0][1], lo6t[0][2], lo6t[0][3], lo6t[0][4], lo6t[0][5], lo6t[0][6], lo6t[0][7],
                             lo6t[0][8], lo6t[0][9], lo6t[0][10], lo6t[0][11], lo6t[0][12], lo6t[0][13], lo6t[0][14], lo6t[0][15]};
          const f32x16 l1 = {lo6t[1][0], lo6t[1][1], lo6t[1][2], lo6t[1][3], lo6t[1][4], lo6t[1][5], lo6t[1][6], lo6t[1][7],
                             lo6t[1][8], lo6t[1][9], lo6t[1][10], lo6t[1][11], lo6t[1][12], lo6t[1][13], lo6t[1][14], lo6t[1][15]};
          const u32x6 r = cvt_2xpk16_fp6_f32(l0, l1, scf6 * 0.00048828125f);
          X6[out][q][1][0] = r[0]; X6[out][q][1][1] = r[1]; X6[out][q][1][2] = r[2]; X6[out][q][1][3] = r[3]; X6[out][q][1][4] = r[4]; X6[out][q][1][5] = r[5];
        }
      } else if constexpr (MX) {
        if constexpr (sub == 0) {
          if constexpr (RT == 0 && p == 0) {   // the scales of this layer's output rows (see xs_hi): fixed before its first row tile is converted
            if constexpr (L == 0) set_scale(std::integral_constant<int, 0>{}, cur_b1);
            else if constexpr (L == 1) set_scale(std::integral_constant<int, 1>{}, fmaf(bnd_B2, row_amax(), bnd_bm2));
            else set_scale(std::integral_constant<int, 0>{}, fmaf(bnd_B3, row_amax(), bnd_bm3));
            amax = 0.f;
          }
          ev0 = acc[AB][2 * p]; ev1 = acc[AB][2 * p + 1];
          if constexpr (L < 2) ehi = lrelu_hi2_f16_amax(ev0, ev1, amax); else ehi = lrelu_hi2_f16(ev0, ev1);
          Xh[out][fo][d] = ehi;
        } else mx_bytes2<d & 1>(ev0, ev1, ehi, X8[out][fo >> 2][0][2 * (fo & 3) + (d >> 1)], X8[out][fo >> 2][1][2 * (fo & 3) + (d >> 1)], xs_hi[out], xs_lo[out]);
      } else if constexpr (sub == 0) {
        ev0 = acc[AB][2 * p]; ev1 = acc[AB][2 * p + 1];
        ehi = F16 ? lrelu_hi2_f16(ev0, ev1) : lrelu_hi2(ev0, ev1);
        Xh[out][fo][d] = ehi;
        if constexpr (KEEP) {   // sign bits of the layer's outputs: bit 16 (RT & 1) + r of dword RT >> 1 <-> accumulator register r of row tile RT
          if constexpr (RT == 0 && p == 0) { kmask[0] = kmask[1] = kmask[2] = kmask[3] = 0u; }
          kmask[RT >> 1] |= ((ev0 > 0.f ? 1u : 0u) | (ev1 > 0.f ? 2u : 0u)) << (16 * (RT & 1) + 2 * p);
        }
      } else {
        Xl[out][fo][d] = F16 ? lo2_f16(ev0, ev1, ehi) : lo2(ev0, ev1, ehi);
        if constexpr (KEEP && RT == NRT - 1 && p == 7)
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{kmask[0], NRT > 2 ? kmask[1] : 0u, NRT > 4 ? kmask[2] : 0u, NRT > 4 ? kmask[3] : 0u},
                                                 L == 0 ? rM0 : L == 1 ? rM1 : rM2, (unsigned)((PREV ? tile_prev : tile) * 4 + wave) * 1024u + lane * 16u, 0, 0);
      }
    } else if constexpr (RT < 4) {   // k projection of head RT: scores, softmax over the 8 neighbours (lanes) of a sample
      if constexpr (E < 4) {
        if constexpr (E == 0) ap = Qr[0][0] * acc[AB][0]; else ap = fmaf(Qr[E][0], acc[AB][4 * E], ap);
        ap = fmaf(Qr[E][1], acc[AB][4 * E + 1], ap);
        ap = fmaf(Qr[E][2], acc[AB][4 * E + 2], ap);
        ap = fmaf(Qr[E][3], acc[AB][4 * E + 3], ap);
      } else if constexpr (E == 4) {
        ap += __shfl_xor(ap, 32, 64);
        ap *= 1.0f / 5.656854249492381f;   // temperature sqrt(d_k) (ibrnet.py:84)
        lmax = (ap == ap) ? fmaxf(lmax, fabsf(ap)) : __builtin_inff();   // a NaN logit counts as beyond every mode's validated range (fmaxf alone would drop it)
      } else if constexpr (E == 5) amx = nl_max8(ap);
      else if constexpr (E == 6) aee = expf(ap - amx);
      else if constexpr (E == 7) ase = nl_sum8(aee);
      else if constexpr (E == 8) att[RT] = aee / ase;
      else if constexpr (E == 9) satt[RT * 32 + j] = att[RT];   // both halves hold the same value; every lane writes (no divergent store, no branch)
      else {   // KEEP: the k rows of head RT: row = lane's neighbour row, columns 32 RT + 8 g + 4 hh .. + 3
        constexpr int g = E - 10;
        // (a float copy first: __builtin_bit_cast applied to an ext-vector ELEMENT reads element 0 whatever the index)
        const float k0 = acc[AB][4 * g], k1 = acc[AB][4 * g + 1], k2 = acc[AB][4 * g + 2], k3 = acc[AB][4 * g + 3];
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(k0), __float_as_uint(k1), __float_as_uint(k2), __float_as_uint(k3)},
                                               rKV, kvrow + (unsigned)(32 * RT + 8 * g + 4 * hh) * 4u, 0, 0);
      }
    } else {
      // v projection of head h, computed TRANSPOSED (activations = A operand): lane = output dim n, registers = neighbour rows
      // m(r, hh) = (r & 3) + 8 (r >> 2) + 4 hh, i.e. sample r >> 2, neighbours (r & 3) + 4 hh.  The attention-weighted sum over the
      // 8 neighbours is 4 FMAs per sample in the lane + one cross-half add, instead of a 3-step DPP reduction per value.
      constexpr int h = RT - 4;
      if constexpr (E < 4) aw[E] = *reinterpret_cast<const f32x4*>(satt + h * 32 + 8 * E + 4 * hh);   // weights of sample E's neighbours 4 hh .. 4 hh + 3
      else if constexpr (E < 8) {
        constexpr int sI = E - 4;
        float o = aw[sI][0] * acc[AB][4 * sI];
        o = fmaf(aw[sI][1], acc[AB][4 * sI + 1], o);
        o = fmaf(aw[sI][2], acc[AB][4 * sI + 2], o);
        o = fmaf(aw[sI][3], acc[AB][4 * sI + 3], o);
        ov[sI] = o;
      } else if constexpr (E == 8) {
        ov[0] += __shfl_xor(ov[0], 32, 64); ov[1] += __shfl_xor(ov[1], 32, 64);
        ov[2] += __shfl_xor(ov[2], 32, 64); ov[3] += __shfl_xor(ov[3], 32, 64);
      } else if constexpr (E == 9) {
        // lanes of half 0 store dim n of the wave's 4 samples; half 1 and samples past N are out of range (dropped by the bounds check)
        const unsigned oo = (PREV ? ooff_prev : ooff_cur) + 4 * 32 * h;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov[0]), rO, oo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov[1]), rO, oo + 512, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov[2]), rO, oo + 1024, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov[3]), rO, oo + 1536, 0, 0);
      } else {   // KEEP: the v values of head h: lane = dim, registers = neighbour rows m(r, hh) -> column 128 + 32 h + lane's dim of those rows
        constexpr int g = E - 10;
        const unsigned base = (PREV ? kvtile_prev : kvtile) + (unsigned)(128 + 32 * h + j) * 4u;
        static_for<4>([&](auto Rc) __attribute__((always_inline)) {
          constexpr int r = 4 * g + decltype(Rc)::value;
          const float vr = acc[AB][r];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vr), rKV, base + (unsigned)((r & 3) + 8 * (r >> 2) + 4 * hh) * 1024u, 0, 0);
        });
      }
    }
  };

  // everything that is issued in the shadow of MFMA slot K of region G
  auto fill = [&](auto Gc, auto Kc) __attribute__((always_inline)) {
    constexpr int G = decltype(Gc)::value, K = decltype(Kc)::value;
    // slots of a region: one per MFMA (3 per k-step); an MX region has 8 units per slab (f16 MFMA = 1, fp8 MFMA = 2: its issue shadow is twice as long),
    // its barrier sits in front of unit 8 (NSL - 1) + 3 (every LDS-DMA piece is issued before it), and a layer epilogue must be through before the last slab
    constexpr int NKS = GG::nks(G);
    constexpr bool L1R = MX6 && L1MX && GG::layer(G) == 0;   // a layer-1 region in the MX-FP6 arithmetic: 10 units, the barrier in front of unit 7
    constexpr bool MXR = MX && GG::layer(G) > 0;
    // (MX-FP6: its cross-term instructions take 8 passes like the f16 ones — 6 units per slab, the barrier in front of unit 6 (NSL - 1) + 3)
    constexpr int NS = L1R ? 10 : MXR ? (MX6 ? 6 * (NKS / 4) : 2 * NKS) : MPK * NKS, NSD = L1R ? 7 : MXR ? (MX6 ? 6 * (NKS / 4) - 3 : 2 * NKS - 5) : MPK * (NKS - 1),
                  NSEL = L1R ? 8 : MXR ? (MX6 ? 6 * (NKS / 4) - 6 : 2 * NKS - 8) : NSD;
    // LDS-DMA pieces of chunk G+3 (its slot held chunk G-1, which every wave left behind at the previous barrier)
    if constexpr (K < NSD && !(KO & 4)) {
      constexpr int ND = GG::ppw(G + 3), d0 = K * ND / NSD, d1 = (K + 1) * ND / NSD;
      static_for<d1 - d0>([&](auto Ic) __attribute__((always_inline)) { dma_piece(std::integral_constant<int, G + 3>{}, std::integral_constant<int, d0 + decltype(Ic)::value>{}); });
    }
    // accumulator initial values: table rows of the layer-1 chunk two regions ahead (a gather), bias of the next layer-2/3 chunk (LDS)
    if constexpr (GG::layer(G + 2) == 0 && (K == 0 || K == NS / 4 || K == NS / 2 || K == 3 * NS / 4)) {
      constexpr int g = K == 0 ? 0 : K == NS / 4 ? 1 : K == NS / 2 ? 2 : 3;
      if constexpr (G + 2 < NC) load_Tg(std::integral_constant<int, G + 2>{}, std::integral_constant<int, g>{}, toff);
      else load_Tg(std::integral_constant<int, G + 2 - NC>{}, std::integral_constant<int, g>{}, pn_toff);   // the next tile's first two row tiles
    }
    // the next tile's prologue: tile-number loads at the start of layer 3, neighbour gathers half a layer later, arithmetic under k / v
    if constexpr (!(KO & 1)) {
      if constexpr (K == 0 && G == 2 * NRT) pro_step(std::integral_constant<int, 0>{});
      if constexpr (K == 0 && G == 2 * NRT + NRT / 2) pro_step(std::integral_constant<int, 1>{});
      if constexpr (G >= 3 * NRT && G < 3 * NRT + 6) {
        constexpr int lo = pro_lo(G), cnt = pro_lo(G + 1) - lo, c0 = K * cnt / NS, c1 = (K + 1) * cnt / NS;
        static_for<c1 - c0>([&](auto Ic) __attribute__((always_inline)) { pro_step(std::integral_constant<int, NPL + lo + c0 + decltype(Ic)::value>{}); });
      }
    }
    if constexpr (K == NS / 2 && (GG::layer(G + 1) == 1 || GG::layer(G + 1) == 2)) load_bias(std::integral_constant<int, G + 1>{});
    // query slice of the head whose k projection this region computes (scored in the next region)
    if constexpr (GG::layer(G) == 3 && GG::rt(G) < 4 && K >= NS / 2 && (K - NS / 2) % (NS / 8) == 0 && (K - NS / 2) / (NS / 8) < 4) {
      constexpr int g = (K - NS / 2) / (NS / 8);
      Qr[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rQ, qoff + (32 * GG::rt(G) + 8 * g) * 4, 0, 0));
    }
    // epilogue of the previous chunk
    {
      // A layer's last row tile is finished inside the first region of the NEXT layer, which consumes the fragments it produces
      // in its last two k-step groups (operands are assembled at the start of a group): layer epilogues end one group early
      constexpr int NE = GG::epi_steps(G - 1), NSE = GG::layer(G - 1) < 3 ? NSEL : NS;
      if constexpr (K < NSE && !((KO & 2) && GG::layer(G - 1) < 3) && !((KO & 8) && GG::layer(G - 1) == 3)) {
        constexpr int e0 = K * NE / NSE, e1 = (K + 1) * NE / NSE;
        static_for<e1 - e0>([&](auto Ec) __attribute__((always_inline)) {
          epi_step(std::integral_constant<int, G - 1>{}, std::integral_constant<int, e0 + decltype(Ec)::value>{}, std::integral_constant<bool, G == 0>{});
        });
      }
    }
  };

  // ---------------------------------------------------------------- one region = one output row tile accumulated over all K
  int trace_it = 0;
  (void)trace_it;
  auto region = [&](auto Gc) __attribute__((always_inline)) {
    constexpr int G = decltype(Gc)::value;
#ifdef PF2_TRACE
    if (blockIdx.x == 0 && wave == 0 && trace_it < 4) {
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) pf2_trace[trace_it * 64 + G] = t;
    }
#endif
    constexpr int L = GG::layer(G), NKS = GG::nks(G), AB = G & 3, CK = GG::cumks(G);
    constexpr bool ZI = L == 3;   // k / v projections have no bias: the first MFMA takes C = 0
    constexpr bool NEXT_MX = MX && (GG::layer(G + 1) > 0 || (MX6 && L1MX));   // the next chunk's part 1 holds fp8 / fp6 images (its part 0: f16 fragments)
    if constexpr (MX6 && L1MX && L == 0) {
      // layer 1 on the same arithmetic: slab 0 = k-steps 0-3, slab 1 = k-steps 4, 5 (+ two empty ones: zero positions in both operands' images); 6 + 4 matrix instructions of
      // 8 passes; units: 0-3 | 4, 5 | 6, 7 | 8, 9, the barrier in front of unit 7
      read_wsc6(Gc);
      static_for<2>([&](auto Qc) __attribute__((always_inline)) {
        constexpr int q = decltype(Qc)::value, NKQ = q == 0 ? 4 : 2;
        static_for<NKQ>([&](auto Sc) __attribute__((always_inline)) {
          constexpr int sI = decltype(Sc)::value, ks = 4 * q + sI, pos = GG::rpos(CK + ks);
          if constexpr (ks == NKS - 1) {
            if constexpr (!(KO & 4)) wait_vmcnt<GG::ppw(G + 2) + GG::ppw(G + 3)>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(KO & 256) && !((KO & 512) && (G & 1))) __builtin_amdgcn_s_barrier();
          }
          const u32x4 bh = __builtin_shufflevector(P16[q], P16[q], 4 * sI, 4 * sI + 1, 4 * sI + 2, 4 * sI + 3);
          if constexpr (ks + 2 < NKS) read_frag(Gc, std::integral_constant<int, ks + 2>{}, std::integral_constant<int, 0>{});
          else if constexpr (ks == NKS - 1) {
            read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 0>{});
            read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 0>{});
          }
          acc[AB] = mfma_h(frh[pos], bh, acc[AB]);   // (the accumulator starts from the table row: load_T)
          fill(Gc, std::integral_constant<int, 6 * q + sI>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        static_for<2>([&](auto Ic) __attribute__((always_inline)) {
          constexpr int im = decltype(Ic)::value;   // 0: w_hi6 x a_lo6, 1: w_lo6 x a_hi6
          const i32x8 wa = {(int)w6a[q & 1][im][0], (int)w6a[q & 1][im][1], (int)w6a[q & 1][im][2], (int)w6a[q & 1][im][3], (int)w6b[q & 1][im][0], (int)w6b[q & 1][im][1], 0, 0};
          const unsigned(&xd)[6] = P6[q][1 - im];
          const i32x8 xb = {(int)xd[0], (int)xd[1], (int)xd[2], (int)xd[3], (int)xd[4], (int)xd[5], 0, 0};
          const int sw = (int)wsc6[im], sx = (int)(im == 0 ? pxs6l : pxs6h);
          acc[AB] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, xb, acc[AB], 2, 2, q, sw, q, sx);
          if constexpr (q == 0) {
            read_w6(Gc, std::integral_constant<int, 1>{}, std::integral_constant<int, 2 * im>{});
            read_w6(Gc, std::integral_constant<int, 1>{}, std::integral_constant<int, 2 * im + 1>{});
          } else {
            read_w6(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * im>{});
            read_w6(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * im + 1>{});
          }
          fill(Gc, std::integral_constant<int, q == 0 ? 4 + im : 8 + im>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    } else if constexpr (MX6 && L > 0) {
      constexpr int NSL = NKS / 4, IN = (L + 1) & 1;
      constexpr bool TR = L == 3 && GG::rt(G) >= 4;   // v heads: D = X . Wv^T (rows = neighbour rows) instead of D^T
      read_wsc6(Gc);
      static_for<NSL>([&](auto Qc) __attribute__((always_inline)) {
        constexpr int q = decltype(Qc)::value;
        static_for<4>([&](auto Sc) __attribute__((always_inline)) {
          constexpr int sI = decltype(Sc)::value, ks = 4 * q + sI, pos = GG::rpos(CK + ks);
          if constexpr (ks == NKS - 1) {
            if constexpr (!(KO & 4)) wait_vmcnt<GG::ppw(G + 2) + GG::ppw(G + 3)>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(KO & 256) && !((KO & 512) && (G & 1))) __builtin_amdgcn_s_barrier();
          }
          const u32x4 bh = __builtin_shufflevector(Xh16[IN][q], Xh16[IN][q], 4 * sI, 4 * sI + 1, 4 * sI + 2, 4 * sI + 3);
          if constexpr (ks + 2 < NKS) read_frag(Gc, std::integral_constant<int, ks + 2>{}, std::integral_constant<int, 0>{});
          else if constexpr (ks == NKS - 1) {
            read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 0>{});
            read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 0>{});
            if constexpr (!NEXT_MX) {   // the next tile's first layer-1 chunk: split-bf16 parts
              read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 1>{});
              read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 1>{});
            }
          }
          const f32x16 c0 = (ZI && ks == 0) ? zero16 : acc[AB];
          acc[AB] = TR ? mfma_h(bh, frh[pos], c0) : mfma_h(frh[pos], bh, c0);
          fill(Gc, std::integral_constant<int, 6 * q + sI>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        // the two cross terms, 8 passes each: w_hi6 x a_lo6, w_lo6 x a_hi6; scales: byte q of the weights' and of the activations' scale dwords
        static_for<2>([&](auto Ic) __attribute__((always_inline)) {
          constexpr int im = decltype(Ic)::value;   // 0: w_hi6 x a_lo6, 1: w_lo6 x a_hi6
          const i32x8 wa = {(int)w6a[q & 1][im][0], (int)w6a[q & 1][im][1], (int)w6a[q & 1][im][2], (int)w6a[q & 1][im][3], (int)w6b[q & 1][im][0], (int)w6b[q & 1][im][1], 0, 0};
          const unsigned(&xd)[6] = X6[IN][q][1 - im];
          const i32x8 xb = {(int)xd[0], (int)xd[1], (int)xd[2], (int)xd[3], (int)xd[4], (int)xd[5], 0, 0};
          const int sw = (int)wsc6[im], sx = (int)(im == 0 ? xs6l[IN] : xs6h[IN]);
          if constexpr (!((KO & 16) && im == 1) && !((KO & 32) && im == 0) && !((KO & 64) && q >= 2) && !((KO & 128) && q < 2)) {   // (32 / 64 / 128: debugging knock-outs of single cross terms)
            if constexpr (TR) acc[AB] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xb, wa, acc[AB], 2, 2, q, sx, q, sw);
            else acc[AB] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, xb, acc[AB], 2, 2, q, sw, q, sx);
          }
          if constexpr (q + 1 < NSL) {
            read_w6(Gc, std::integral_constant<int, q + 1>{}, std::integral_constant<int, 2 * im>{});
            read_w6(Gc, std::integral_constant<int, q + 1>{}, std::integral_constant<int, 2 * im + 1>{});
          } else if constexpr (NEXT_MX) {
            read_w6(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * im>{});
            read_w6(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * im + 1>{});
          }
          fill(Gc, std::integral_constant<int, 6 * q + 4 + im>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    } else if constexpr (MX && L > 0) {
      constexpr int NSL = NKS / 4, IN = (L + 1) & 1;
      constexpr bool TR = L == 3 && GG::rt(G) >= 4;   // v heads: D = X . Wv^T (rows = neighbour rows) instead of D^T
      const int swh = ssc[2 * GG::cm(G)], swl = ssc[2 * GG::cm(G) + 1];
      static_for<NSL>([&](auto Qc) __attribute__((always_inline)) {
        constexpr int q = decltype(Qc)::value;
        static_for<4>([&](auto Sc) __attribute__((always_inline)) {
          constexpr int ks = 4 * q + decltype(Sc)::value, pos = GG::rpos(CK + ks);
          if constexpr (ks == NKS - 1) {
            if constexpr (!(KO & 4)) wait_vmcnt<GG::ppw(G + 2) + GG::ppw(G + 3)>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(KO & 256) && !((KO & 512) && (G & 1))) __builtin_amdgcn_s_barrier();
          }
          const u32x4 bh = frag4(Xh[IN][ks]);
          if constexpr (ks + 2 < NKS) read_frag(Gc, std::integral_constant<int, ks + 2>{}, std::integral_constant<int, 0>{});
          else if constexpr (ks == NKS - 1) {
            read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 0>{});
            read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 0>{});
            if constexpr (!NEXT_MX) {   // the next tile's first layer-1 chunk: split-bf16 parts
              read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 1>{});
              read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 1>{});
            }
          }
          const f32x16 c0 = (ZI && ks == 0) ? zero16 : acc[AB];
          acc[AB] = TR ? mfma_h(bh, frh[pos], c0) : mfma_h(frh[pos], bh, c0);
          fill(Gc, std::integral_constant<int, 8 * q + decltype(Sc)::value>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        // the two cross terms: w_hi8 x a_lo8 (block scale 2^-11 on the activations' side), w_lo8 x a_hi8 (the weights' lo image carries its own scale)
        {
          const i32x8 wa = cat8(w8[q & 1][0], w8[q & 1][1]), xb = frag8(X8[IN][q][1]);
          acc[AB] = TR ? mfma_8(xb, wa, acc[AB], xs_e8l[IN], swh) : mfma_8(wa, xb, acc[AB], swh, xs_e8l[IN]);
          if constexpr (q + 1 < NSL) {
            read_w8(Gc, std::integral_constant<int, q + 1>{}, std::integral_constant<int, 0>{});
            read_w8(Gc, std::integral_constant<int, q + 1>{}, std::integral_constant<int, 1>{});
          } else if constexpr (NEXT_MX) {
            read_w8(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            read_w8(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
          }
          fill(Gc, std::integral_constant<int, 8 * q + 4>{});
          fill(Gc, std::integral_constant<int, 8 * q + 5>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          const i32x8 wa = cat8(w8[q & 1][2], w8[q & 1][3]), xb = frag8(X8[IN][q][0]);
          if constexpr (!(KO & 16)) acc[AB] = TR ? mfma_8(xb, wa, acc[AB], xs_e8h[IN], swl) : mfma_8(wa, xb, acc[AB], swl, xs_e8h[IN]);
          else asm volatile("" :: "v"(wa), "v"(xb));
          if constexpr (q + 1 < NSL) {
            read_w8(Gc, std::integral_constant<int, q + 1>{}, std::integral_constant<int, 2>{});
            read_w8(Gc, std::integral_constant<int, q + 1>{}, std::integral_constant<int, 3>{});
          } else if constexpr (NEXT_MX) {
            read_w8(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            read_w8(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
          }
          fill(Gc, std::integral_constant<int, 8 * q + 6>{});
          fill(Gc, std::integral_constant<int, 8 * q + 7>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    } else
    static_for<NKS>([&](auto Kc) __attribute__((always_inline)) {
      constexpr int ks = decltype(Kc)::value, pos = GG::rpos(CK + ks);
      if constexpr (ks == NKS - 1) {
        // chunk G+1 must have landed (only the pieces of G+2, G+3 may still fly) and every wave must be through with chunk G's
        // slot reads; its last fragments are in registers already
        if constexpr (!(KO & 4)) wait_vmcnt<GG::ppw(G + 2) + GG::ppw(G + 3)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(KO & 256) && !((KO & 512) && (G & 1))) __builtin_amdgcn_s_barrier();
      }
      auto bsel = [&](auto Hi) __attribute__((always_inline)) {
        if constexpr (L == 0) { if constexpr (decltype(Hi)::value) return frag4(Ph[ks]); else return frag4(Pl[ks]); }
        else { if constexpr (decltype(Hi)::value) return frag4(Xh[(L + 1) & 1][ks]); else return frag4(Xl[(L + 1) & 1][ks]); }
      };
      const u32x4 bh = bsel(std::true_type{});
      u32x4 bl = bh;
      if constexpr (X3) bl = bsel(std::false_type{});
      static_for<MPK>([&](auto Mc) __attribute__((always_inline)) {
        constexpr int m = decltype(Mc)::value, K = MPK * ks + m;
        // A fragments two k-steps ahead; the first two of the next chunk wait for the barrier of the last group
        if constexpr (ks + 2 < NKS) {
          if constexpr (m == 0) read_frag(Gc, std::integral_constant<int, ks + 2>{}, std::integral_constant<int, 0>{});
          if constexpr (X3 && m == 1) read_frag(Gc, std::integral_constant<int, ks + 2>{}, std::integral_constant<int, 1>{});
        } else if constexpr (ks == NKS - 1 && NEXT_MX) {   // a layer-1 region hands over to an MX region: f16 fragments of its k-steps 0, 1 + the fp8 images of its slab 0
          if constexpr (m == 0) {
            read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 0>{});
            read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 0>{});
          } else if constexpr (MX6) {
            read_w6(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * (m - 1)>{});
            read_w6(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * (m - 1) + 1>{});
          } else {
            read_w8(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * (m - 1)>{});
            read_w8(std::integral_constant<int, G + 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * (m - 1) + 1>{});
          }
        } else if constexpr (ks == NKS - 1) {
          if constexpr (m == 0) {
            read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 0>{});
            if constexpr (X3) read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 1>{});
            if constexpr (!X3) read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 0>{});
          }
          if constexpr (X3 && m == 1) {
            read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 0>{});
            read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 1>{});
          }
        }
        constexpr bool TR = L == 3 && GG::rt(G) >= 4;   // v heads: D = X . Wv^T (rows = neighbour rows) instead of D^T
        auto mm = [&](const u32x4& wf, const u32x4& xf, const f32x16& c) __attribute__((always_inline)) { return TR ? mfma(xf, wf, c) : mfma(wf, xf, c); };
        if constexpr (X3) {
          if constexpr (m == 0) acc[AB] = mm(frl[pos], bh, (ZI && ks == 0) ? zero16 : acc[AB]);
          else if constexpr (m == 1) acc[AB] = mm(frh[pos], bl, acc[AB]);
          else acc[AB] = mm(frh[pos], bh, acc[AB]);
        } else acc[AB] = mm(frh[pos], bh, (ZI && ks == 0) ? zero16 : acc[AB]);
        fill(Gc, std::integral_constant<int, K>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---------------------------------------------------------------- pipeline start
  soff = (unsigned)(GG::gkb(NC - 1) * 1024 + ((GG::ppw(NC - 1) - 1) / 4) * 4096) + (unsigned)wave * (unsigned)(GG::ppw(NC - 1) * 1024);   // the "previous group" of the very first one
  static_for<3>([&](auto Cc) __attribute__((always_inline)) {
    static_for<GG::ppw(decltype(Cc)::value)>([&](auto Ic) __attribute__((always_inline)) { dma_piece(Cc, Ic); });
  });
  static_for<NPRO>(pro_step);   // the first tile's prologue, back to back
  toff = pn_toff; qoff = pn_qoff; ooff_cur = pn_ooff; cur_b1 = pn_b1;
  kvtile = ((unsigned)tile * 128u + (unsigned)wave * 32u) * 1024u; kvrow = kvtile + (unsigned)j * 1024u;
  load_T(std::integral_constant<int, 0>{}, toff); load_T(std::integral_constant<int, 1>{}, toff);
  wait_vmcnt<GG::ppw(1) + GG::ppw(2)>();   // conservative: the prologue's own loads are younger than every piece
  __builtin_amdgcn_s_barrier();
  read_frag(std::integral_constant<int, NC - 1>{}, std::integral_constant<int, GG::nks(NC - 1)>{}, std::integral_constant<int, 0>{});
  read_frag(std::integral_constant<int, NC - 1>{}, std::integral_constant<int, GG::nks(NC - 1) + 1>{}, std::integral_constant<int, 0>{});
  if constexpr (MX6 && L1MX) {   // chunk 0's slab-0 images instead of its lo fragments
    static_for<4>([&](auto Ic) __attribute__((always_inline)) { read_w6(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, Ic); });
  } else if constexpr (X3) {
    read_frag(std::integral_constant<int, NC - 1>{}, std::integral_constant<int, GG::nks(NC - 1)>{}, std::integral_constant<int, 1>{});
    read_frag(std::integral_constant<int, NC - 1>{}, std::integral_constant<int, GG::nks(NC - 1) + 1>{}, std::integral_constant<int, 1>{});
  }

  for (;;) {
    pn_tile = tile + (int)nwg;   // rows past N are clamped inside the steps: the last tile prepares a tile that is never computed
    static_for<NC>(region);
#ifdef PF2_TRACE
    if (blockIdx.x == 0 && wave == 0 && trace_it < 4) {
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) pf2_trace[trace_it * 64 + NC] = t;
    }
    ++trace_it;
#endif
    ooff_prev = ooff_cur; ooff_cur = pn_ooff; toff = pn_toff; qoff = pn_qoff; cur_b1 = pn_b1;
    tile_prev = tile; kvtile_prev = kvtile;
    tile = pn_tile;
    kvtile = ((unsigned)tile * 128u + (unsigned)wave * 32u) * 1024u; kvrow = kvtile + (unsigned)j * 1024u;
    if (tile >= sc.ntiles) break;
  }
  // the last v head of the last tile
  static_for<GG::epi_steps(NC - 1)>([&](auto Ec) __attribute__((always_inline)) { epi_step(std::integral_constant<int, NC - 1>{}, Ec, std::integral_constant<bool, true>{}); });
  wait_vmcnt<0>();   // LDS-DMA prefetched for a tile that does not exist must land before the LDS is handed to another workgroup
  if (sc.clk && blockIdx.x == 0 && tid == 0) { sc.clk[0] = __builtin_readcyclecounter() - clk_c0; sc.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0; }
  if (sc.logit_amax) {
    const float m = wave_max(lmax);
    if (lane == 0) atomicMax(sc.logit_amax, __float_as_uint(m));   // (non-negative floats order like their bit patterns; a NaN logit was recorded as +inf above)
  }
}

// ---------------------------------------------------------------------------------------------------- packing
__device__ __forceinline__ unsigned short pf2_f2bf(float x) {
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ int pf2_m(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }   // accumulator register -> row of the 32x32 tile

// Stream: chunk (layer, rt) = [part hi/lo][k-step][lane][8 bf16] in A-fragment order (lane: out row 32 rt + (lane & 31), k-slots
// 8 (lane >> 5) + t); then the resident block: ray_diff_fc fragments [layer][part][lane][8] (4 KB) and the bias tables (floats).
// f32 (already divided by the block scale) -> OCP e4m3 byte, round to nearest even, saturating at +-448 (what v_cvt_scalef32_pk_fp8_f32 does under FP16_OVFL)
__device__ __forceinline__ unsigned char pf2_e4m3(float x) {
  const unsigned char sg = x < 0.f ? 0x80 : 0;
  const float a = fabsf(x);
  if (!(a == a)) return sg | 0x7f;
  if (a >= 448.f) return sg | 0x7e;
  int e;
  (void)frexpf(a, &e);            // a = m 2^e, m in [0.5, 1)
  int E = e - 1;                  // a = 1.xxx 2^E
  if (a == 0.f || E < -6) {       // subnormal grid: multiples of 2^-9
    const int qn = (int)rintf(a * 512.f);
    return sg | (unsigned char)qn;   // qn == 8 is the smallest normal (0x08): the encodings are contiguous
  }
  int qn = (int)rintf(ldexpf(a, 3 - E));   // 8 .. 16
  if (qn == 16) { qn = 8; ++E; }
  const int b = ((E + 7) << 3) | (qn - 8);
  return sg | (unsigned char)(b > 0x7e ? 0x7e : b);
}
__device__ __forceinline__ unsigned short pf2_f2h(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float pf2_h2f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// the weight of (layer 1..3, output row, k-slot) exactly as the stream orders it
__device__ __forceinline__ float pf2_weight(const float* w2, const float* w3, const float* wk, const float* wv, int layer, int orow, int fin, int W) {
  if (layer == 1) return w2[(size_t)orow * W + fin];
  if (layer == 2) return w3[(size_t)orow * W + fin];
  return orow < 128 ? wk[(size_t)orow * W + fin] : wv[(size_t)(orow - 128) * W + fin];
}

// MX mode: E8M0 scale bytes of every chunk's fp8 images.  One block per chunk c of layers 1..3 (c = NRT .. NC-1): the largest |f16(w)| and the largest
// |w - f16(w)| over the chunk's 32 rows x K, scale = the power of two that puts it at or below e4m3's 448.  sc[2 c] = w_hi8, sc[2 c + 1] = w_lo8.
__global__ void pf2_mx_scale_kernel(const float* __restrict__ w2, const float* __restrict__ w3, const float* __restrict__ wk, const float* __restrict__ wv,
                                    int* __restrict__ sc, int NRT) {
  const int W = 32 * NRT, c = blockIdx.x;
  __shared__ float smh[256], sml[256];
  float mh = 0.f, ml = 0.f;
  if (c >= NRT) {
    const int layer = c < 2 * NRT ? 1 : c < 3 * NRT ? 2 : 3, rt = c < 3 * NRT ? c % NRT : c - 3 * NRT;
    for (int i = threadIdx.x; i < 32 * W; i += blockDim.x) {
      const float v = pf2_weight(w2, w3, wk, wv, layer, 32 * rt + i / W, i % W, W);
      const float h = pf2_h2f(pf2_f2h(v));
      mh = fmaxf(mh, fabsf(h)); ml = fmaxf(ml, fabsf(v - h));
    }
  }
  smh[threadIdx.x] = mh; sml[threadIdx.x] = ml;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { smh[threadIdx.x] = fmaxf(smh[threadIdx.x], smh[threadIdx.x + st]); sml[threadIdx.x] = fmaxf(sml[threadIdx.x], sml[threadIdx.x + st]); }
    __syncthreads();
  }
  if (threadIdx.x < 2) {
    const float m = threadIdx.x == 0 ? smh[0] : sml[0];
    int e = -40;
    if (m > 0.f) { int ex; const float fr = frexpf(m / 448.f, &ex); e = fr == 0.5f ? ex - 1 : ex; }   // ceil(log2(m / 448))
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    sc[2 * c + threadIdx.x] = 127 + e;
  }
}

// MX mode: the constants the activation block scales are bounded from (point_fused2_kernel: xs_hi), 8 floats behind the chunks' scale bytes:
//   [0] c1a = max_c (sum |W1[c, sin / cos columns]| + r1 . sum |W1[c, ray-difference columns]|),  r1 >= |ray_diff_fc output| (its inputs are unit-vector
//       components and a cosine: |.| <= 1; LeakyReLU does not increase a magnitude)
//   [1] c1x = max_c sum |W1[c, the three raw-offset columns]|      [2] B2 = max_c ||W2[c, :]||_1   [3] max |b2|   [4] B3   [5] max |b3|
__global__ void pf2_mx_bounds_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3, const float* __restrict__ b2,
                                     const float* __restrict__ b3, const float* __restrict__ rd_w, float* __restrict__ out, int W, int F) {
  __shared__ float red[6][256];
  __shared__ float s_r1;
  const int t = threadIdx.x;
  if (t == 0) {
    float hmax = 0.f;
    for (int i = 0; i < 16; ++i) {
      float a = fabsf(rd_w[64 + i]);
      for (int jj = 0; jj < 4; ++jj) a += fabsf(rd_w[i * 4 + jj]);
      hmax = fmaxf(hmax, a);
    }
    float r1 = 0.f;
    for (int o = 0; o < 27; ++o) {
      float a = 0.f;
      for (int i = 0; i < 16; ++i) a += fabsf(rd_w[80 + o * 16 + i]);
      r1 = fmaxf(r1, a * hmax + fabsf(rd_w[80 + 432 + o]));
    }
    s_r1 = r1 * 1.01f;
  }
  __syncthreads();
  float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = t; c < W; c += 256) {
    const float* r = w1 + (size_t)c * (F + 90);
    float a = 0.f, x = 0.f, e = 0.f;
    for (int jj = 0; jj < 3; ++jj) x += fabsf(r[F + jj]);
    for (int jj = 3; jj < 63; ++jj) a += fabsf(r[F + jj]);
    for (int jj = 63; jj < 90; ++jj) e += fabsf(r[F + jj]);
    m[0] = fmaxf(m[0], a + e * s_r1); m[1] = fmaxf(m[1], x);
    float n2 = 0.f, n3 = 0.f;
    for (int k = 0; k < W; ++k) { n2 += fabsf(w2[(size_t)c * W + k]); n3 += fabsf(w3[(size_t)c * W + k]); }
    m[2] = fmaxf(m[2], n2); m[3] = fmaxf(m[3], fabsf(b2[c])); m[4] = fmaxf(m[4], n3); m[5] = fmaxf(m[5], fabsf(b3[c]));
  }
  for (int i = 0; i < 6; ++i) red[i][t] = m[i];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (t < st) for (int i = 0; i < 6; ++i) red[i][t] = fmaxf(red[i][t], red[i][t + st]);
    __syncthreads();
  }
  if (t < 8) out[t] = t < 6 ? red[t][0] * 1.0001f : 0.f;
}

// max |x| over n floats -> *out (as a float; *out must be zeroed first: non-negative floats order like their bit patterns)
__global__ void pf2_absmax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__global__ void pack_point_stream2_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                          const float* __restrict__ wk, const float* __restrict__ wv, const float* __restrict__ b2,
                                          const float* __restrict__ b3, const float* __restrict__ rd_w, unsigned short* __restrict__ out,
                                          int NRT, int F, int mx, const int* __restrict__ mxsc) {
  const int W = 32 * NRT, KSL = 2 * NRT;
  const long long n_l1 = (long long)NRT * 6 * 512, n_lw = (long long)NRT * KSL * 512, n_kv = (long long)8 * KSL * 512;
  const long long total = n_l1 + 2 * n_lw + n_kv;   // (chunk, k-step, lane, t) elements of one part
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) {
    int layer; long long r = e;
    if (r < n_l1) layer = 0; else if ((r -= n_l1) < n_lw) layer = 1; else if ((r -= n_lw) < n_lw) layer = 2; else { r -= n_lw; layer = 3; }
    const int nks = layer == 0 ? 6 : KSL;
    const int t = (int)(r & 7), lane = (int)((r >> 3) & 63);
    long long r2 = r >> 9;
    const int ks = (int)(r2 % nks), rt = (int)(r2 / nks);
    const int hh = lane >> 5, orow = 32 * rt + (lane & 31);
    float v = 0.f;
    if (layer == 0) {
      int col = -1;
      if (ks < 4) {
        const int ee = 8 * ks + t;
        if (ee < 30) { const int a = ee / 10, rem = ee - 10 * a, f = (rem >> 1) + 5 * hh, comp = rem & 1; col = F + 3 + 6 * f + 3 * comp + a; }
        else if (ee == 30) col = hh ? F + 2 : F;
        else col = hh ? -1 : F + 1;
      } else { const int o = pf2_m(8 * (ks - 4) + t, hh); col = o < 27 ? F + 63 + o : -1; }
      if (col >= 0) v = w1[(size_t)orow * (F + 90) + col];
    } else {
      const int fin = 32 * (ks >> 1) + pf2_m(8 * (ks & 1) + t, hh);
      if (layer == 1) v = w2[(size_t)orow * W + fin];
      else if (layer == 2) v = w3[(size_t)orow * W + fin];
      else v = orow < 128 ? wk[(size_t)orow * W + fin] : wv[(size_t)(orow - 128) * W + fin];
    }
    long long base;   // chunk base in bf16 elements (a chunk holds 2 parts x nks x 512)
    if (layer == 0) base = (long long)rt * 2 * 6 * 512;
    else base = (long long)NRT * 2 * 6 * 512 + ((long long)(layer - 1) * NRT + rt) * 2 * KSL * 512;
    const long long in_part = ((long long)ks * 64 + lane) * 8 + t;
    if (mx == 3 && (layer > 0 || L1MX)) out[base + in_part] = pf2_f2h(v);   // MX-FP6: the f16 fragments; pack_point_mx6_kernel writes the images and their scales
    else if (mx == 1 && layer > 0) {
      // MX chunk: part 0 = f16(w) in the same fragment order; part 1 = per slab q of 4 k-steps [w_hi8 bytes 0-15 | 16-31 | w_lo8 bytes 0-15 | 16-31][lane][16], byte
      // u = 8 (ks & 3) + t of a lane <-> this element; hi8 = e4m3(f16(w) / s_hi), lo8 = e4m3((w - f16(w)) / s_lo), scales per chunk (pf2_mx_scale_kernel)
      const int c = (layer - 1) * NRT + NRT + rt;   // chunk index (k / v heads: rt = 0..7 behind layer 3's base)
      const unsigned short hb = pf2_f2h(v);
      const float hf = pf2_h2f(hb);
      out[base + in_part] = hb;
      unsigned char* ob = reinterpret_cast<unsigned char*>(out + base + (long long)nks * 512);
      const int q = ks >> 2, sI = ks & 3;
      const float s_hi = ldexpf(1.f, mxsc[2 * c] - 127), s_lo = ldexpf(1.f, mxsc[2 * c + 1] - 127);
      const long long bo = (((long long)(4 * q + (sI >> 1)) * 64 + lane) * 16) + 8 * (sI & 1) + t;
      ob[bo] = pf2_e4m3(hf / s_hi);
      ob[bo + 2 * 64 * 16] = pf2_e4m3((v - hf) / s_lo);
    } else if (mx == 2) {   // split-FP16 stream (F16 mode): hi = f16(w), lo = f16(w - hi), every layer
      const unsigned short h = pf2_f2h(v);
      out[base + in_part] = h;
      out[base + (long long)nks * 512 + in_part] = pf2_f2h(v - pf2_h2f(h));
    } else {
      const unsigned short h = pf2_f2bf(v);
      out[base + in_part] = h;
      out[base + (long long)nks * 512 + in_part] = pf2_f2bf(v - __uint_as_float(((unsigned int)h) << 16));
    }
  }
  // resident block
  const long long res = 2LL * (NRT * 6 + (2 * NRT + 8) * KSL) * 512;   // bf16 elements of the stream
  if (e < 2 * 512) {   // ray_diff_fc fragments: rd_w = W0[16][4], b0[16], W2[27][16], b2[27]
    const int l = (int)(e >> 9), t = (int)(e & 7), lane = (int)((e >> 3) & 63), i = lane & 31, hh = lane >> 5;
    float v = 0.f;
    if (l == 0) { if (i < 16 && hh == 0 && t < 4) v = rd_w[i * 4 + t]; }
    else if (i < 27) v = rd_w[80 + i * 16 + pf2_m(t, hh)];
    if (mx == 2) {
      const unsigned short h = pf2_f2h(v);
      out[res + (long long)l * 1024 + lane * 8 + t] = h;
      out[res + (long long)l * 1024 + 512 + lane * 8 + t] = pf2_f2h(v - pf2_h2f(h));
    } else {
      const unsigned short h = pf2_f2bf(v);
      out[res + (long long)l * 1024 + lane * 8 + t] = h;
      out[res + (long long)l * 1024 + 512 + lane * 8 + t] = pf2_f2bf(v - __uint_as_float(((unsigned int)h) << 16));
    }
  }
  if (e < 64 + 2 * W) {   // bias tables in accumulator order [rt][hh][r]
    float* bt = reinterpret_cast<float*>(out + res + 2048);
    const int i = (int)e;
    float v;
    if (i < 64) { const int l = i >> 5, hh = (i >> 4) & 1, m = pf2_m(i & 15, hh); v = l == 0 ? (m < 16 ? rd_w[64 + m] : 0.f) : (m < 27 ? rd_w[80 + 432 + m] : 0.f); }
    else { const int q = i - 64, l = q / W, c = q - l * W, rt = c >> 5, hh = (c >> 4) & 1, f = 32 * rt + pf2_m(c & 15, hh); v = l == 0 ? b2[f] : b3[f]; }
    bt[i] = v;
  }
  if (mx == 1 && e < 2 * (3 * NRT + 8)) {   // the chunks' scale bytes behind the bias tables
    int* st = reinterpret_cast<int*>(out + res + 2048) + 64 + 2 * W;
    st[e] = mxsc[e];
  }
}

// MX-FP6 images of the wide chunks (layers 2, 3, k / v heads): one thread = one MX block = (chunk, slab q, lane, image).  The 32 weights of output row 32 rt + (lane & 31)
// whose k-slots belong to half lane >> 5 of slab q, in the POSITION order of the activation image they meet in the matrix instruction:
//   image 0 = w_hi6 = e2m3(f16(w)) meets the residual image (v_cvt_scalef32_2xpk16_fp6_f32: position P <- row tile 2 q + (P & 1), accumulator register P >> 1,
//             i.e. k-step 4 q + 2 (P & 1) + (P >> 4), element (P >> 1) & 7);
//   image 1 = w_lo6 = e2m3(w - f16(w)) meets the hi image (v_cvt_scalef32_pk32_fp6_f16: position P <- k-step 4 q + (P >> 3), element P & 7).
// Block scale 2^(floor(log2 max) - 2) (the largest magnitude lands in [4, 8); e2m3 saturates at 7.5: at most its own half-ulp), stored as E8M0 byte q of the lane's scale dword.
// Chunk image in the stream (32 KB): [f16 fragments 16 K][per slab: w_hi6 dwords 0-3 (1 K) | 4-5 (512) | w_lo6 dwords 0-3 | 4-5]; the scale dwords {w_hi6, w_lo6} per (wide chunk, lane): a table in the resident block
__device__ __forceinline__ unsigned pf2_e2m3(float a) {   // a >= 0, already divided by the block scale; round to nearest even, saturating
  if (!(a < 7.5f)) return 31u;
  if (a < 1.f) return (unsigned)rintf(a * 8.f);   // subnormals 0 .. 0.875; 8 = the smallest normal (encodings are contiguous)
  const int e = a < 2.f ? 0 : a < 4.f ? 1 : 2;
  unsigned m = (unsigned)rintf(ldexpf(a, 3 - e));   // 8 .. 16
  unsigned c = ((unsigned)(e + 1) << 3) + (m - 8u);    // m == 16 carries into the exponent
  return c > 31u ? 31u : c;
}
// (L1MX: the layer-1 chunks too — chunk index c = 0 .. NC - 1, layer-1 chunks hold 6 k-steps = slab 0 + half of slab 1, the k-slots of k-steps 6, 7 are zero)
__global__ void pack_point_mx6_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3, const float* __restrict__ wk,
                                      const float* __restrict__ wv, unsigned char* __restrict__ out, int NRT, int F) {
  const int W = 32 * NRT, KSL = 2 * NRT, NSL = NRT / 2;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int nwide = 2 * NRT + 8, nch = L1MX ? NRT + nwide : nwide;
  if (e >= nch * NSL * 64 * 2) return;
  const int im = e & 1, lane = (e >> 1) & 63, q = (e >> 7) % NSL, ci = (e >> 7) / NSL;
  const int c = L1MX ? ci : ci + NRT;              // chunk index 0 .. NC - 1
  const int cw = c - NRT;                          // wide chunk 0 .. 2 NRT + 7 (negative: a layer-1 chunk)
  if (c < NRT && q >= 2) return;                   // layer 1: two slabs
  const int layer = c < NRT ? 0 : cw < NRT ? 1 : cw < 2 * NRT ? 2 : 3, rt = c < NRT ? c : cw < 2 * NRT ? cw % NRT : cw - 2 * NRT;
  const int hh = lane >> 5, orow = 32 * rt + (lane & 31);
  float v[32], mx = 0.f;
  for (int P = 0; P < 32; ++P) {
    const int sI = im == 0 ? 2 * (P & 1) + (P >> 4) : (P >> 3), t = im == 0 ? (P >> 1) & 7 : (P & 7);
    const int ks = 4 * q + sI;
    float w = 0.f;
    if (layer == 0) {   // the column map of pack_point_stream2_kernel's layer-1 branch
      int col = -1;
      if (ks < 4) {
        const int ee = 8 * ks + t;
        if (ee < 30) { const int a = ee / 10, rem = ee - 10 * a, f = (rem >> 1) + 5 * hh, comp = rem & 1; col = F + 3 + 6 * f + 3 * comp + a; }
        else if (ee == 30) col = hh ? F + 2 : F;
        else col = hh ? -1 : F + 1;
      } else if (ks < 6) { const int o = pf2_m(8 * (ks - 4) + t, hh); col = o < 27 ? F + 63 + o : -1; }
      if (col >= 0) w = w1[(size_t)orow * (F + 90) + col];
    } else {
      const int fin = 32 * (ks >> 1) + pf2_m(8 * (ks & 1) + t, hh);
      w = pf2_weight(w2, w3, wk, wv, layer, orow, fin, W);
    }
    const float h = pf2_h2f(pf2_f2h(w));
    v[P] = im == 0 ? h : w - h;
    mx = fmaxf(mx, fabsf(v[P]));
  }
  int E = -60;
  if (mx > 0.f) { int ex; (void)frexpf(mx, &ex); E = ex - 1; }   // mx = 1.xxx 2^E
  int sb = E - 2 + 127;
  sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
  const float inv = ldexpf(1.f, 127 - sb);
  unsigned d[6] = {0u, 0u, 0u, 0u, 0u, 0u};
  for (int P = 0; P < 32; ++P) {
    const unsigned c = pf2_e2m3(fabsf(v[P]) * inv) | (v[P] < 0.f ? 32u : 0u);
    const int b = 6 * P;
    d[b >> 5] |= c << (b & 31);
    if ((b & 31) > 26) d[(b >> 5) + 1] |= c >> (32 - (b & 31));
  }
  // chunk base in the stream: the layer-1 chunks (12 KB each: 6 KB of f16 fragments + 2 slabs of images) first, then 32 KB per wide chunk
  unsigned char* cb = c < NRT ? out + (size_t)c * 2 * 6 * 1024 : out + (size_t)NRT * 2 * 6 * 1024 + (size_t)cw * 2 * KSL * 1024;
  const size_t f16b = c < NRT ? 6 * 1024 : (size_t)KSL * 1024;   // bytes of f16 fragments in front of the images
  unsigned* a = reinterpret_cast<unsigned*>(cb + f16b + (size_t)q * 3072 + (size_t)im * 1536 + (size_t)lane * 16);
  a[0] = d[0]; a[1] = d[1]; a[2] = d[2]; a[3] = d[3];
  unsigned* b2 = reinterpret_cast<unsigned*>(cb + f16b + (size_t)q * 3072 + (size_t)im * 1536 + 1024 + (size_t)lane * 8);
  b2[0] = d[4]; b2[1] = d[5];
  // the scale byte: resident table behind the stream's bias tables, scale ints and bounds block
  const int NC = 3 * NRT + 8;
  unsigned char* tab = out + ((size_t)NRT * 12 + (size_t)(2 * NRT + 8) * 2 * KSL) * 1024 + 4096 + (size_t)(64 + 2 * W) * 4 + (size_t)((2 * NC + 3) / 4 + 2) * 16;
  tab[((size_t)(L1MX ? c : cw) * 64 + lane) * 8 + im * 4 + q] = (unsigned char)sb;
}

}  // namespace

size_t nl_point_stream2_bytes(int W) {
  const int NRT = W / 32;
  return (size_t)2 * (NRT * 6 + (2 * NRT + 8) * 2 * NRT) * 1024 + 4096 + (size_t)(64 + 2 * W) * 4 + 4096   // + slack: bf16 L1 chunks copy 8 k-steps
         + (size_t)(3 * NRT + 8) * 512;                                                                      // + MX-FP6: the weight-scale table (every chunk)
}

// mx = 1: the stream of the MX mode (layer 1 split-bf16 as ever; layers 2, 3, k / v: f16 fragments + fp8 images + their scales); mx_scratch: >= 2 (3 W / 32 + 8) ints
// mx = 2: the split-FP16 stream (every layer hi / lo in fp16: the gradient path's forward)
int nl_pack_point_stream2(const float* w1, const float* w2, const float* w3, const float* wk, const float* wv, const float* b2, const float* b3,
                          const float* rd_w, void* out, int W, int F, hipStream_t st, int mx, int* mx_scratch) {
  const int NRT = W / 32;
  const long long total = ((long long)NRT * 6 + (2LL * NRT + 8) * 2 * NRT) * 512;
  const bool mx6 = mx == 1 && MXK == 2;   // NL_PREC_F16MX with fp6 cross terms (W = 256 has 28.5 of a chunk's 32 KB in use, W = 128 less)
  if (mx6) {
    if (W != 128 && W != 256) return NL_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_point_stream2_kernel, dim3((unsigned)nl_cdiv(total, 256)), dim3(256), 0, st, w1, w2, w3, wk, wv, b2, b3, rd_w, (unsigned short*)out, NRT, F, 3, nullptr);
    NL_LAUNCH_CHECK();
    hipLaunchKernelGGL(pack_point_mx6_kernel, dim3((unsigned)nl_cdiv((long long)(3 * NRT + 8) * (NRT / 2) * 128, 256)), dim3(256), 0, st, w1, w2, w3, wk, wv, (unsigned char*)out, NRT, F);
    NL_LAUNCH_CHECK();
    return NL_OK;
  }
  if (mx == 1) {
    if (!mx_scratch) return NL_ERR_BAD_ARG;
    hipLaunchKernelGGL(pf2_mx_scale_kernel, dim3(3 * NRT + 8), dim3(256), 0, st, w2, w3, wk, wv, mx_scratch, NRT);
    NL_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(pack_point_stream2_kernel, dim3((unsigned)nl_cdiv(total, 256)), dim3(256), 0, st, w1, w2, w3, wk, wv, b2, b3, rd_w,
                     (unsigned short*)out, NRT, F, mx, mx_scratch);
  NL_LAUNCH_CHECK();
  if (mx == 1) {   // the bounds block: behind the resident block's ray_diff_fc fragments (4 KB), bias tables and scale bytes (padded to 16 B)
    const int NC = 3 * NRT + 8;
    float* bnd = reinterpret_cast<float*>((char*)out + (size_t)2 * total * 2 + 4096 + (size_t)(64 + 2 * W) * 4 + (size_t)((2 * NC + 3) / 4) * 16);
    hipLaunchKernelGGL(pf2_mx_bounds_kernel, dim3(1), dim3(256), 0, st, w1, w2, w3, b2, b3, rd_w, bnd, W, F);
    NL_LAUNCH_CHECK();
  }
  return NL_OK;
}

// max |T| over a frame's table (the MX mode's bound on base_mlp.0's outputs): *out <- max |x[0 .. n)|
int nl_table_absmax(const float* x, size_t n, float* out, hipStream_t st) {
  NL_CHECK_HIP(hipMemsetAsync(out, 0, 4, st));
  if (n == 0) return NL_OK;
  const unsigned blocks = (unsigned)(n / 4096 + 1 > 1024 ? 1024 : n / 4096 + 1);
  hipLaunchKernelGGL(pf2_absmax_kernel, dim3(blocks), dim3(256), 0, st, x, n, reinterpret_cast<unsigned*>(out));
  NL_LAUNCH_CHECK();
  return NL_OK;
}

#ifdef PF2_TRACE
extern "C" __attribute__((visibility("default"))) int nl_debug_pf2_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pf2_trace), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1;
}
#endif

bool nl_point_fused2_supported(int W, int precision) { return precision != NL_PREC_F32 && (W == 128 || W == 256); }

// keep_kv != null (with the split-FP16 stream in a.wstream2): the F16 + KEEP instance — also writes the k / v rows (N x 8, 256) and the three layers' sign bits
int nl_launch_point_fused2(const NlPointFusedArgs& a, int W, int precision, hipStream_t st, bool mx, float* keep_kv, unsigned* const* keep_mk, const float* tmax,
                           unsigned* logit_amax, unsigned long long* clk) {
  if (a.N <= 0) return NL_OK;
  const int g_num_cu = nl_persistent_cus();
  if (g_num_cu < 0) return g_num_cu;
  Pf2Scalars sc;
  sc.dir_stride = a.dir_stride; sc.dir_div = a.dir_div > 0 ? a.dir_div : 1; sc.dir_magic = 0; sc.dir_shift = 0; sc.dir_one = sc.dir_div == 1 ? 1u : 0u;   // divisor 1: magic 0 (mulhi = 0) + n * 1
  if (sc.dir_div > 1) {   // n / d for n < 2^31 as mulhi(n, ceil(2^(31+s) / d)) >> (s - 1), s = ceil(log2 d)
    int s = 0;
    while ((1ll << s) < sc.dir_div) ++s;
    sc.dir_magic = (unsigned)(((1ull << (31 + s)) + (unsigned long long)sc.dir_div - 1) / (unsigned long long)sc.dir_div);
    sc.dir_shift = s - 1;
  }
  sc.N = a.N; sc.M = a.M; sc.inv_span = a.inv_span; sc.ntiles = (int)nl_cdiv(a.N, 16);
  if ((int64_t)a.N * 512 > 0x7fffffffll || ((int64_t)a.M + 1) * W * 4 > 0x7fffffffll) return NL_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  sc.t_bytes = (unsigned)(((int64_t)a.M + 1) * W * 4);
  sc.tmax = tmax; sc.logit_amax = logit_amax; sc.clk = clk;
  if (mx && !tmax) return NL_ERR_BAD_ARG;   // (the MX kernel bounds layer 1's outputs with it)
  const int nwg = sc.ntiles < g_num_cu ? (int)nl_xcd_grid(sc.ntiles) : g_num_cu;
  dim3 grid(nwg);
  const bool x3 = precision == NL_PREC_BF16X3;
  if (mx && !x3) return NL_ERR_BAD_ARG;
  Pf2Keep kp;
  memset(&kp, 0, sizeof(kp));
  if (keep_kv) {
    if (mx || !keep_mk || (int64_t)a.N * 8 * 1024 > 0x7fffffffll) return NL_ERR_UNSUPPORTED;
    kp.kv = keep_kv; kp.mk[0] = keep_mk[0]; kp.mk[1] = keep_mk[1]; kp.mk[2] = keep_mk[2];
    kp.kv_bytes = (unsigned)((int64_t)a.N * 8 * 1024);
    kp.mk_bytes = (unsigned)(nl_cdiv((int64_t)a.N * 8, 32) * 1024);
  }
#define NL_PF2(NRT)                                                                                                                                  \
  do {                                                                                                                                               \
    if (keep_kv) hipLaunchKernelGGL((point_fused2_kernel<NRT, true, 0, true, true>), grid, dim3(256), 0, st, a.xyz, a.dir, a.idx, a.Q, a.O, a.ptt, \
                                    a.sp_xyz, a.sp_dir, a.wstream2, sc, kp);                                                                         \
    else if (mx) hipLaunchKernelGGL((point_fused2_kernel<NRT, true, MXK>), grid, dim3(256), 0, st, a.xyz, a.dir, a.idx, a.Q, a.O, a.ptt, a.sp_xyz,   \
                                    a.sp_dir, a.wstream2, sc, kp);                                                                                   \
    else if (x3) hipLaunchKernelGGL((point_fused2_kernel<NRT, true, 0>), grid, dim3(256), 0, st, a.xyz, a.dir, a.idx, a.Q, a.O, a.ptt, a.sp_xyz, \
                                    a.sp_dir, a.wstream2, sc, kp);                                                                                   \
    else hipLaunchKernelGGL((point_fused2_kernel<NRT, false, 0>), grid, dim3(256), 0, st, a.xyz, a.dir, a.idx, a.Q, a.O, a.ptt, a.sp_xyz,        \
                            a.sp_dir, a.wstream2, sc, kp);                                                                                           \
  } while (0)
  if (W == 256) NL_PF2(8);
#if PF2_KO == 0
  else if (W == 128) NL_PF2(4);
#endif
  else return NL_ERR_UNSUPPORTED;
#undef NL_PF2
  NL_LAUNCH_CHECK();
  return NL_OK;
}
