// Streaming transposed segment-GEMM (bf16 / bf16x3 modes): the same contract as gemm.hip's kernels
//   C[out_row(m), n] = act( sum_seg sum_k A_seg[in_row_seg(m), k] * B[koff_seg + k, n] + bias[n] )
// but organised like point_fused.hip instead of a classic LDS-tiled GEMM:
//   * D^T = W . X^T : the WEIGHTS are the MFMA A operand, streamed from L2 through registers into a double-buffered LDS
//     slot in chunks of 32 k whose global image is already the A-fragment order (coalesced 16-B loads, one conflict-free
//     ds_read_b128 per fragment); every chunk is shared by all waves of the workgroup.  (An LDS-DMA ring measured the
//     same speed for N = 256 and slower for N <= 128.)
//   * the ACTIVATIONS are the B operand and never touch LDS: a wave owns 32 output rows, lane (j, hh) fetches the 8 floats
//     of row j it needs for a k-step straight from global memory (conv taps / concat / phase mapping = address arithmetic)
//     one chunk ahead and splits them to bf16 hi/lo in registers.
//   * all N <= 256 output columns of a row live in one wave's accumulators (N/32 tiles of 32x32), so X is read once.
//   * <= 256 registers and 64 KB of LDS: two workgroups per CU, which run out of phase and hide each other's memory
//     latency, conversion VALU and barrier time behind MFMAs; no data-dependent control flow around the MFMAs.
#include <utility>
#include "common.h"

typedef __bf16 tg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float tg_f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ void tg_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void tg_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
template <int... Is, class F>
__device__ __forceinline__ void tg_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void tg_static_for(F&& f) {
  tg_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <bool X3>
__device__ __forceinline__ void tg_split8(const float (&v)[8], tg_bf16x8& hi, tg_bf16x8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    __bf16 h = (__bf16)v[t];
    hi[t] = h;
    if (X3) lo[t] = (__bf16)(v[t] - (float)h);
  }
}

// chunk c (k-space [32c, 32c+32)) lies in exactly one segment (every segment is a multiple of 32 wide here);
// kstart[] are the segments' first k (INT_MAX for unused slots) -> branch-free, wave-uniform lookup
__device__ __forceinline__ int tg_find_seg(const NlGemmArgs& a, int k0) {
  int s = 0;
#pragma unroll
  for (int j = 1; j < NL_GEMM_MAX_SEG; ++j) s += (k0 >= a.kstart[j]) ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(s);
}

template <int NRT, int NW, bool X3, int EPI>
__global__ __launch_bounds__(64 * NW, 2) void tgemm_kernel(const NlGemmArgs a, const char* __restrict__ p_bst, float* __restrict__ p_c,
                                                            const float* __restrict__ p_zeros, const float* __restrict__ p_bias) {
  constexpr int PARTS = X3 ? 2 : 1;
  constexpr int PIECES = PARTS * 2 * NRT;   // 1-KB pieces of weights per chunk
  constexpr int NPW = PIECES / NW;          // pieces staged by each wave
  static_assert(PIECES % NW == 0, "pieces must split evenly over the waves");
  constexpr int CH16 = 4 * NRT * 64;        // 16-B units per chunk in the global stream (hi and lo parts are always stored)
  constexpr int SLOT16 = PIECES * 64;       // 16-B units per LDS slot
  __shared__ uint4 lds_all[2 * SLOT16 + NRT * 8 * (EPI == NL_EPI_LNROW ? 3 : 1) + (EPI == NL_EPI_LNSLAB ? 8 : 0)];
  // native vector element type everywhere (struct-typed uint4 arrays in registers do not survive SROA)
  tg_bf16x8 (*ring)[SLOT16] = reinterpret_cast<tg_bf16x8 (*)[SLOT16]>(lds_all);
  float* sbias = reinterpret_cast<float*>(lds_all + 2 * SLOT16);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31;
  int tile = blockIdx.x * NW + wave;
  if (a.tile_map) {   // compacted tile list: workgroups past the end leave before any barrier; a partial last workgroup repeats its last tile
    const int nt = *a.tile_count;
    if ((int)blockIdx.x * NW >= nt) return;
    tile = a.tile_map[tile < nt ? tile : nt - 1];
  }
  const int m = tile * 32 + j;
  const bool mok = m < a.M;
  int q = 0, t = 0;
  if (a.So > 0) { q = m / a.So; t = m - q * a.So; }
  const int NC = a.Kpad >> 5;

  for (int i = tid; i < NRT * 32; i += 64 * NW) {
    sbias[i] = (p_bias && i < a.N) ? p_bias[i] : 0.f;
    if (EPI == NL_EPI_LNROW) { sbias[NRT * 32 + i] = a.ep_gamma[i]; sbias[2 * NRT * 32 + i] = a.ep_beta[i]; }   // N == 32 * NRT
  }

  // weights of chunk c: this wave's NPW pieces, 16 B per lane, fully coalesced
  auto load_w = [&](int c, tg_bf16x8 (&w)[NPW]) __attribute__((always_inline)) {
    const tg_bf16x8* src = reinterpret_cast<const tg_bf16x8*>(p_bst) + (size_t)c * CH16;
#pragma unroll
    for (int jj = 0; jj < NPW; ++jj) w[jj] = src[(wave + NW * jj) * 64 + lane];
  };
  auto store_w = [&](int slot, const tg_bf16x8 (&w)[NPW]) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < NPW; ++jj) ring[slot][(wave + NW * jj) * 64 + lane] = w[jj];
  };
  // activation fragments of chunk c: 2 k-steps x 8 floats of this lane's source row, straight into registers.  Conv halo
  // rows and rows >= M read a device zero page instead (no masking arithmetic).
  auto load_act = [&](int c, float4 (&raw)[4]) __attribute__((always_inline)) {
    const int k0 = 32 * c;
    const int s = tg_find_seg(a, k0);
    const NlGemmSeg& sg = a.seg[s];
    int kbase = k0 - a.kstart[s];
    int ioff = sg.ioff;
    if (sg.ntap > 1) {   // [32-channel block][tap][32]: chunk cc of the segment = (block cc / ntap, tap cc % ntap)
      const int cc = kbase >> 5, cb = cc / sg.ntap;
      ioff += cc - cb * sg.ntap - (sg.ntap >> 1);
      kbase = cb << 5;
    }
    bool ok = mok;
    int row = m;
    if (a.So > 0) {
      const int i = t + ioff;
      ok = ok && i >= 0 && i < a.Li;
      row = q * a.Li + i;
    }
    const float* p = (ok ? sg.ptr + (size_t)row * sg.ld + kbase : p_zeros) + 8 * hh;
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) raw[pc] = *(const float4*)(p + 16 * (pc >> 1) + 4 * (pc & 1));
  };
  auto convert = [&](const float4 (&raw)[4], tg_bf16x8 (&bh)[2], tg_bf16x8 (&bl)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float v[8] = {raw[2 * ks].x, raw[2 * ks].y, raw[2 * ks].z, raw[2 * ks].w,
                          raw[2 * ks + 1].x, raw[2 * ks + 1].y, raw[2 * ks + 1].z, raw[2 * ks + 1].w};
      tg_split8<X3>(v, bh[ks], bl[ks]);
    }
  };

  tg_f32x16 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

  // one chunk: 2 k-steps x NRT row tiles x (3 | 1) MFMAs; A fragments are read two (k-step, tile) pairs ahead
  auto compute = [&](int slot, const tg_bf16x8 (&bh)[2], const tg_bf16x8 (&bl)[2]) __attribute__((always_inline)) {
    const tg_bf16x8* L = ring[slot];
    constexpr int nt = 2 * NRT;
    auto ldA = [&](int tt, tg_bf16x8& ah, tg_bf16x8& al) __attribute__((always_inline)) {
      const int ks = tt / NRT, rt = tt - ks * NRT;
      ah = L[((0 * 2 + ks) * NRT + rt) * 64 + lane];
      if (X3) al = L[((1 * 2 + ks) * NRT + rt) * 64 + lane];
    };
    tg_bf16x8 ah[3], al[3];
    ldA(0, ah[0], al[0]);
    ldA(1, ah[1], al[1]);
#pragma unroll
    for (int tt = 0; tt < nt; ++tt) {
      if (tt + 2 < nt) ldA(tt + 2, ah[(tt + 2) % 3], al[(tt + 2) % 3]);
      const int ks = tt / NRT, rt = tt - ks * NRT;
      if (X3) {
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tt % 3], bh[ks], acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tt % 3], bl[ks], acc[rt], 0, 0, 0);
      }
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tt % 3], bh[ks], acc[rt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Register-staged pipeline, one chunk ahead, two workgroups per CU (<= 256 registers): while chunk g is multiplied out of
  // LDS slot g%2, chunk g+1's weights and activation fragments are in flight to registers; afterwards the weights go to
  // slot (g+1)%2 — every wave left that slot before the previous barrier.  The second workgroup of the CU runs the same
  // loop out of phase and covers this one's latencies, VALU and barrier time with its MFMAs.
  // The chunk past the end re-loads the last real chunk (never used): no data-dependent control flow around the MFMAs
  // (the accumulators must stay in AGPRs).
  tg_bf16x8 wreg[NPW];
  float4 raw[4];
  auto clampc = [&](int c) { return c < NC ? c : NC - 1; };
  load_act(0, raw); load_w(0, wreg);
  store_w(0, wreg);
  __syncthreads();
  for (int g = 0; g < NC; ++g) {   // one straight-line body, no control flow around the MFMAs (accumulators stay put)
    tg_bf16x8 bh[2], bl[2];
    convert(raw, bh, bl);
    load_act(clampc(g + 1), raw);
    load_w(clampc(g + 1), wreg);
    compute(g & 1, bh, bl);
    store_w((g + 1) & 1, wreg);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31 (= this lane's output row), reg r = 4*gq + e <-> n = 32*rt + 8*gq + 4*hh + e
  if constexpr (EPI == NL_EPI_LNSLAB) {
    // The workgroup's 32*NW rows are one, two, four or eight whole rays (launch precondition: So in {128, 64, 32, 16}, M % So == 0; `wpr`
    // waves per ray, or two rays per wave for So = 16): LayerNorm over each ray's whole (So x N) slab with per-(position, channel) affine, ELU, optional
    // MaxPool(2) along the ray.  A trailing workgroup may hold rays past M: their statistics are computed on zero rows and
    // nothing of them is stored.
    float* red = sbias + NRT * 32;   // [2][NW] partial sums
    const int wpr = a.So >= 32 ? a.So >> 5 : 1, gb = (wave / wpr) * wpr;
    float s1 = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = 32 * rt + 8 * gq + 4 * hh;
        if (n < a.N) {
          const float4 b4 = *(const float4*)(sbias + n);
          acc[rt][4 * gq + 0] += b4.x; acc[rt][4 * gq + 1] += b4.y; acc[rt][4 * gq + 2] += b4.z; acc[rt][4 * gq + 3] += b4.w;
          s1 += (acc[rt][4 * gq + 0] + acc[rt][4 * gq + 1]) + (acc[rt][4 * gq + 2] + acc[rt][4 * gq + 3]);
        }
      }
    // a ray of 16 rows is half a wave (lanes j = 0..15 or 16..31 of both halves hh): summed with four DPP permutations inside the
    // 16-lane row + the partner row 32 lanes away; no LDS round trip
    auto ray16_sum = [](float v) __attribute__((always_inline)) {
      v += nl_dpp<0xB1>(v); v += nl_dpp<0x4E>(v); v += nl_dpp<0x141>(v); v += nl_dpp<0x140>(v);   // quad_perm x2, row_half_mirror, row_mirror
      return v + __shfl_xor(v, 32, 64);
    };
    float tot = 0.f;
    if (a.So == 16) tot = ray16_sum(s1);
    else {
      s1 = wave_sum(s1);
      if (lane == 0) red[wave] = s1;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += w < wpr ? red[gb + (w < wpr ? w : 0)] : 0.f;
    }
    const float cnt = (float)a.So * (float)a.N;
    const float mean = tot / cnt;
    float s2 = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        if (32 * rt + 8 * gq + 4 * hh < a.N) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = acc[rt][4 * gq + e] - mean; s2 += d * d; }
        }
    float tot2 = 0.f;
    if (a.So == 16) tot2 = ray16_sum(s2);
    else {
      s2 = wave_sum(s2);
      if (lane == 0) red[NW + wave] = s2;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NW; ++w) tot2 += w < wpr ? red[NW + gb + (w < wpr ? w : 0)] : 0.f;
    }
    const float rstd = 1.f / sqrtf(tot2 / cnt + a.ep_eps);
    const float* grow = a.ep_gamma + (size_t)t * a.N;
    const float* brow = a.ep_beta + (size_t)t * a.N;
    const bool pool = a.ep_pool != 0;
    float* orow_p = p_c + (size_t)(pool ? q * (a.So / 2) + (t >> 1) : m) * a.ldc;
    float sg = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = 32 * rt + 8 * gq + 4 * hh;
        if (n < a.N) {
          const float4 g4 = *(const float4*)(grow + n), be4 = *(const float4*)(brow + n);
          float4 v;
          v.x = nl_elu_fast((acc[rt][4 * gq + 0] - mean) * rstd * g4.x + be4.x);
          v.y = nl_elu_fast((acc[rt][4 * gq + 1] - mean) * rstd * g4.y + be4.y);
          v.z = nl_elu_fast((acc[rt][4 * gq + 2] - mean) * rstd * g4.z + be4.z);
          v.w = nl_elu_fast((acc[rt][4 * gq + 3] - mean) * rstd * g4.w + be4.w);
          if (pool) {   // positions 2p, 2p+1 are neighbouring lanes
            v.x = fmaxf(v.x, nl_dpp<0xB1>(v.x, v.x)); v.y = fmaxf(v.y, nl_dpp<0xB1>(v.y, v.y));   // quad_perm [1,0,3,2]: lane ^ 1
            v.z = fmaxf(v.z, nl_dpp<0xB1>(v.z, v.z)); v.w = fmaxf(v.w, nl_dpp<0xB1>(v.w, v.w));
          }
          if (p_c && mok && (!pool || !(j & 1))) *(float4*)(orow_p + n) = v;   // p_c == null: only the density head's output is wanted
          if (a.ep_sig_w) {
            const float4 w4 = *(const float4*)(a.ep_sig_w + n);
            sg = fmaf(v.x, w4.x, sg); sg = fmaf(v.y, w4.y, sg); sg = fmaf(v.z, w4.z, sg); sg = fmaf(v.w, w4.w, sg);
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the gamma/beta loads of later tiles from being hoisted (register budget)
      }
    if (a.ep_sig_w) {   // the density head rides along: the other half of the row is in lane ^ 32
      sg += __shfl_xor(sg, 32, 64);
      if (hh == 0 && mok) a.ep_sig_out[m] = nl_softplus(sg + a.ep_sig_b[0]);
    }
    return;
  }
  if (!mok) return;
  size_t orow;
  if (a.So > 0) orow = (size_t)(q * a.Lo + t * a.ostride + a.ooff) * a.ldc;
  else orow = (size_t)m * a.ldc;
  float* crow = p_c + orow;
  if constexpr (EPI == NL_EPI_LNROW) {
    // LayerNorm over the N = 32*NRT outputs of this lane's row (the other half of the row lives in lane ^ 32), after adding
    // the residual row; two-pass mean / variance like torch
    const float* rrow = a.ep_res + (size_t)m * a.ep_ldres;
    float s1 = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = 32 * rt + 8 * gq + 4 * hh;
        const float4 b4 = *(const float4*)(sbias + n);
        const float4 r4 = *(const float4*)(rrow + n);
        acc[rt][4 * gq + 0] += b4.x + r4.x; acc[rt][4 * gq + 1] += b4.y + r4.y;
        acc[rt][4 * gq + 2] += b4.z + r4.z; acc[rt][4 * gq + 3] += b4.w + r4.w;
        s1 += (acc[rt][4 * gq + 0] + acc[rt][4 * gq + 1]) + (acc[rt][4 * gq + 2] + acc[rt][4 * gq + 3]);
      }
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 / (float)(32 * NRT);
    float s2 = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[rt][r] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = 1.f / sqrtf(s2 / (float)(32 * NRT) + a.ep_eps);
    const float sc = a.ep_scale ? a.ep_scale[m] : 1.f;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = 32 * rt + 8 * gq + 4 * hh;
        const float4 g4 = *(const float4*)(sbias + NRT * 32 + n), be4 = *(const float4*)(sbias + 2 * NRT * 32 + n);
        float4 v;
        v.x = ((acc[rt][4 * gq + 0] - mean) * rstd * g4.x + be4.x) * sc;
        v.y = ((acc[rt][4 * gq + 1] - mean) * rstd * g4.y + be4.y) * sc;
        v.z = ((acc[rt][4 * gq + 2] - mean) * rstd * g4.z + be4.z) * sc;
        v.w = ((acc[rt][4 * gq + 3] - mean) * rstd * g4.w + be4.w) * sc;
        *(float4*)(crow + n) = v;
      }
    return;
  }
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = 32 * rt + 8 * gq + 4 * hh;
      if (n < a.N) {   // N % 4 == 0 and ldc % 4 == 0 are launch preconditions
        const float4 b4 = *(const float4*)(sbias + n);
        float4 v;
        v.x = nl_act(acc[rt][4 * gq + 0] + b4.x, a.act);
        v.y = nl_act(acc[rt][4 * gq + 1] + b4.y, a.act);
        v.z = nl_act(acc[rt][4 * gq + 2] + b4.z, a.act);
        v.w = nl_act(acc[rt][4 * gq + 3] + b4.w, a.act);
        *(float4*)(crow + n) = v;
      }
    }
}


// ====================================================================================================================
// Per-sample chain after the neural-point branch (W = 256):   feature_agg = LayerNorm(fc(O) + G) * wscale     (ibrnet.py:110-117,
//   model.py:419-427)  ->  feat_mlp.0 + LeakyReLU (model.py:85-89)  and  the feature_agg columns of rgb_blending_mlp.0 (model.py:532).
// As three launches feature_agg (N x W fp32) is written once and read back twice (1.07 GB per config-2 batch) by kernels that are
// HBM-bound.  Here a wave keeps its 32 rows: the LayerNorm output stays in the accumulator registers, is split to bf16 hi/lo in place
// and IS the B operand of the next two products (their weight streams are packed with K in accumulator order, like the fused
// neural-point kernel); feature_agg is still written (the ray U-Net reads it), but never read back here.  One workgroup per CU
// (the two activation sets + 128 accumulators need the 512-register file); the K-outer chunk pipeline of tgemm_kernel otherwise.
//
// The kernel runs one of three chunk programs (MODE):
//   0: fc | feat_mlp.0 | blend projection, the residual rows G = multiview feature (N x W) read from memory
//   1: out_fc.2 | fc | feat_mlp.0 | blend projection: G = ELU(out_fc.2(t64)) (ibrnet.py:104-106 `out_fc`) is recomputed from the 64-wide
//      hidden rows and lands in the accumulators fc continues on — G is never written or read (0.5 GB + 0.5 GB per config-2 batch)
//   2: out_fc.2 | w_qs: the attention query rows Q = w_qs(G) (model.py:391-396) from the same recomputed G, for the neural-point kernel
struct NlChainArgs {
  const float* O; const float* G; const float* wscale; const float* gamma; const float* beta; float eps;
  const char* wbase; unsigned off_fc, off_f0, off_ba;   // weight streams as byte offsets into one packed blob
  const float* bias_f0;
  float* FA; float* fth; float* blA;
  int M;
  const float* T64; unsigned off_g2, off_q; const float* bias_g2; float* Q;   // MODE 1, 2
};

typedef unsigned int tg_u32x4 __attribute__((ext_vector_type(4)));
typedef float tg_f32x4 __attribute__((ext_vector_type(4)));

enum { CK_G2 = 0, CK_FC, CK_F0, CK_BL, CK_Q, CK_NOP };

// weight chunks (32 k each) per 128-row tile, e.g. MODE 0: fc 0..3 | feat_mlp.0 4..11 | blend projection 12..19
template <int MODE>
struct ChainGeo {
  static constexpr int NST = MODE == 0 ? 3 : MODE == 1 ? 4 : 2;
  static constexpr int kind_at(int s) {
    return MODE == 0 ? (s == 0 ? CK_FC : s == 1 ? CK_F0 : CK_BL) : MODE == 1 ? (s == 0 ? CK_G2 : s == 1 ? CK_FC : s == 2 ? CK_F0 : CK_BL) : (s == 0 ? CK_G2 : CK_Q);
  }
  static constexpr int nch_k(int k) { return k == CK_G2 ? 2 : k == CK_FC ? 4 : 8; }
  static constexpr int nrt_k(int k) { return k == CK_NOP ? 0 : k == CK_BL ? 2 : k == CK_Q ? 4 : 8; }
  static constexpr int start(int s) { int c = 0; for (int i = 0; i < s; ++i) c += nch_k(kind_at(i)); return c; }
  // the ring slot of a chunk is c % 4 at compile time: programs are padded with empty chunks (barrier only) to a multiple of 4
  static constexpr int NREAL = start(NST), NCH = (NREAL + 3) / 4 * 4;
  static constexpr int cm(int c) { return ((c % NCH) + NCH) % NCH; }
  static constexpr int spos(int c) { int s = 0; for (int i = 1; i < NST; ++i) if (cm(c) >= start(i)) s = i; return s; }
  static constexpr int kind(int c) { return cm(c) >= NREAL ? CK_NOP : kind_at(spos(c)); }
  static constexpr int idx(int c) { return cm(c) - start(spos(c)); }
  static constexpr int nrt(int c) { return nrt_k(kind(c)); }
  static constexpr bool last(int c) { return kind(c) != CK_NOP && idx(c) == nch_k(kind(c)) - 1; }
  // the next tile's input rows are fetched in the first chunk of the last stage (MODE 0: 16 attention-output + 32 residual loads;
  // MODE 1: 16 + 8 hidden-row loads; MODE 2: 8)
  static constexpr int PF = start(NST - 1);
  static constexpr int NPF = MODE == 0 ? 48 : MODE == 1 ? 24 : 8;
  // VMEM operations other than LDS-DMA pieces issued in chunk c's slot, after its own pieces went out (per lane-instruction): the
  // prefetch above; after the fc: wscale load + 32 feature_agg stores; after feat_mlp.0: 32 stores; after the blend projection: 4
  // stores; after w_qs: 16 stores.  Waits count them: vmcnt retires in issue order.
  static constexpr int post(int c, bool feat) {
    int n = cm(c) == PF ? NPF : 0;
    if (last(c)) { const int k = kind(c); n += k == CK_FC ? 33 : k == CK_F0 ? (feat ? 32 : 0) : k == CK_BL ? 4 : k == CK_Q ? 16 : 0; }
    return n;
  }
};

template <bool X3, bool FEAT, int MODE>
__global__ __launch_bounds__(256, 1) void sample_chain_kernel(const NlChainArgs a, const int ntiles) {
  using Geo = ChainGeo<MODE>;
  constexpr int NW = 4, PARTS = X3 ? 2 : 1, NCH = Geo::NCH, NB = 4;
  constexpr int SLOT16 = PARTS * 2 * 8 * 64;   // 16-B units per ring slot (sized for 8 row tiles)
  __shared__ uint4 lds_all[NB * SLOT16 + 4 * 64];
  tg_bf16x8 (*ring)[SLOT16] = reinterpret_cast<tg_bf16x8 (*)[SLOT16]>(lds_all);
  float* stab = reinterpret_cast<float*>(lds_all + NB * SLOT16);   // gamma | beta | feat_mlp.0 bias | out_fc.2 bias
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31;
  int tile = (int)nl_xcd_block();
  if (tile >= ntiles) return;
  // stores go through buffer descriptors: rows past M carry an out-of-range offset and are dropped, so every store instruction is
  // always issued and the vmcnt bookkeeping below is exact
  const __amdgpu_buffer_rsrc_t rFA = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == 2 ? a.Q : a.FA), 0, a.M * (MODE == 2 ? 512 : 1024), 0x00020000);
  const __amdgpu_buffer_rsrc_t rFT = __builtin_amdgcn_make_buffer_rsrc((void*)(FEAT ? a.fth : a.FA), 0, a.M * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBL = __builtin_amdgcn_make_buffer_rsrc((void*)a.blA, 0, a.M * 128, 0x00020000);
  for (int i = tid; i < 256; i += 256) {
    if (MODE != 2) { stab[i] = a.gamma[i]; stab[256 + i] = a.beta[i]; stab[512 + i] = a.bias_f0 ? a.bias_f0[i] : 0.f; }
    if (MODE != 0) stab[768 + i] = a.bias_g2[i];
  }

  // weight chunks by LDS-DMA into a 4-slot ring, three chunks ahead (buffer form: see point_fused2.hip); piece p = 4 i + wave
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wbase, 0, 0x7fffffff, 0x00020000);
  const unsigned wvoff = wave * 1024 + lane * 16;
  uint4* lw = lds_all + wave * 64;
  auto ppw = [](int c) constexpr { return PARTS * 2 * Geo::nrt(c) / NW; };
  auto dma_chunk = [&](auto Cc) __attribute__((always_inline)) {
    constexpr int c = Geo::cm(decltype(Cc)::value), kd = Geo::kind(c), nrt = Geo::nrt(c);
    unsigned so = (kd == CK_FC ? a.off_fc : kd == CK_F0 ? a.off_f0 : kd == CK_BL ? a.off_ba : kd == CK_G2 ? a.off_g2 : a.off_q) +
                  (unsigned)(Geo::idx(c) * 4 * nrt * 1024);
    tg_static_for<ppw(c)>([&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value;
      unsigned s2 = so + i * 4096;
      asm volatile("" : "+s"(s2));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lw + (c % NB) * SLOT16 + i * 256), 16, wvoff, s2, 0, 0);
    });
  };

  tg_f32x16 acc[8];
  tg_bf16x8 Xh[16], Xl[16];
  tg_f32x4 oraw[MODE == 2 ? 1 : 16];
  tg_f32x4 traw[MODE == 0 ? 1 : 8];
  int m = 0, mm = 0;
  bool mok = false;
  // MODE 0: attention output rows (fc's B operand: 8 k-steps x 8 floats per lane) and the residual rows as fc's accumulator init.
  // The blend projection only uses accumulators 0 and 1: the next tile's inputs are fetched while it runs — residual rows of row
  // tiles 2..7 straight into their accumulators, those of row tiles 0, 1 into `gtmp` (moved over when the projection is stored).
  // MODE 1, 2: the 64-wide hidden rows (out_fc.2's B operand: 4 k-steps x 8 floats per lane) instead of the residual rows.
  tg_f32x4 gtmp[MODE == 0 ? 8 : 1];
  auto load_tile_inputs = [&](int t) __attribute__((always_inline)) {
    m = t * 128 + 32 * wave + j;
    mok = m < a.M;
    mm = mok ? m : a.M - 1;
    if constexpr (MODE != 2) {
      const float* p = a.O + (size_t)mm * 128 + 8 * hh;
#pragma unroll
      for (int i = 0; i < 16; ++i) oraw[i] = *reinterpret_cast<const tg_f32x4*>(p + 16 * (i >> 1) + 4 * (i & 1));
    }
    if constexpr (MODE == 0) {
      const float* rrow = a.G + (size_t)mm * 256 + 4 * hh;
#pragma unroll
      for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const tg_f32x4 r4 = *reinterpret_cast<const tg_f32x4*>(rrow + 32 * rt + 8 * gq);
          if (rt < 2) gtmp[4 * rt + gq] = r4;
          else { acc[rt][4 * gq] = r4[0]; acc[rt][4 * gq + 1] = r4[1]; acc[rt][4 * gq + 2] = r4[2]; acc[rt][4 * gq + 3] = r4[3]; }
        }
    } else {
      const float* p = a.T64 + (size_t)mm * 64 + 8 * hh;
#pragma unroll
      for (int i = 0; i < 8; ++i) traw[i] = *reinterpret_cast<const tg_f32x4*>(p + 16 * (i >> 1) + 4 * (i & 1));
    }
  };
  auto adopt_gtmp = [&]() __attribute__((always_inline)) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const tg_f32x4 r4 = gtmp[4 * rt + gq];
          acc[rt][4 * gq] = r4[0]; acc[rt][4 * gq + 1] = r4[1]; acc[rt][4 * gq + 2] = r4[2]; acc[rt][4 * gq + 3] = r4[3];
        }
    }
  };

  // one chunk (32 k): 2 k-steps x NRT row tiles x (3 | 1) MFMAs out of ring slot `slot`
  auto compute = [&](auto Nc, int slot, const tg_bf16x8 (&bh)[2], const tg_bf16x8 (&bl)[2]) __attribute__((always_inline)) {
    constexpr int NRT = decltype(Nc)::value, nt = 2 * NRT;
    const tg_bf16x8* L = ring[slot];
    auto ldA = [&](int tt, tg_bf16x8& ah, tg_bf16x8& al) __attribute__((always_inline)) {
      const int ks = tt / NRT, rt = tt - ks * NRT;
      ah = L[((0 * 2 + ks) * NRT + rt) * 64 + lane];
      if (X3) al = L[((1 * 2 + ks) * NRT + rt) * 64 + lane];
    };
    tg_bf16x8 ah[3], al[3];
    ldA(0, ah[0], al[0]);
    ldA(1, ah[1], al[1]);
#pragma unroll
    for (int tt = 0; tt < nt; ++tt) {
      if (tt + 2 < nt) ldA(tt + 2, ah[(tt + 2) % 3], al[(tt + 2) % 3]);
      const int ks = tt / NRT, rt = tt - ks * NRT;
      if (X3) {
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tt % 3], bh[ks], acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tt % 3], bl[ks], acc[rt], 0, 0, 0);
      }
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tt % 3], bh[ks], acc[rt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto zero_acc = [&](auto N0, auto N1) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = decltype(N0)::value; rt < decltype(N1)::value; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
  };
  using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I8 = std::integral_constant<int, 8>;

  // ---------------------------------------------------------------- pipeline start
  dma_chunk(std::integral_constant<int, 0>{}); dma_chunk(std::integral_constant<int, 1>{}); dma_chunk(std::integral_constant<int, 2>{});
  if constexpr (MODE != 0) zero_acc(I0{}, I8{});
  load_tile_inputs(tile);
  adopt_gtmp();
  tg_wait_vmcnt<0>();   // the first pass of the counted waits below assumes nothing older is in flight
  __syncthreads();      // stab

  for (;;) {
    const bool mok_c = mok;
    const int m_c = m, mm_c = mm;
    const int tile_next = tile + (int)gridDim.x;
    tg_static_for<NCH>([&](auto Cc) __attribute__((always_inline)) {
      constexpr int c = decltype(Cc)::value, kd = Geo::kind(c), g = Geo::idx(c);
      // chunk c must have landed: younger operations are the pieces of chunks c+1, c+2 and the other traffic issued since chunk c-3
      constexpr int younger = ppw(c + 1) + ppw(c + 2) + Geo::post(c - 3, FEAT) + Geo::post(c - 2, FEAT) + Geo::post(c - 1, FEAT);
      tg_wait_vmcnt<(younger < 63 ? younger : 63)>();
      __builtin_amdgcn_s_barrier();
      dma_chunk(std::integral_constant<int, c + 3>{});   // its slot held chunk c-1, which every wave has left
      if constexpr (c == Geo::PF) load_tile_inputs(tile_next < ntiles ? tile_next : tile);   // (always issued: the wait counts stay exact)
      tg_bf16x8 bh[2], bl[2];
      if constexpr (kd == CK_FC || kd == CK_G2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          tg_f32x4 u0, u1;
          if constexpr (kd == CK_FC) { u0 = oraw[(4 * g + 2 * ks) % (MODE == 2 ? 1 : 16)]; u1 = oraw[(4 * g + 2 * ks + 1) % (MODE == 2 ? 1 : 16)]; }
          else { u0 = traw[(4 * g + 2 * ks) % (MODE == 0 ? 1 : 8)]; u1 = traw[(4 * g + 2 * ks + 1) % (MODE == 0 ? 1 : 8)]; }
          const float v[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
          tg_split8<X3>(v, bh[ks], bl[ks]);
        }
        compute(I8{}, c % NB, bh, bl);
      } else if constexpr (kd != CK_NOP) {
        bh[0] = Xh[2 * g]; bh[1] = Xh[2 * g + 1]; bl[0] = Xl[2 * g]; bl[1] = Xl[2 * g + 1];
        if constexpr (kd == CK_F0) { if constexpr (FEAT) compute(I8{}, c % NB, bh, bl); }
        else if constexpr (kd == CK_BL) compute(I2{}, c % NB, bh, bl);
        else if constexpr (kd == CK_Q) compute(std::integral_constant<int, 4>{}, c % NB, bh, bl);
      }
      if constexpr (kd == CK_G2 && Geo::last(c)) {
        // ---- G = ELU(out_fc.2 + bias): stays in the accumulators as fc's residual (MODE 1) | becomes w_qs' B operand (MODE 2)
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const float4 b4 = *(const float4*)(stab + 768 + 32 * rt + 8 * gq + 4 * hh);
            acc[rt][4 * gq + 0] = nl_elu_fast(acc[rt][4 * gq + 0] + b4.x);
            acc[rt][4 * gq + 1] = nl_elu_fast(acc[rt][4 * gq + 1] + b4.y);
            acc[rt][4 * gq + 2] = nl_elu_fast(acc[rt][4 * gq + 2] + b4.z);
            acc[rt][4 * gq + 3] = nl_elu_fast(acc[rt][4 * gq + 3] + b4.w);
          }
          if constexpr (MODE == 2) {
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
              const float u[8] = {acc[rt][8 * sI], acc[rt][8 * sI + 1], acc[rt][8 * sI + 2], acc[rt][8 * sI + 3],
                                  acc[rt][8 * sI + 4], acc[rt][8 * sI + 5], acc[rt][8 * sI + 6], acc[rt][8 * sI + 7]};
              tg_split8<X3>(u, Xh[2 * rt + sI], Xl[2 * rt + sI]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
          }
        }
      }
      if constexpr (kd == CK_FC && Geo::last(c)) {
        // ---- (fc + residual) -> LayerNorm(row) * aggregation scale -> feature_agg, kept as the next products' B operand
        float s1 = 0.f;
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
          for (int r = 0; r < 16; r += 4) s1 += (acc[rt][r] + acc[rt][r + 1]) + (acc[rt][r + 2] + acc[rt][r + 3]);
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 / 256.f;
        float s2 = 0.f;
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float d = acc[rt][r] - mean; s2 += d * d; }
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = 1.f / sqrtf(s2 / 256.f + a.eps);
        const float sc = a.wscale[mm_c];
        const unsigned crow = mok_c ? (unsigned)m_c * 1024u + 16u * hh : 0x80000000u;
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
          float v[16];
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int n = 32 * rt + 8 * gq + 4 * hh;
            const float4 g4 = *(const float4*)(stab + n), be4 = *(const float4*)(stab + 256 + n);
            v[4 * gq + 0] = ((acc[rt][4 * gq + 0] - mean) * rstd * g4.x + be4.x) * sc;
            v[4 * gq + 1] = ((acc[rt][4 * gq + 1] - mean) * rstd * g4.y + be4.y) * sc;
            v[4 * gq + 2] = ((acc[rt][4 * gq + 2] - mean) * rstd * g4.z + be4.z) * sc;
            v[4 * gq + 3] = ((acc[rt][4 * gq + 3] - mean) * rstd * g4.w + be4.w) * sc;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tg_u32x4, tg_f32x4{v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]}), rFA,
                                                   crow + (32 * rt + 8 * gq) * 4, 0, 0);
          }
#pragma unroll
          for (int sI = 0; sI < 2; ++sI) {   // accumulator registers 8 s .. 8 s + 7 of row tile rt = k-step 2 rt + s in accumulator order
            const float u[8] = {v[8 * sI], v[8 * sI + 1], v[8 * sI + 2], v[8 * sI + 3], v[8 * sI + 4], v[8 * sI + 5], v[8 * sI + 6], v[8 * sI + 7]};
            tg_split8<X3>(u, Xh[2 * rt + sI], Xl[2 * rt + sI]);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        }
      }
      if constexpr (kd == CK_F0 && Geo::last(c)) {
        if constexpr (FEAT) {
          const unsigned frow = mok_c ? (unsigned)m_c * 1024u + 16u * hh : 0x80000000u;
#pragma unroll
          for (int rt = 0; rt < 8; ++rt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int n = 32 * rt + 8 * gq + 4 * hh;
              const float4 b4 = *(const float4*)(stab + 512 + n);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tg_u32x4, tg_f32x4{nl_lrelu(acc[rt][4 * gq] + b4.x), nl_lrelu(acc[rt][4 * gq + 1] + b4.y),
                                                                                         nl_lrelu(acc[rt][4 * gq + 2] + b4.z), nl_lrelu(acc[rt][4 * gq + 3] + b4.w)}),
                                                     rFT, frow + (32 * rt + 8 * gq) * 4, 0, 0);
            }
        }
        // MODE 0: row tiles 2..7 are about to receive the next tile's residual rows; MODE 1: the next tile starts from zeros
        if constexpr (MODE == 0) zero_acc(I0{}, I2{}); else zero_acc(I0{}, I8{});
      }
      if constexpr (kd == CK_Q && Geo::last(c)) {
        const unsigned qrow = mok_c ? (unsigned)m_c * 512u + 16u * hh : 0x80000000u;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tg_u32x4, tg_f32x4{acc[rt][4 * gq], acc[rt][4 * gq + 1], acc[rt][4 * gq + 2], acc[rt][4 * gq + 3]}),
                                                   rFA, qrow + (32 * rt + 8 * gq) * 4, 0, 0);
        zero_acc(I0{}, std::integral_constant<int, 4>{});
      }
      if constexpr (kd == CK_BL && Geo::last(c)) {
        const unsigned brow = mok_c ? (unsigned)m_c * 128u + 16u * hh : 0x80000000u;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tg_u32x4, tg_f32x4{acc[0][4 * gq], acc[0][4 * gq + 1], acc[0][4 * gq + 2], acc[0][4 * gq + 3]}), rBL,
                                                 brow + 32 * gq, 0, 0);
        if constexpr (MODE == 0) adopt_gtmp(); else zero_acc(I0{}, I2{});
      }
    });
    tile = tile_next;
    if (tile >= ntiles) break;
  }
  tg_wait_vmcnt<0>();   // LDS-DMA prefetched past the last tile must land before the LDS goes to another workgroup
}

}  // namespace

// stream layout helpers (also used by the packer in abi.hip)
int nl_tgemm_nrt(int N) { return N <= 64 ? 2 : (N <= 128 ? 4 : 8); }
size_t nl_tgemm_stream_bytes(int Kpad, int N) { return (size_t)(Kpad / 32) * 4 * nl_tgemm_nrt(N) * 1024; }

bool nl_tgemm_supported(const NlGemmArgs& a, int precision) {
  if (a.tile_map && (a.epi != NL_EPI_NONE || a.So > 0 || !a.tile_count)) return false;
  if (precision == NL_PREC_F32 || !a.Bst || a.N > 256 || (a.N & 3) || (a.ldc & 3) || (((size_t)a.C) & 15) || a.M <= 0 || !a.zeros) return false;
  if (a.epi == NL_EPI_LNROW && (a.N != 32 * nl_tgemm_nrt(a.N) || a.So > 0 || !a.ep_res || (a.ep_ldres & 3) || (((size_t)a.ep_res) & 15))) return false;
  if (a.epi == NL_EPI_LNSLAB && ((a.So != 128 && a.So != 64 && a.So != 32 && a.So != 16) || a.M % a.So || a.Li != a.So || a.ostride != 1 || a.ooff != 0 || !a.ep_gamma || !a.ep_beta)) return false;
  for (int s = 0; s < a.nseg; ++s) {
    const NlGemmSeg& g = a.seg[s];
    if (!g.vec || (g.k & 31) || g.rdiv > 1 || g.ld < g.k || g.ntap < 1) return false;
  }
  return true;
}

int nl_tgemm_launch(const NlGemmArgs& a, int precision, hipStream_t st) {
  const bool x3 = precision == NL_PREC_BF16X3;
  const int nrt = nl_tgemm_nrt(a.N);
#define NL_TG(NRT, NW, X3)                                                                                  \
  do {                                                                                                       \
    dim3 grid((unsigned)nl_cdiv(a.M, 32 * NW));                                                              \
    if (a.epi == NL_EPI_LNROW)                                                                               \
      hipLaunchKernelGGL((tgemm_kernel<NRT, NW, X3, NL_EPI_LNROW>), grid, dim3(64 * NW), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias); \
    else if (a.epi == NL_EPI_LNSLAB)                                                                         \
      hipLaunchKernelGGL((tgemm_kernel<NRT, NW, X3, NL_EPI_LNSLAB>), grid, dim3(64 * NW), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias); \
    else                                                                                                     \
      hipLaunchKernelGGL((tgemm_kernel<NRT, NW, X3, NL_EPI_NONE>), grid, dim3(64 * NW), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias);  \
  } while (0)
  if (nrt == 8) { if (x3) NL_TG(8, 4, true); else NL_TG(8, 4, false); }
  else if (nrt == 4) { if (x3) NL_TG(4, 4, true); else NL_TG(4, 4, false); }
  else { if (x3) NL_TG(2, 4, true); else NL_TG(2, 4, false); }
#undef NL_TG
  return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}

namespace {
int chain_grid(int ntiles, dim3* grid) {
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NL_ERR_HIP;
    num_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
  }
  *grid = dim3(ntiles < num_cu ? nl_xcd_grid(ntiles) : num_cu);
  return NL_OK;
}
}  // namespace

// T64 != null: G is not read — it is recomputed from the 64-wide out_fc hidden rows (weight stream off_g2, bias bias_g2)
int nl_launch_sample_chain(const float* O, const float* G, const float* wscale, const float* gamma, const float* beta, float eps, const void* wbase,
                           size_t off_fc, size_t off_f0, size_t off_ba, const float* bias_f0, float* FA, float* fth, float* blA, int64_t M, int precision,
                           hipStream_t st, const float* T64, size_t off_g2, const float* bias_g2) {
  if (M <= 0) return NL_OK;
  if ((int64_t)M * 1024 > 0x7fffffffll) return NL_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  const int ntiles = (int)nl_cdiv(M, 128);
  dim3 grid;
  if (chain_grid(ntiles, &grid) != NL_OK) return NL_ERR_HIP;
  NlChainArgs a{O, G, wscale, gamma, beta, eps, (const char*)wbase, (unsigned)off_fc, (unsigned)off_f0, (unsigned)off_ba, bias_f0, FA, fth, blA, (int)M,
                T64, (unsigned)off_g2, 0u, bias_g2, nullptr};
  const bool x3 = precision == NL_PREC_BF16X3;
#define NL_CH(X3, FEAT)                                                                                              \
  do {                                                                                                               \
    if (T64) hipLaunchKernelGGL((sample_chain_kernel<X3, FEAT, 1>), grid, dim3(256), 0, st, a, ntiles);              \
    else hipLaunchKernelGGL((sample_chain_kernel<X3, FEAT, 0>), grid, dim3(256), 0, st, a, ntiles);                  \
  } while (0)
  if (x3 && fth) NL_CH(true, true);
  else if (x3) NL_CH(true, false);
  else if (fth) NL_CH(false, true);
  else NL_CH(false, false);
#undef NL_CH
  return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}

// Q = w_qs(ELU(out_fc.2(t64))): the attention query rows without materialising the multiview feature rows (model.py:391-396)
int nl_launch_query_chain(const float* T64, const void* wbase, size_t off_g2, const float* bias_g2, size_t off_q, float* Q, int64_t M, int precision,
                          hipStream_t st) {
  if (M <= 0) return NL_OK;
  if ((int64_t)M * 1024 > 0x7fffffffll) return NL_ERR_UNSUPPORTED;
  const int ntiles = (int)nl_cdiv(M, 128);
  dim3 grid;
  if (chain_grid(ntiles, &grid) != NL_OK) return NL_ERR_HIP;
  NlChainArgs a{};
  a.wbase = (const char*)wbase; a.M = (int)M; a.T64 = T64; a.off_g2 = (unsigned)off_g2; a.off_q = (unsigned)off_q; a.bias_g2 = bias_g2; a.Q = Q;
  if (precision == NL_PREC_BF16X3) hipLaunchKernelGGL((sample_chain_kernel<true, false, 2>), grid, dim3(256), 0, st, a, ntiles);
  else hipLaunchKernelGGL((sample_chain_kernel<false, false, 2>), grid, dim3(256), 0, st, a, ntiles);
  return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}
