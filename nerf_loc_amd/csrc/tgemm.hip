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
typedef unsigned int tg_u32x4 __attribute__((ext_vector_type(4)));
typedef float tg_f32x4 __attribute__((ext_vector_type(4)));

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
  unsigned h[4], l[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (X3) nl_split_bf16_pair(v[2 * t], v[2 * t + 1], h[t], l[t]);
    else h[t] = nl_bf16_pair(v[2 * t], v[2 * t + 1]);
  }
  hi = __builtin_bit_cast(tg_bf16x8, tg_u32x4{h[0], h[1], h[2], h[3]});
  if (X3) lo = __builtin_bit_cast(tg_bf16x8, tg_u32x4{l[0], l[1], l[2], l[3]});
}

// three-term split-FP16 variant (internal precision NL_PREC_F16X3_INTERNAL: the backward passes' recomputed forward): the same storage type (16-bit
// lanes), fp16 bit patterns; products good to ~2^-22 at the speed of split-bf16
typedef _Float16 tg_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void tg_split8_f16(const float (&v)[8], tg_bf16x8& hi, tg_bf16x8& lo) {
  tg_f16x8 h, l;
#pragma unroll
  for (int t = 0; t < 8; ++t) { const _Float16 x = (_Float16)v[t]; h[t] = x; l[t] = (_Float16)(v[t] - (float)x); }
  hi = __builtin_bit_cast(tg_bf16x8, h);
  lo = __builtin_bit_cast(tg_bf16x8, l);
}
template <bool F16>
__device__ __forceinline__ tg_f32x16 tg_mfma(const tg_bf16x8& a, const tg_bf16x8& b, const tg_f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tg_f16x8, a), __builtin_bit_cast(tg_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// chunk c (k-space [32c, 32c+32)) lies in exactly one segment (every segment is a multiple of 32 wide here);
// kstart[] are the segments' first k (INT_MAX for unused slots) -> branch-free, wave-uniform lookup
__device__ __forceinline__ int tg_find_seg(const NlGemmArgs& a, int k0) {
  int s = 0;
#pragma unroll
  for (int j = 1; j < NL_GEMM_MAX_SEG; ++j) s += (k0 >= a.kstart[j]) ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(s);
}

#ifdef TG_TRACE   // debug build (tools/tg_trace.py): cycle counter of block 0, wave 0 of the NRT = 8 LNSLAB launch: [chunk starts ... | epilogue marks]
__device__ unsigned long long tg_trace[64];
#define TG_T(i)                                                                                      \
  do {                                                                                               \
    if (NRT == 8 && EPI == NL_EPI_LNSLAB && blockIdx.x == 0 && wave == 0 && a.Kpad > 512) {          \
      const unsigned long long t_ = __builtin_readcyclecounter();                                    \
      if (lane == 0) tg_trace[(i)] = t_;                                                             \
    }                                                                                                \
  } while (0)
#else
#define TG_T(i)
#endif

template <int NRT, int NW, bool X3, int EPI, bool F16 = false>
__global__ __launch_bounds__(64 * NW, NW > 4 ? 1 : 2) void tgemm_kernel(const NlGemmArgs a, const char* __restrict__ p_bst, float* __restrict__ p_c,
                                                            const float* __restrict__ p_zeros, const float* __restrict__ p_bias) {
  // split-FP16 arithmetic (the backward passes' recomputed forward): MODE.FP16_OVFL makes the f32 -> f16 conversions SATURATE at 65504 instead of producing inf,
  // so an activation beyond fp16's range costs accuracy, not a NaN gradient (tests/test_backward_kernels.py::test_huge_feature_maps_...)
  if (F16) __builtin_amdgcn_s_setreg(1473, 1);   // hwreg(HW_REG_MODE, 23, 1)
  constexpr int PARTS = X3 ? 2 : 1;
  constexpr int PIECES = PARTS * 2 * NRT;   // 1-KB pieces of weights per chunk
  // waves that stage weights: all of them, or the first four of a six-wave workgroup (S = 192 / 96 slabs: a ray = six / three row tiles; 8 ... 32
  // pieces do not split over six)
  constexpr int NWS = PIECES % NW == 0 ? NW : 4;
  constexpr int NPW = PIECES / NWS;         // pieces staged by each staging wave
  static_assert(PIECES % NWS == 0 && NWS <= NW, "pieces must split evenly over the staging waves");
  constexpr int CH16 = 4 * NRT * 64;        // 16-B units per chunk in the global stream (hi and lo parts are always stored)
  constexpr int SLOT16 = PIECES * 64;       // 16-B units per LDS slot
  __shared__ uint4 lds_all[2 * SLOT16 + NRT * 8 * (EPI == NL_EPI_LNROW ? 3 : EPI == NL_EPI_LNSLAB ? 2 : 1) + (EPI == NL_EPI_LNSLAB ? 8 : 0)];
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
    if (EPI == NL_EPI_LNSLAB) sbias[NRT * 32 + 32 + i] = (a.ep_sig_w && i < a.N) ? a.ep_sig_w[i] : 0.f;   // density-head weights (after the 32 floats of `red`)
  }

  // weights of chunk c: this wave's NPW pieces, 16 B per lane, fully coalesced
  auto w_ptr = [&](int c) __attribute__((always_inline)) { return reinterpret_cast<const tg_bf16x8*>(p_bst) + (size_t)c * CH16 + wave * 64 + lane; };
  const bool stager = NWS == NW || wave < NWS;   // (wave-uniform; a compile-time `true` for the four-wave kernels)
  auto load_w = [&](int c, tg_bf16x8 (&w)[NPW]) __attribute__((always_inline)) {
    if (!stager) return;
    const tg_bf16x8* src = w_ptr(c);
#pragma unroll
    for (int jj = 0; jj < NPW; ++jj) w[jj] = src[NWS * jj * 64];
  };
  auto store_w = [&](int slot, const tg_bf16x8 (&w)[NPW]) __attribute__((always_inline)) {
    if (!stager) return;
#pragma unroll
    for (int jj = 0; jj < NPW; ++jj) ring[slot][(wave + NWS * jj) * 64 + lane] = w[jj];
  };
  // activation fragments of chunk c: 2 k-steps x 8 floats of this lane's source row, straight into registers.  Conv halo
  // rows and rows >= M read a device zero page instead (no masking arithmetic).
  // fr (wave-uniform) = NlGemmSeg::frag of the chunk's segment.  1: a fragment-native split-bf16 image — the four 16-byte pieces of the lane are
  // [k-step 0: hi | lo | k-step 1: hi | lo], 1 KB apart; 0 / 2: fp32 rows, 2 k-steps x 8 floats
  auto act_ptr = [&](int c, int& fr) __attribute__((always_inline)) -> const float* {
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
    fr = F16 ? 0 : sg.frag;
    if (fr == 1) return ok ? sg.ptr + ((size_t)(row >> 5) * (sg.k >> 4) + (kbase >> 4)) * 512 + ((row & 31) + 32 * hh) * 4 : p_zeros;
    // fp32 rows: k-slot 8 hh + t of a k-step is channel 8 hh + t (natural order), or (frag == 2: the layer's weights are packed for the chain kernel's
    // fragments) channel (t & 3) + 8 (t >> 2) + 4 hh of the step's 16 — two 16-byte loads either way
    return (ok ? sg.ptr + (size_t)row * sg.ld + kbase : p_zeros) + (fr == 2 ? 4 : 8) * hh;
  };
  auto act_off = [](int fr, int pc) __attribute__((always_inline)) { return fr == 1 ? 256 * pc : 16 * (pc >> 1) + (fr == 2 ? 8 : 4) * (pc & 1); };
  auto load_act = [&](int c, float4 (&raw)[4], int& fr) __attribute__((always_inline)) {
    const float* p = act_ptr(c, fr);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) raw[pc] = *(const float4*)(p + act_off(fr, pc));
  };
  auto convert = [&](const float4 (&raw)[4], int fr, tg_bf16x8 (&bh)[2], tg_bf16x8 (&bl)[2]) __attribute__((always_inline)) {
    if (fr == 1) {   // (wave-uniform branch around vector moves only: the MFMAs stay in one block)
      bh[0] = __builtin_bit_cast(tg_bf16x8, raw[0]); bl[0] = __builtin_bit_cast(tg_bf16x8, raw[1]);
      bh[1] = __builtin_bit_cast(tg_bf16x8, raw[2]); bl[1] = __builtin_bit_cast(tg_bf16x8, raw[3]);
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float v[8] = {raw[2 * ks].x, raw[2 * ks].y, raw[2 * ks].z, raw[2 * ks].w,
                          raw[2 * ks + 1].x, raw[2 * ks + 1].y, raw[2 * ks + 1].z, raw[2 * ks + 1].w};
      if constexpr (F16) tg_split8_f16(v, bh[ks], bl[ks]); else tg_split8<X3>(v, bh[ks], bl[ks]);
    }
  };

  tg_f32x16 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

  // one chunk: 2 k-steps x NRT row tiles x (3 | 1) MFMAs; A fragments are read two (k-step, tile) pairs ahead
  auto compute = [&](int slot, const tg_bf16x8 (&bh)[2], const tg_bf16x8 (&bl)[2], auto&& filler) __attribute__((always_inline)) {
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
        acc[rt] = tg_mfma<F16>(al[tt % 3], bh[ks], acc[rt]);
        acc[rt] = tg_mfma<F16>(ah[tt % 3], bl[ks], acc[rt]);
      }
      acc[rt] = tg_mfma<F16>(ah[tt % 3], bh[ks], acc[rt]);
      filler(tt);
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
  int raw_fr = 0;   // what `raw` holds (see act_ptr)
  auto clampc = [&](int c) { return c < NC ? c : NC - 1; };
  load_act(0, raw, raw_fr); load_w(0, wreg);
  store_w(0, wreg);
  __syncthreads();
  for (int g = 0; g < NC; ++g) {   // one straight-line body, no control flow around the MFMAs (accumulators stay put)
    TG_T(g < 40 ? g : 40);
    tg_bf16x8 bh[2], bl[2];
    convert(raw, raw_fr, bh, bl);
    // The next chunk's 4 activation loads (lane = row: ~64 cycles each in the CU's address unit) and NPW weight loads go out one at a
    // time between the MFMA groups instead of as a burst in front of them: a burst makes every wave of the workgroup wait at issue
    // (first half of the chunk only: the weights are stored to LDS right after it; chunks of fewer row tiles are too short for this and
    // keep the loads in front)
    if constexpr (NRT == 8) {
      int nfr = 0;
      const float* ap = act_ptr(clampc(g + 1), nfr);
      const tg_bf16x8* wp = w_ptr(clampc(g + 1));
      compute(g & 1, bh, bl, [&](int tt) __attribute__((always_inline)) {
        constexpr int NLD = 4 + NPW, nh = NRT;   // slots 0 .. NRT-1
#pragma unroll
        for (int l = 0; l < NLD; ++l)
          if (l * nh / NLD == tt) {
            if (l < 4) raw[l] = *(const float4*)(ap + act_off(nfr, l));
            else if (stager) wreg[l - 4] = wp[NWS * (l - 4) * 64];
          }
      });
      raw_fr = nfr;
    } else {
      load_act(clampc(g + 1), raw, raw_fr);
      load_w(clampc(g + 1), wreg);
      compute(g & 1, bh, bl, [](int) __attribute__((always_inline)) {});
    }
    store_w((g + 1) & 1, wreg);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31 (= this lane's output row), reg r = 4*gq + e <-> n = 32*rt + 8*gq + 4*hh + e
  if constexpr (EPI == NL_EPI_LNSLAB) {
    // The workgroup's 32*NW rows are one, two, four or eight whole rays (launch precondition: So in {128, 64, 32, 16} with four waves, {192, 96} with six, M % So == 0; `wpr`
    // waves per ray, or two rays per wave for So = 16): LayerNorm over each ray's whole (So x N) slab with per-(position, channel) affine, ELU, optional
    // MaxPool(2) along the ray.  A trailing workgroup may hold rays past M: their statistics are computed on zero rows and
    // nothing of them is stored.
    TG_T(48);
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
    TG_T(49);
    // the affine tables come in accumulator-lane order (abi.hip ln_lane_major_kernel): [wave of the ray][rt][gq][lane][4] — one contiguous
    // KB per load instruction; a row tile's eight loads are issued one row tile ahead of their use
    const float* grow = a.ep_gamma + ((size_t)(wave % wpr) * NRT * 4 * 64 + lane) * 4;
    const float* brow = a.ep_beta + ((size_t)(wave % wpr) * NRT * 4 * 64 + lane) * 4;
    float4 gbuf[2][4], bbuf[2][4];
    auto load_gb = [&](int rt, float4 (&gd)[4], float4 (&bd)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) { gd[gq] = *(const float4*)(grow + (rt * 4 + gq) * 256); bd[gq] = *(const float4*)(brow + (rt * 4 + gq) * 256); }
    };
    load_gb(0, gbuf[0], bbuf[0]);
    const bool pool = a.ep_pool != 0;
    float* orow_p = p_c + (size_t)(pool ? q * (a.So / 2) + (t >> 1) : m) * a.ldc;
    float sg = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      if (rt + 1 < NRT) load_gb(rt + 1, gbuf[(rt + 1) & 1], bbuf[(rt + 1) & 1]);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = 32 * rt + 8 * gq + 4 * hh;
        if (n < a.N) {
          const float4 g4 = gbuf[rt & 1][gq], be4 = bbuf[rt & 1][gq];
          float4 v = nl_ln_elu4(acc[rt][4 * gq + 0], acc[rt][4 * gq + 1], acc[rt][4 * gq + 2], acc[rt][4 * gq + 3], mean, rstd, g4, be4);
          if (pool) {   // positions 2p, 2p+1 are neighbouring lanes
            v.x = nl_max_lane_xor1(v.x); v.y = nl_max_lane_xor1(v.y);   // quad_perm [1,0,3,2]: lane ^ 1
            v.z = nl_max_lane_xor1(v.z); v.w = nl_max_lane_xor1(v.w);
          }
          if (p_c && mok && (!pool || !(j & 1))) *(float4*)(orow_p + n) = v;   // p_c == null: only the density head's output is wanted
          if (a.ep_sig_w) {
            const float4 w4 = *(const float4*)(sbias + NRT * 32 + 32 + n);
            sg = fmaf(v.x, w4.x, sg); sg = fmaf(v.y, w4.y, sg); sg = fmaf(v.z, w4.z, sg); sg = fmaf(v.w, w4.w, sg);
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the gamma/beta loads of later tiles from being hoisted (register budget)
      }
    }
    TG_T(50);
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
  unsigned mb[(NRT + 1) / 2];   // sign bits of this lane's outputs (ep_maskout) / of the forward layer's (ep_maskin)
#pragma unroll
  for (int i = 0; i < (NRT + 1) / 2; ++i) mb[i] = 0u;
  const float* trow = nullptr;   // this row's table row (ep_tab)
  if (a.ep_tab) trow = a.ep_tab + (size_t)((int)((unsigned)m % (unsigned)a.ep_tabK) < a.ep_tabM ? a.ep_tabidx[m] : a.ep_tabM) * a.ep_ldtab;
  if (a.ep_maskin) {
    const uint4 mi = *(const uint4*)(a.ep_maskin + ((size_t)tile * 64 + lane) * 4);
    mb[0] = mi.x;
    if (NRT > 2) mb[1] = mi.y;
    if (NRT > 4) { mb[2] = mi.z; mb[3] = mi.w; }
  }
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = 32 * rt + 8 * gq + 4 * hh;
      const int sh = 16 * (rt & 1) + 4 * gq;
      if (n < a.N) {   // N % 4 == 0 and ldc % 4 == 0 are launch preconditions
        float4 b4 = *(const float4*)(sbias + n);
        if (trow) {   // the table's columns are in accumulator order (point_fused.hip: pack_ptt_kernel): slot 32 rt + 16 hh + r holds feature 32 rt + (r & 3) + 8 (r >> 2) + 4 hh
          const float4 t4 = *(const float4*)(trow + 32 * rt + 16 * hh + 4 * gq);
          b4.x += t4.x; b4.y += t4.y; b4.z += t4.z; b4.w += t4.w;
        }
        float4 v;
        if (a.act == NL_ACT_LRELU_MASK) {   // input gradient through a LeakyReLU: the mask comes from the forward output's sign
          if (a.ep_maskin) {
            const unsigned w = mb[rt >> 1] >> sh;
            v.x = acc[rt][4 * gq + 0] * ((w & 1u) ? 1.f : 0.01f); v.y = acc[rt][4 * gq + 1] * ((w & 2u) ? 1.f : 0.01f);
            v.z = acc[rt][4 * gq + 2] * ((w & 4u) ? 1.f : 0.01f); v.w = acc[rt][4 * gq + 3] * ((w & 8u) ? 1.f : 0.01f);
          } else {
            const float4 h4 = *(const float4*)(a.ep_res + (size_t)m * a.ep_ldres + n);
            v.x = acc[rt][4 * gq + 0] * (h4.x > 0.f ? 1.f : 0.01f); v.y = acc[rt][4 * gq + 1] * (h4.y > 0.f ? 1.f : 0.01f);
            v.z = acc[rt][4 * gq + 2] * (h4.z > 0.f ? 1.f : 0.01f); v.w = acc[rt][4 * gq + 3] * (h4.w > 0.f ? 1.f : 0.01f);
          }
        } else {
          v.x = nl_act(acc[rt][4 * gq + 0] + b4.x, a.act);
          v.y = nl_act(acc[rt][4 * gq + 1] + b4.y, a.act);
          v.z = nl_act(acc[rt][4 * gq + 2] + b4.z, a.act);
          v.w = nl_act(acc[rt][4 * gq + 3] + b4.w, a.act);
          if (a.ep_maskout) mb[rt >> 1] |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << sh;
        }
        *(float4*)(crow + n) = v;
      }
    }
  if (a.ep_maskout) *(uint4*)(a.ep_maskout + ((size_t)tile * 64 + lane) * 4) = make_uint4(mb[0], NRT > 2 ? mb[1] : 0u, NRT > 4 ? mb[2] : 0u, NRT > 4 ? mb[3] : 0u);
}


// ====================================================================================================================
// conv_out in the f16mx arithmetic (round 6; NL_PREC_F16MX, W = 256, S = 128): the product of `tgemm_kernel<8, 4, true, NL_EPI_LNSLAB>` as fp16 hi.hi + two MX-FP6
// cross terms — per K = 64 slab and 32 x 32 tile 4 x v_mfma_f32_32x32x16_f16 + 2 x v_mfma_scale_f32_32x32x64_f8f6f4 (48 matrix passes) instead of 12 bf16
// instructions (96), the arithmetic the fused neural-point kernel has had since round 5.
//   * B operand (activations): built PER SLAB IN REGISTERS from whatever the producer left — the chain kernel's split-bf16 fragments (hi + lo = the value to 16 bits) or
//     fp32 rows: 32 values per lane -> f16 pairs + running maximum, exact residuals, block scale 2^(floor(log2 max) - 2) from the values themselves, ONE
//     v_cvt_scalef32_pk32_fp6_f16 for the hi image and ONE v_cvt_scalef32_2xpk16_fp6_f32 for the residual image (on the hi image's scale x 2^-11).  No producer changes
//     its output format; the conversion (~170 vector instructions per slab) is paid once per 32 rows x 256 columns = 48 matrix instructions.
//   * A operand (weights): the layer's fp16 stream (hi fragments: the `bsh` image nl_pack_weights writes anyway) + fp6 images of f16(w) and w - f16(w) with one E8M0
//     scale per (output row, half-wave, slab) (abi.hip: pack_tgemm_mx6_kernel), both in the natural position order P = 8 s + t (k-step s, element t) that the
//     activation images have.  64 KB per slab through a two-slot LDS ring by LDS-DMA (no staging registers: 128 accumulators + the next slab's 32 raw activation
//     words + the converted operand leave no room for them at two waves per SIMD).
//   * 130 KB of LDS = ONE workgroup per CU, so the workgroup is eight waves = 256 rows = two rays: the same two waves per SIMD as the two four-wave workgroups of
//     the bf16x3 kernel, every weight byte staged once per 256 rows instead of once per 128.
// Epilogue: LayerNorm over each ray's (128 x 256) slab + ELU + the density head, as tgemm_kernel's NL_EPI_LNSLAB.
typedef int tg_i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int tg_u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned int tg_u32x16 __attribute__((ext_vector_type(16)));

// f16 pair of two values + the running maximum of their magnitudes
__device__ __forceinline__ unsigned tg_hi2_f16_amax(float v0, float v1, float& m) {
  unsigned hi;
  asm("v_max3_f32 %1, |%2|, |%3|, %1\n\tv_cvt_pk_f16_f32 %0, %2, %3" : "=&v"(hi), "+v"(m) : "v"(v0), "v"(v1));
  return hi;
}
// residuals of a pair as floats: v - float(hi half) (exact)
__device__ __forceinline__ void tg_lo2_f32(float v0, float v1, unsigned hi, float& l0, float& l1) {
  asm("v_fma_mix_f32 %0, %4, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %4, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l0), "=&v"(l1) : "v"(v0), "v"(v1), "v"(hi));
}
// split-FP16 of 8 values on the packed conversions (hi = f16(v), lo = f16(v - hi); 16 instructions): the chain kernel's F16FRAG rows
__device__ __forceinline__ void tg_split8_f16_pk(const float (&v)[8], tg_bf16x8& hi, tg_bf16x8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[t]) : "v"(v[2 * t]), "v"(v[2 * t + 1]));
    float l0, l1;
    tg_lo2_f32(v[2 * t], v[2 * t + 1], h[t], l0, l1);
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l[t]) : "v"(l0), "v"(l1));
  }
  hi = __builtin_bit_cast(tg_bf16x8, tg_u32x4{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(tg_bf16x8, tg_u32x4{l[0], l[1], l[2], l[3]});
}
// The f16mx operand of a 64-k slab from split-FP16 fragment words (NlGemmSeg::frag == 3): the hi plane IS the f16 operand, the block maximum is an integer maximum of
// its magnitudes, the lo plane becomes the residual fp6 image with the same instruction that makes the hi image (on the hi image's scale x 2^-11).  raw[ci][2 ks] /
// raw[ci][2 ks + 1] = hi / lo words of chunk ci (0, 1), k-step ks: positions P = 16 ci + 8 ks + 0 .. 7.  Returns the block exponent byte eb (scale 2^(eb - 127)).
typedef unsigned short tg_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tg_u32x6 tg_cvt_pk32_fp6_f16(tg_u32x16 h, float sc);
__device__ __forceinline__ int tg_mx_operand_f16frag(const float4 (&raw)[2][4], tg_u32x16& H, tg_u32x6& xh6, tg_u32x6& xl6) {
  tg_u32x16 Lw;
  tg_u16x2 mx = {0, 0};
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float4 fh = raw[ci][2 * ks], fl = raw[ci][2 * ks + 1];
      const unsigned uh[4] = {__float_as_uint(fh.x), __float_as_uint(fh.y), __float_as_uint(fh.z), __float_as_uint(fh.w)};
      const unsigned ul[4] = {__float_as_uint(fl.x), __float_as_uint(fl.y), __float_as_uint(fl.z), __float_as_uint(fl.w)};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        H[8 * ci + 4 * ks + d] = uh[d];
        Lw[8 * ci + 4 * ks + d] = ul[d];
        mx = __builtin_elementwise_max(mx, __builtin_bit_cast(tg_u16x2, uh[d] & 0x7fff7fffu));   // (f16 magnitudes order like their bit patterns)
      }
    }
  const unsigned short m16 = mx[0] > mx[1] ? mx[0] : mx[1];
  const float amax = (float)__builtin_bit_cast(_Float16, m16);
  int eb = __builtin_amdgcn_frexp_expf(amax) + 124;   // as tgemm_mx_kernel: the largest value lands in [4, 8)
  eb = eb < 12 ? 12 : (eb > 254 ? 254 : eb);
  const float scf = __builtin_bit_cast(float, eb << 23);
  xh6 = tg_cvt_pk32_fp6_f16(H, scf);
  xl6 = tg_cvt_pk32_fp6_f16(Lw, scf * 0.00048828125f);
  return eb;
}
// (asm with early-clobber results: hipcc 7.2 lets the builtins' 6-register result overlap the scale operand — point_fused2.hip)
__device__ __forceinline__ tg_u32x6 tg_cvt_pk32_fp6_f16(tg_u32x16 h, float sc) {
  tg_u32x6 r;
  asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(r) : "v"(h), "v"(sc));
  return r;
}
__device__ __forceinline__ tg_u32x6 tg_cvt_2xpk16_fp6_f32(tg_f32x16 a, tg_f32x16 b, float sc) {
  tg_u32x6 r;
  asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(sc));
  return r;
}

constexpr int TGMX_NRT = 8, TGMX_NW = 8;
constexpr int TGMX_F16B = 4 * TGMX_NRT * 1024;                 // f16 hi fragments of a slab: [chunk of the slab][k-step][row tile][lane] x 16 B
constexpr int TGMX_IMA = 2 * TGMX_NRT * 1024, TGMX_IMB = TGMX_IMA;   // fp6 images of [row tile][w_hi6, w_lo6][lane]: dwords 0-3 | {dword 4, dword 5, E8M0 scale byte, 0} — 16-byte reads only
                                                                     // (every LDS read of the loop has ONE native vector type: a scalar-typed read made the waitcnt pass drain the DMA in front of it)
constexpr int TGMX_IMG = TGMX_IMA + TGMX_IMB;                  // bytes per slab in the image stream (32 KB)
constexpr int TGMX_SLOT = TGMX_F16B + TGMX_IMG;                // 64 KB: eight 1-KB pieces per wave
constexpr int TGMX_PIECES = TGMX_SLOT / 1024;

__global__ __launch_bounds__(64 * TGMX_NW, 1) void tgemm_mx_kernel(const NlGemmArgs a, const char* __restrict__ p_bsh, const char* __restrict__ p_bmx, float* __restrict__ p_c,
                                                                    const float* __restrict__ p_zeros, const float* __restrict__ p_bias) {
  constexpr int NRT = TGMX_NRT, NW = TGMX_NW;
  __builtin_amdgcn_s_setreg(1473, 1);   // hwreg(HW_REG_MODE, 23, 1): MODE.FP16_OVFL — f32 -> f16 / fp6 conversions saturate instead of producing inf / NaN
  // (The two waves of a SIMD — wave w and w + 4: the two rays — run the same program between the same barriers: both convert, then both multiply.  A phase-shifted
  // program, one group converting while the other multiplies, needs a wave-uniform branch around the matrix instructions, and hipcc then spills the accumulators:
  // 2 768 spilled registers; a static issue priority for the first ray's waves, s_setprio 3, measured no change: 504 / 519 against 508 / 523 us.)
  __shared__ uint4 lds_all[2 * TGMX_SLOT / 16 + NRT * 8 * 2 + 8];
  float* sbias = reinterpret_cast<float*>(lds_all + 2 * TGMX_SLOT / 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31;
  const int tile = blockIdx.x * NW + wave;
  const int m = tile * 32 + j;
  const bool mok = m < a.M;
  const int q = m / a.So, t = m - q * a.So;
  const int NC = a.Kpad >> 5;

  for (int i = tid; i < NRT * 32; i += 64 * NW) {
    sbias[i] = (p_bias && i < a.N) ? p_bias[i] : 0.f;
    sbias[NRT * 32 + 32 + i] = (a.ep_sig_w && i < a.N) ? a.ep_sig_w[i] : 0.f;   // density-head weights (after the 32 floats of `red`)
  }

  // ---- weights of slab s -> LDS slot: 1-KB pieces by BUFFER LDS-DMA (counted like any load: the compiler's waits stay exact; a flat global_load_lds turns every
  // later wait into vmcnt(0)), each lane 16 bytes, pieces dealt round-robin to the waves.  The slot is a COMPILE-TIME constant here and in the reads below: with a
  // run-time slot the waitcnt pass cannot tell the DMA's destination from the slot being read and drains vmcnt(0) in front of every LDS read.
  const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)p_bsh, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmx = __builtin_amdgcn_make_buffer_rsrc((void*)p_bmx, 0, 0x7fffffff, 0x00020000);
  const unsigned lane16 = lane * 16;
  // (stage_piece: piece i of this wave; stage: all of them.  Since late round 6 the pieces and the raw activation loads of the next slab go out BETWEEN the matrix
  // instructions of the current one, one memory instruction every third unit: issued in one go behind the slab's barrier they cost the wave 1.1-3.5 k cycles of a
  // 6.3-7.8 k-cycle slab before its first matrix instruction — a CU accepts one 1-KB wave load per ~17 cycles and all eight waves arrive at once; tools/tgmx_trace.py)
  constexpr int TGMX_PPW = (TGMX_PIECES + NW - 1) / NW;
  auto stage_piece = [&](int sl, auto SLOTc, auto Ic) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOTc)::value;
    const int cA = 2 * sl, cB = (2 * sl + 1 < NC) ? 2 * sl + 1 : NC - 1;   // (an odd chunk count: the last slab's second half re-reads the last chunk — its activations are zero)
    const unsigned oA = (unsigned)cA * (4 * NRT * 1024), oB = (unsigned)cB * (4 * NRT * 1024);   // the hi part (first 2 NRT KB) of each chunk of the fp16 stream
    const unsigned oI = (unsigned)sl * TGMX_IMG;
    {
      constexpr int i = decltype(Ic)::value;
      const int pp = wave + NW * i;   // (wave-uniform; TGMX_PIECES = 8 NW: no tail)
      static_assert(TGMX_PIECES % TGMX_NW == 0, "pieces per wave");
      {
        auto* dst = (__attribute__((address_space(3))) void*)(lds_all + (slot * TGMX_SLOT) / 16 + pp * 64);
        // (which stream a piece comes from is a function of i alone — pp = wave + 8 i: i < 2 chunk A's hi part, i < 4 chunk B's, else the fp6 images — so no
        // wave-uniform branch sits between the matrix instructions this is issued among)
        static_assert(TGMX_NW == 8 && TGMX_NRT == 8, "piece -> stream by i");
        if constexpr (i < 4) {
          unsigned so = i < 2 ? oA + pp * 1024 : oB + (pp - 2 * NRT) * 1024;
          asm volatile("" : "+s"(so));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsh, dst, 16, lane16, so, 0, 0);
        } else {
          unsigned so = oI + (pp - 4 * NRT) * 1024;
          asm volatile("" : "+s"(so));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rmx, dst, 16, lane16, so, 0, 0);
        }
      }
    }
  };
  auto stage = [&](int sl, auto SLOTc) __attribute__((always_inline)) {
    tg_static_for<TGMX_PPW>([&](auto Ic) __attribute__((always_inline)) { stage_piece(sl, SLOTc, Ic); });
  };
  // ---- activations.  The layer's K structure is conv_out's (W = 256; checked by nl_tgemm_mx_supported): chunks 0 .. 23 = feature_agg (eight 32-channel blocks x three
  // taps, [block][tap]), 24 .. 26 = the three taps of x2's one block, 27 = nothing.  Everything that depends on the chunk is therefore a compile-time constant, and what
  // depends on the lane — the row of each of the three taps, whether it exists (a ray's first / last position), the lane's slice of it — is worked out ONCE here:
  // tgemm_kernel's act_ptr costs ~150 scalar instructions and several kernarg loads per chunk (segment look-up, a division by the tap count), eight waves x 28 chunks of
  // them through the CU's one scalar unit.
  const NlGemmSeg& g0 = a.seg[0];
  const NlGemmSeg& g1 = a.seg[1];
  const int frg = __builtin_amdgcn_readfirstlane(g0.frag);        // 1: the chain kernel's fragment image (3: its split-FP16 form), 2 / 0: fp32 rows (2: the same K order inside a block as the image)
  const bool f16in = frg == 3;                                     // split-FP16 fragments: the hi plane is the f16 operand as it stands (tg_mx_operand_f16frag)
  const int fr0 = f16in ? 1 : frg;                                 // (addressing: the two fragment forms have one layout)
  const int cstride0 = fr0 == 1 ? 1024 : 32;                       // floats between consecutive 32-channel blocks of a row
  const float* P0[3]; const float* P1[3]; int okm[3];
#pragma unroll
  for (int tp = 0; tp < 3; ++tp) {
    const int i = t + tp - 1;
    const bool ok = mok && i >= 0 && i < a.Li;
    const int row = q * a.Li + i;
    okm[tp] = ok ? 1 : 0;
    const float* f = fr0 == 1 ? g0.ptr + ((size_t)(row >> 5) * (g0.k >> 4)) * 512 + ((row & 31) + 32 * hh) * 4 : g0.ptr + (size_t)row * g0.ld + (fr0 == 2 ? 4 : 8) * hh;
    P0[tp] = ok ? f : p_zeros;
    P1[tp] = ok ? g1.ptr + (size_t)row * g1.ld + 8 * hh : p_zeros;
  }
  auto act_off = [](int fr, int pc) __attribute__((always_inline)) { return fr == 1 ? 256 * pc : 16 * (pc >> 1) + (fr == 2 ? 8 : 4) * (pc & 1); };
  float4 raw[2][4];
  int rfr[2] = {0, 0};
  auto load_chunk = [&](auto Cc, auto CIc, auto PCc_) __attribute__((always_inline)) {   // PCc: the one 16-byte piece to load (-1: all four)
    constexpr int c = decltype(Cc)::value, ci = decltype(CIc)::value, PCc = decltype(PCc_)::value;
    const float* p; int fr;
#ifndef TGMX_KO
#define TGMX_KO 0   // timing experiment (results wrong): 1 = every tap of feature_agg reads the centre row
#endif
    if constexpr (c < 24) { constexpr int tp = TGMX_KO ? 1 : c % 3, cb = c / 3; p = P0[tp] + okm[tp] * (cb * cstride0); fr = fr0; }
    else if constexpr (c < 27) { p = P1[c - 24]; fr = 0; }
    else { p = p_zeros; fr = 0; }
    rfr[ci] = fr;
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) if (PCc < 0 || pc == PCc) raw[ci][pc] = *(const float4*)(p + act_off(fr, pc));
  };
  auto load_act = [&](auto SLc) __attribute__((always_inline)) {   // the raw words of slab SL (past the end: zero rows)
    constexpr int sl = decltype(SLc)::value;
    load_chunk(std::integral_constant<int, (2 * sl < 28 ? 2 * sl : 27)>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, -1>{});
    load_chunk(std::integral_constant<int, (2 * sl + 1 < 28 ? 2 * sl + 1 : 27)>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, -1>{});
  };
  auto load_act_piece = [&](auto SLc, auto Kc) __attribute__((always_inline)) {   // memory instruction k (0 .. 7) of slab SL's raw words: chunk k / 4, piece k % 4
    constexpr int sl = decltype(SLc)::value, k = decltype(Kc)::value, ci = k >> 2;
    load_chunk(std::integral_constant<int, (2 * sl + ci < 28 ? 2 * sl + ci : 27)>{}, std::integral_constant<int, ci>{}, std::integral_constant<int, (k & 3)>{});
  };

  tg_f32x16 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

  stage(0, std::integral_constant<int, 0>{});
  load_act(std::integral_constant<int, 0>{});
  tg_wait_vmcnt<0>();
  __syncthreads();

  constexpr int NSC = 14;   // slabs of the layer (27 chunks)
#if defined(TG_TRACE) && !defined(TG_TRACE_C1)   // (tools/tgmx_trace.py: phases of slabs 4 .. 9 of block 0, waves 0 and 4 — the two waves of SIMD 0)
#define TGMX_T(k) do { if (blockIdx.x == 0 && (wave == 0 || wave == 4) && g >= 4 && g < 10) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) tg_trace[(wave ? 32 : 0) + 5 * (g - 4) + (k)] = t_; } } while (0)
#else
#define TGMX_T(k)
#endif
  auto slab = [&](auto Gc) __attribute__((always_inline)) {
    constexpr int g = decltype(Gc)::value, SL = g & 1;
    TGMX_T(0);
    // ---- this slab's B operand from the raw words: 32 values (position P = 8 s + t, k-step s = 2 (chunk of the slab) + ks)
    tg_u32x16 H;
    tg_u32x6 xh6, xl6;
    int eb;
    if (g < 12 && f16in) eb = tg_mx_operand_f16frag(raw, H, xh6, xl6);   // feature_agg's slabs from split-FP16 fragments: no sums, no conversions to f16, no residuals
    else {
    float v[32];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      if (rfr[ci] == 1) {   // split-bf16 fragments [ks 0: hi | lo | ks 1: hi | lo]: value = hi + lo (wave-uniform branch around vector moves only)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const float4 fh = raw[ci][2 * ks], fl = raw[ci][2 * ks + 1];
          const unsigned uh[4] = {__float_as_uint(fh.x), __float_as_uint(fh.y), __float_as_uint(fh.z), __float_as_uint(fh.w)};
          const unsigned ul[4] = {__float_as_uint(fl.x), __float_as_uint(fl.y), __float_as_uint(fl.z), __float_as_uint(fl.w)};
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            v[16 * ci + 8 * ks + 2 * d] = __uint_as_float(uh[d] << 16) + __uint_as_float(ul[d] << 16);
            v[16 * ci + 8 * ks + 2 * d + 1] = __uint_as_float(uh[d] & 0xffff0000u) + __uint_as_float(ul[d] & 0xffff0000u);
          }
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const float4 f0 = raw[ci][2 * ks], f1 = raw[ci][2 * ks + 1];
          v[16 * ci + 8 * ks + 0] = f0.x; v[16 * ci + 8 * ks + 1] = f0.y; v[16 * ci + 8 * ks + 2] = f0.z; v[16 * ci + 8 * ks + 3] = f0.w;
          v[16 * ci + 8 * ks + 4] = f1.x; v[16 * ci + 8 * ks + 5] = f1.y; v[16 * ci + 8 * ks + 6] = f1.z; v[16 * ci + 8 * ks + 7] = f1.w;
        }
        if (rfr[ci] == 2) {   // the same rows the chain kernel would have handed over as fragments: take the value its split carries (bf16 hi + bf16 lo), so that a batch
                              // renders to the same bits whichever source format its chunk took (early termination, a sample count that is not a multiple of 32)
#pragma unroll
          for (int i = 0; i < 16; i += 2) {   // (pairs: nl_split_bf16_pair's instruction sequence, the two words put back together)
            unsigned h, l;
            nl_split_bf16_pair(v[16 * ci + i], v[16 * ci + i + 1], h, l);
            const nl_f32x2 r = nl_f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)} + nl_f32x2{__uint_as_float(l << 16), __uint_as_float(l & 0xffff0000u)};
            v[16 * ci + i] = r[0]; v[16 * ci + i + 1] = r[1];
          }
        }
      }
    }
    float amax = 0.f;
    float lo[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      H[i] = tg_hi2_f16_amax(v[2 * i], v[2 * i + 1], amax);
      tg_lo2_f32(v[2 * i], v[2 * i + 1], H[i], lo[2 * i], lo[2 * i + 1]);
    }
    eb = __builtin_amdgcn_frexp_expf(amax) + 124;   // block scale 2^(ex - 3) for amax = m 2^ex, m in [0.5, 1): the largest value lands in [4, 8)
    eb = eb < 12 ? 12 : (eb > 254 ? 254 : eb);          // (12: the residual image's byte eb - 11 stays positive; an all-zero block takes any scale)
    const float scf = __builtin_bit_cast(float, eb << 23);
    xh6 = tg_cvt_pk32_fp6_f16(H, scf);
    const tg_f32x16 le = {lo[0], lo[2], lo[4], lo[6], lo[8], lo[10], lo[12], lo[14], lo[16], lo[18], lo[20], lo[22], lo[24], lo[26], lo[28], lo[30]};
    const tg_f32x16 lod = {lo[1], lo[3], lo[5], lo[7], lo[9], lo[11], lo[13], lo[15], lo[17], lo[19], lo[21], lo[23], lo[25], lo[27], lo[29], lo[31]};
    xl6 = tg_cvt_2xpk16_fp6_f32(le, lod, scf * 0.00048828125f);   // (interleaves its operands: position 2 i <- le[i], 2 i + 1 <- lod[i] = the natural order)
    }
    const tg_i32x8 bh6 = {(int)xh6[0], (int)xh6[1], (int)xh6[2], (int)xh6[3], (int)xh6[4], (int)xh6[5], 0, 0};
    const tg_i32x8 bl6 = {(int)xl6[0], (int)xl6[1], (int)xl6[2], (int)xl6[3], (int)xl6[4], (int)xl6[5], 0, 0};
    const int sxh = eb, sxl = eb - 11;
    TGMX_T(1);

    // ---- the next slab: weights by LDS-DMA into the other slot (every wave left it before the barrier that ended the previous iteration), its raw activation words
    // (issued between the matrix instructions below: mem_slot)
    auto mem_slot = [&](auto Mc) __attribute__((always_inline)) {   // memory instruction m (0 .. 15) of the next slab: 8 weight pieces, then the 8 raw-word loads
      constexpr int m = decltype(Mc)::value;
      if constexpr (m < TGMX_PPW) stage_piece(g + 1 < NSC ? g + 1 : NSC - 1, std::integral_constant<int, 1 - SL>{}, Mc);   // (past the end: the last slab again, never used)
      else if constexpr (m < TGMX_PPW + 8) load_act_piece(std::integral_constant<int, g + 1>{}, std::integral_constant<int, m - TGMX_PPW>{});   // (past the end: zero rows)
    };

    // ---- the slab's product: per row tile 4 f16 k-steps + the two cross terms = 48 units; the A operand of unit u + 2 is read from LDS before the matrix instruction of
    // unit u is issued (three rotating register sets: without the read-ahead every matrix instruction waits out an LDS round trip, ~170 cycles for a 32-cycle instruction)
    const tg_u32x4* Lf = reinterpret_cast<const tg_u32x4*>(lds_all + (SL * TGMX_SLOT) / 16);
    const tg_u32x4* La = reinterpret_cast<const tg_u32x4*>(lds_all + (SL * TGMX_SLOT + TGMX_F16B) / 16);
    const tg_u32x4* Lb = reinterpret_cast<const tg_u32x4*>(lds_all + (SL * TGMX_SLOT + TGMX_F16B + TGMX_IMA) / 16);
#ifndef TGMX_RD
#define TGMX_RD 4   // measured 2 / 3 / 4 units ahead: 481-483 / 475 / 472-473 us (241 / 245 registers at 3 / 4, no scratch)
#endif
    constexpr int RD = TGMX_RD, RR = RD + 1;   // weight fragments read RD units ahead (RR rotating register sets)
    TGMX_T(2);
    tg_u32x4 ra[RR], rb[RR];
    auto rdA = [&](auto Uc) __attribute__((always_inline)) {
      constexpr int u = decltype(Uc)::value, rt = u / 6, k = u % 6, r = u % RR;
      if constexpr (k < 4) ra[r] = Lf[(k * NRT + rt) * 64 + lane];
      else { ra[r] = La[(rt * 2 + (k - 4)) * 64 + lane]; rb[r] = Lb[(rt * 2 + (k - 4)) * 64 + lane]; }   // rb: {dword 4, dword 5, scale byte, 0}
    };
    tg_static_for<RD>([&](auto Uc) __attribute__((always_inline)) { rdA(Uc); });
    tg_static_for<6 * NRT>([&](auto Uc) __attribute__((always_inline)) {
      constexpr int u = decltype(Uc)::value, rt = u / 6, k = u % 6, r = u % RR;
      if constexpr (u + RD < 6 * NRT) rdA(std::integral_constant<int, u + RD>{});
      if constexpr (u % 3 == 0) mem_slot(std::integral_constant<int, u / 3>{});   // 16 slots in 48 units
      if constexpr (k < 4) {
        const tg_f16x8 bf = __builtin_bit_cast(tg_f16x8, (tg_u32x4){H[4 * k], H[4 * k + 1], H[4 * k + 2], H[4 * k + 3]});
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tg_f16x8, ra[r]), bf, acc[rt], 0, 0, 0);
      } else {
        const tg_i32x8 w6 = {(int)ra[r][0], (int)ra[r][1], (int)ra[r][2], (int)ra[r][3], (int)rb[r][0], (int)rb[r][1], 0, 0};
        if constexpr (k == 4) acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w6, bl6, acc[rt], 2, 2, 0, (int)rb[r][2], 0, sxl);   // w_hi6 x a_lo6
        else acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w6, bh6, acc[rt], 2, 2, 0, (int)rb[r][2], 0, sxh);                     // w_lo6 x a_hi6
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    TGMX_T(3);
    tg_wait_vmcnt<0>();
    TGMX_T(4);
    __syncthreads();
  };
  tg_static_for<NSC>([&](auto Gc) __attribute__((always_inline)) { slab(Gc); });

  // ---- epilogue: LayerNorm over each ray's whole (So x N) slab (So = 128: four waves per ray, two rays per workgroup), ELU, the density head (tgemm_kernel's NL_EPI_LNSLAB)
  float* red = sbias + NRT * 32;   // [2][NW] partial sums
  const int wpr = a.So >> 5, gb = (wave / wpr) * wpr;
  float s1 = 0.f;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = 32 * rt + 8 * gq + 4 * hh;
      const float4 b4 = *(const float4*)(sbias + n);
      nl_f32x2 p0, p1;
      nl_bias_sum4(acc[rt][4 * gq + 0], acc[rt][4 * gq + 1], acc[rt][4 * gq + 2], acc[rt][4 * gq + 3], b4, p0, p1, s1);
      acc[rt][4 * gq + 0] = p0[0]; acc[rt][4 * gq + 1] = p0[1]; acc[rt][4 * gq + 2] = p1[0]; acc[rt][4 * gq + 3] = p1[1];
    }
  s1 = wave_sum(s1);
  if (lane == 0) red[wave] = s1;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) tot += red[gb + w];
  const float cnt = (float)a.So * (float)a.N;
  const float mean = tot / cnt;
  float s2;
  {
    nl_f32x2 sp = {0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) nl_sumsq_dev2(sp, acc[rt][r], acc[rt][r + 1], mean);
    s2 = sp[0] + sp[1];
  }
  s2 = wave_sum(s2);
  if (lane == 0) red[NW + wave] = s2;
  __syncthreads();
  float tot2 = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) tot2 += red[NW + gb + w];
  const float rstd = 1.f / sqrtf(tot2 / cnt + a.ep_eps);
  const float* grow = a.ep_gamma + ((size_t)(wave % wpr) * NRT * 4 * 64 + lane) * 4;
  const float* brow = a.ep_beta + ((size_t)(wave % wpr) * NRT * 4 * 64 + lane) * 4;
  float4 gbuf[2][4], bbuf[2][4];
  auto load_gb = [&](int rt, float4 (&gd)[4], float4 (&bd)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) { gd[gq] = *(const float4*)(grow + (rt * 4 + gq) * 256); bd[gq] = *(const float4*)(brow + (rt * 4 + gq) * 256); }
  };
  load_gb(0, gbuf[0], bbuf[0]);
  float* orow_p = p_c + (size_t)m * a.ldc;
  float sg = 0.f;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    if (rt + 1 < NRT) load_gb(rt + 1, gbuf[(rt + 1) & 1], bbuf[(rt + 1) & 1]);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = 32 * rt + 8 * gq + 4 * hh;
      const float4 g4 = gbuf[rt & 1][gq], be4 = bbuf[rt & 1][gq];
      const float4 vv = nl_ln_elu4(acc[rt][4 * gq + 0], acc[rt][4 * gq + 1], acc[rt][4 * gq + 2], acc[rt][4 * gq + 3], mean, rstd, g4, be4);
      if (p_c && mok) *(float4*)(orow_p + n) = vv;   // p_c == null: only the density head's output is wanted
      if (a.ep_sig_w) {
        const float4 w4 = *(const float4*)(sbias + NRT * 32 + 32 + n);
        sg = fmaf(vv.x, w4.x, sg); sg = fmaf(vv.y, w4.y, sg); sg = fmaf(vv.z, w4.z, sg); sg = fmaf(vv.w, w4.w, sg);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (a.ep_sig_w) {
    sg += __shfl_xor(sg, 32, 64);
    if (hh == 0 && mok) a.ep_sig_out[m] = nl_softplus(sg + a.ep_sig_b[0]);
  }
}


// ====================================================================================================================
// feat_mlp.0 + LeakyReLU + the compositing of its rows along the ray as ONE kernel in the f16mx arithmetic (round 6; NL_PREC_F16MX, W = 256):
//   hc[ray][c] = sum_s w_s LeakyReLU(feat_mlp.0 . feature_agg_s + b)[c]          (model.py:594-597 composites feat_mlp's output; its last Linear is applied afterwards:
// abi.hip do_heads).  Before, the chain kernel ran feat_mlp.0 as 8 of its 22 chunks (three-term split-bf16: 384 of a tile's 720 matrix instructions), wrote the hidden rows
// (N x 256 fp32 = 0.54 GB at config 2) and composite_kernel read them back to weight and sum them.  Here the product runs AFTER the density is known, with the weights of
// the samples at hand: the hidden rows are never written, feature_agg's fragment image (which conv1 and conv_out read anyway) is read once more.
// The main loop is tgemm_mx_kernel's with K = 256 (four 64-k slabs, chunks = the eight 32-channel blocks of the fragment image) and the two MFMA operands SWAPPED:
// activations = A, weights = B, so the 32x32 tile comes out as D[sample][channel] — lane = channel, registers = the wave's 32 samples — and the sum over samples is 16
// in-lane multiply-adds + one exchange with lane ^ 32 (in the transposed orientation of every other kernel here the samples live in lanes: 5 cross-lane steps for each
// of 128 registers).  Same operand registers, same LDS images, same weight streams (G_FEAT0P's fp16 stream + its fp6 images: pack_tgemm_mx6_kernel).
// A group = NW waves = NW x 32 consecutive samples = whole rays (S / 32 waves each); the waves of a ray add their partial sums in wave order through LDS.
// Persistent: one workgroup per CU walks the groups blockIdx.x, + gridDim.x, ...; the last slab of a group stages the first slab of the next (four slabs: the ring's
// slot parity carries over), so the exposed load latency of a group's first slab — a fifth of a group's time with 130 KB of LDS = one workgroup per CU — is paid once.
template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void feat_comp_mx_kernel(const float* __restrict__ fa, const float* __restrict__ wts, const int M, const int S,
                                                                   const char* __restrict__ p_bsh, const char* __restrict__ p_bmx, const float* __restrict__ p_bias,
                                                                   float* __restrict__ hc, const int ngroups, const float* __restrict__ w2, const int npad, const int C,
                                                                   const float* __restrict__ wsum, float* __restrict__ feat, const int f16frag) {
  constexpr int NRT = TGMX_NRT;
  constexpr int HB = 16;   // composited rows kept for feat_mlp.2 (w2 != null): 16 rays x 256 floats, channel-major
  const bool f16in = __builtin_amdgcn_readfirstlane(f16frag) != 0;   // the fragment image is split-FP16 (sample_chain_kernel: F16FRAG)
  __builtin_amdgcn_s_setreg(1473, 1);   // MODE.FP16_OVFL: saturating f32 -> f16 / fp6 conversions (tgemm_mx_kernel)
  __shared__ uint4 lds_all[2 * TGMX_SLOT / 16];
  __shared__ float4 hcsT[256 * HB / 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31;
  const int ntiles = M >> 5;
  auto tile_ptr = [&](int grp) __attribute__((always_inline)) {   // the fragment rows of this wave's tile of group grp (past the end: the last tile, with zero weights below)
    const int tl = grp * NW + wave;
    return fa + (size_t)(tl < ntiles ? tl : ntiles - 1) * 8192 + (j + 32 * hh) * 4;
  };

  const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)p_bsh, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmx = __builtin_amdgcn_make_buffer_rsrc((void*)p_bmx, 0, 0x7fffffff, 0x00020000);
  const unsigned lane16 = lane * 16;
  // weights of slab sl -> LDS slot (compile-time): 1-KB pieces by buffer LDS-DMA, dealt round-robin; stage_piece: piece i of this wave
  constexpr int PPW = (TGMX_PIECES + NW - 1) / NW;
  auto stage_piece = [&](int sl, auto SLOTc, auto Ic) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOTc)::value, i = decltype(Ic)::value;
    const unsigned oA = (unsigned)(2 * sl) * (4 * NRT * 1024), oB = (unsigned)(2 * sl + 1) * (4 * NRT * 1024);   // the hi part (first 2 NRT KB) of each chunk of the fp16 stream
    const unsigned oI = (unsigned)sl * TGMX_IMG;
    const int pp = wave + NW * i;   // (wave-uniform)
    auto* dst = (__attribute__((address_space(3))) void*)(lds_all + (slot * TGMX_SLOT) / 16 + pp * 64);
    if constexpr (NW == 8) {   // the stream of a piece is a function of i alone (pp = wave + 8 i): no wave-uniform branch
      if constexpr (i < 4) {
        unsigned so = i < 2 ? oA + pp * 1024 : oB + (pp - 2 * NRT) * 1024;
        asm volatile("" : "+s"(so));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsh, dst, 16, lane16, so, 0, 0);
      } else {
        unsigned so = oI + (pp - 4 * NRT) * 1024;
        asm volatile("" : "+s"(so));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rmx, dst, 16, lane16, so, 0, 0);
      }
    } else if (TGMX_PIECES % NW == 0 || pp < TGMX_PIECES) {
      if (pp < 4 * NRT) {
        unsigned so = pp < 2 * NRT ? oA + pp * 1024 : oB + (pp - 2 * NRT) * 1024;
        asm volatile("" : "+s"(so));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsh, dst, 16, lane16, so, 0, 0);
      } else {
        unsigned so = oI + (pp - 4 * NRT) * 1024;
        asm volatile("" : "+s"(so));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rmx, dst, 16, lane16, so, 0, 0);
      }
    }
  };
  auto stage = [&](int sl, auto SLOTc) __attribute__((always_inline)) {
    tg_static_for<PPW>([&](auto Ic) __attribute__((always_inline)) { stage_piece(sl, SLOTc, Ic); });
  };
  // activations: the chain kernel's fragment image — per 32-row tile 16 k-steps x 512 floats, a 32-channel block = [ks 0: hi | lo | ks 1: hi | lo] x (64 lanes x 4 floats)
  const float* P = tile_ptr(blockIdx.x);
  float4 raw[2][4];
  auto load_act = [&](const float* Pt, int sl) __attribute__((always_inline)) {
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) raw[ci][pc] = *(const float4*)(Pt + (2 * sl + ci) * 1024 + 256 * pc);
  };

  stage(0, std::integral_constant<int, 0>{});
  load_act(P, 0);
  tg_wait_vmcnt<0>();
  __syncthreads();

  constexpr int NSC = 4;   // slabs of the layer (K = 256)
  static_assert(NSC % 2 == 0, "the ring's slot parity must carry over from one group to the next");
  int nbuf = 0, grp0 = 0;   // rays collected for feat_mlp.2 since the last flush, the group of the first of them
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
  const int tile = grp * NW + wave;
  const bool live = tile < ntiles;
  const int tile_c = live ? tile : ntiles - 1;
  const float* Pn = tile_ptr(grp + (int)gridDim.x);   // (past the last group: re-reads the last tile, never used)
  tg_f32x16 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
  auto slab = [&](auto Gc) __attribute__((always_inline)) {
    constexpr int g = decltype(Gc)::value, SL = g & 1;
    // ---- this slab's A operand from the raw words: 32 values (position P = 8 s + t, k-step s = 2 (chunk of the slab) + ks); value = bf16 hi + bf16 lo, or
    // (f16in: split-FP16 fragments) the hi plane as it stands
    tg_u32x16 H;
    tg_u32x6 xh6, xl6;
    int eb;
    if (f16in) eb = tg_mx_operand_f16frag(raw, H, xh6, xl6);
    else {
    float v[32];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float4 fh = raw[ci][2 * ks], fl = raw[ci][2 * ks + 1];
        const unsigned uh[4] = {__float_as_uint(fh.x), __float_as_uint(fh.y), __float_as_uint(fh.z), __float_as_uint(fh.w)};
        const unsigned ul[4] = {__float_as_uint(fl.x), __float_as_uint(fl.y), __float_as_uint(fl.z), __float_as_uint(fl.w)};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          v[16 * ci + 8 * ks + 2 * d] = __uint_as_float(uh[d] << 16) + __uint_as_float(ul[d] << 16);
          v[16 * ci + 8 * ks + 2 * d + 1] = __uint_as_float(uh[d] & 0xffff0000u) + __uint_as_float(ul[d] & 0xffff0000u);
        }
      }
    float amax = 0.f;
    float lo[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      H[i] = tg_hi2_f16_amax(v[2 * i], v[2 * i + 1], amax);
      tg_lo2_f32(v[2 * i], v[2 * i + 1], H[i], lo[2 * i], lo[2 * i + 1]);
    }
    eb = __builtin_amdgcn_frexp_expf(amax) + 124;   // block scale as in tgemm_mx_kernel: the largest value lands in [4, 8)
    eb = eb < 12 ? 12 : (eb > 254 ? 254 : eb);
    const float scf = __builtin_bit_cast(float, eb << 23);
    xh6 = tg_cvt_pk32_fp6_f16(H, scf);
    const tg_f32x16 le = {lo[0], lo[2], lo[4], lo[6], lo[8], lo[10], lo[12], lo[14], lo[16], lo[18], lo[20], lo[22], lo[24], lo[26], lo[28], lo[30]};
    const tg_f32x16 lod = {lo[1], lo[3], lo[5], lo[7], lo[9], lo[11], lo[13], lo[15], lo[17], lo[19], lo[21], lo[23], lo[25], lo[27], lo[29], lo[31]};
    xl6 = tg_cvt_2xpk16_fp6_f32(le, lod, scf * 0.00048828125f);
    }
    const tg_i32x8 bh6 = {(int)xh6[0], (int)xh6[1], (int)xh6[2], (int)xh6[3], (int)xh6[4], (int)xh6[5], 0, 0};
    const tg_i32x8 bl6 = {(int)xl6[0], (int)xl6[1], (int)xl6[2], (int)xl6[3], (int)xl6[4], (int)xl6[5], 0, 0};
    const int sxh = eb, sxl = eb - 11;

    // the next slab — of this group or the first of the next: weights into the other slot (every wave left it before the barrier that ended the previous iteration;
    // after the last slab: before the barrier that ends the epilogue's reads of `red`), its raw words
    constexpr int nsl = g + 1 < NSC ? g + 1 : 0;
    const float* Pnx = g + 1 < NSC ? P : Pn;
    // (All of the next slab's memory instructions go out HERE, in front of the matrix instructions — not between them as in tgemm_mx_kernel: measured, spreading them
    // over the slab made this kernel 216 instead of 174 us.  Its fragment rows come from HBM, not from L2: a load issued 2 k cycles later lands 2 k cycles later.)
    stage(nsl, std::integral_constant<int, 1 - SL>{});
    load_act(Pnx, nsl);

    // ---- the slab's product, weights read two units ahead (tgemm_mx_kernel); operands swapped: D[sample][channel]
    const tg_u32x4* Lf = reinterpret_cast<const tg_u32x4*>(lds_all + (SL * TGMX_SLOT) / 16);
    const tg_u32x4* La = reinterpret_cast<const tg_u32x4*>(lds_all + (SL * TGMX_SLOT + TGMX_F16B) / 16);
    const tg_u32x4* Lb = reinterpret_cast<const tg_u32x4*>(lds_all + (SL * TGMX_SLOT + TGMX_F16B + TGMX_IMA) / 16);
    tg_u32x4 ra[3], rb[3];
    auto rdW = [&](auto Uc) __attribute__((always_inline)) {
      constexpr int u = decltype(Uc)::value, rt = u / 6, k = u % 6, r = u % 3;
      if constexpr (k < 4) ra[r] = Lf[(k * NRT + rt) * 64 + lane];
      else { ra[r] = La[(rt * 2 + (k - 4)) * 64 + lane]; rb[r] = Lb[(rt * 2 + (k - 4)) * 64 + lane]; }
    };
    rdW(std::integral_constant<int, 0>{});
    rdW(std::integral_constant<int, 1>{});
    tg_static_for<6 * NRT>([&](auto Uc) __attribute__((always_inline)) {
      constexpr int u = decltype(Uc)::value, rt = u / 6, k = u % 6, r = u % 3;
      if constexpr (u + 2 < 6 * NRT) rdW(std::integral_constant<int, u + 2>{});
      if constexpr (k < 4) {
        const tg_f16x8 af = __builtin_bit_cast(tg_f16x8, (tg_u32x4){H[4 * k], H[4 * k + 1], H[4 * k + 2], H[4 * k + 3]});
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(tg_f16x8, ra[r]), acc[rt], 0, 0, 0);
      } else {
        const tg_i32x8 w6 = {(int)ra[r][0], (int)ra[r][1], (int)ra[r][2], (int)ra[r][3], (int)rb[r][0], (int)rb[r][1], 0, 0};
        if constexpr (k == 4) acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bl6, w6, acc[rt], 2, 2, 0, sxl, 0, (int)rb[r][2]);   // a_lo6 x w_hi6
        else acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bh6, w6, acc[rt], 2, 2, 0, sxh, 0, (int)rb[r][2]);                     // a_hi6 x w_lo6
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    tg_wait_vmcnt<0>();
    __syncthreads();
  };
  tg_static_for<NSC>([&](auto Gc) __attribute__((always_inline)) { slab(Gc); });

  // ---- epilogue.  Tile rt: lane (c, hh) holds channel 32 rt + c of the wave's rows m(r, hh) = (r & 3) + 8 (r >> 2) + 4 hh, r = 0 .. 15
  const float* wrow = wts + (size_t)tile_c * 32 + 4 * hh;
  nl_f32x2 wp[8];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    float4 w4 = *(const float4*)(wrow + 8 * gq);
    if (!live) w4 = make_float4(0.f, 0.f, 0.f, 0.f);
    wp[2 * gq] = nl_f32x2{w4.x, w4.y}; wp[2 * gq + 1] = nl_f32x2{w4.z, w4.w};
  }
  float* red = reinterpret_cast<float*>(lds_all + TGMX_SLOT / 16);   // [NW][256] partial sums in slot 1: the last slab's, every wave passed its barrier (slot 0 is being filled)
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    const float b = p_bias[32 * rt + j];
    const nl_f32x2 bb = {b, b};
    nl_f32x2 s = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) s = __builtin_elementwise_fma(nl_lrelu2(nl_f32x2{acc[rt][r], acc[rt][r + 1]} + bb), wp[r >> 1], s);
    float p = s[0] + s[1];
    p += __shfl_xor(p, 32, 64);
    if (hh == 0) red[wave * 256 + 32 * rt + j] = p;
  }
  __syncthreads();
  const int wpr = S >> 5, rays_wg = NW / wpr, R = M / S;
  for (int i = tid; i < rays_wg * 256; i += 64 * NW) {
    const int q = i >> 8, c = i & 255;
    float sum = 0.f;
    for (int w = 0; w < wpr; ++w) sum += red[(q * wpr + w) * 256 + c];
    const int ray = grp * rays_wg + q;
    if (w2) reinterpret_cast<float*>(hcsT)[c * HB + nbuf + q] = sum;
    else if (ray < R) hc[(size_t)ray * 256 + c] = sum;
  }
  __syncthreads();   // `red` is read: the next group's first slab may stage slab 1 over it
  P = Pn;
  if (w2) {
    // ---- feat_mlp.2 on the composited rows (it is linear: applied after the sum, abi.hip do_heads), for the rays this workgroup has collected: once per HB rays and
    // after the last group — the per-ray GEMM launch this replaces cost 33 us whatever the batch.  feat[ray][n] = sum_k hc[ray][k] W2[n][k] + wsum[ray] b2[n]; the weights
    // are G_FEAT2's packed fp32 matrix ([k][npad], row 256 = the bias), read once per flush, coalesced over n; two thread sets of npad, eight rays each.
    if (nbuf == 0) grp0 = grp;
    nbuf += rays_wg;
    const bool last = grp + (int)gridDim.x >= ngroups;
    if (nbuf + rays_wg > HB || last) {
      const int n = tid % npad, set = tid / npad;
      if (set < 2 && n < C) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = 0.f;
#pragma unroll 8   // (unroll 32 — four times the loads in flight — measured 252 instead of 178 us for the whole kernel: the allocator then reshapes the main loop)
        for (int k = 0; k < 256; ++k) {
          const float b = w2[(size_t)k * npad + n];
          const float4 h0 = hcsT[k * (HB / 4) + set * 2], h1 = hcsT[k * (HB / 4) + set * 2 + 1];
          a[0] = fmaf(h0.x, b, a[0]); a[1] = fmaf(h0.y, b, a[1]); a[2] = fmaf(h0.z, b, a[2]); a[3] = fmaf(h0.w, b, a[3]);
          a[4] = fmaf(h1.x, b, a[4]); a[5] = fmaf(h1.y, b, a[5]); a[6] = fmaf(h1.z, b, a[6]); a[7] = fmaf(h1.w, b, a[7]);
        }
        const float b2 = w2[(size_t)256 * npad + n];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int slot = set * 8 + i;
          const int ray = (grp0 + (slot / rays_wg) * (int)gridDim.x) * rays_wg + slot % rays_wg;
          if (slot < nbuf && ray < R) feat[(size_t)ray * C + n] = fmaf(wsum[ray], b2, a[i]);
        }
      }
      nbuf = 0;
      __syncthreads();   // the rows are read: the next group's sums may overwrite them
    }
  }
  }
}

// (Two row tiles per wave — four waves, one per SIMD, 64 samples x 256 channels in 256 AGPR accumulators, every weight fragment read from LDS feeding two matrix
// instructions — was built and measured: 251 + 256 registers, no scratch, parity green, and SLOWER: 192-195 against 156-163 us on the same box, with the weight
// fragments read two or four units ahead alike.  The kernel streams 537 MB of fragments (0.11 ms of HBM time at the ~5 TB/s these kernels see) in lock-step
// phases; with one wave per SIMD nothing runs under the operand conversion or the load waits.  Removed; profiles/r6_feat_comp_two_tiles.txt.)

// ====================================================================================================================
// conv1 of the ray U-Net (W = 256 -> 64, k = 3, S = 128; round 6): `tgemm_kernel<2, 4, .., NL_EPI_LNSLAB>` ran this layer at ~2 k cycles per 32-k chunk for 12 matrix
// instructions of 32 — the chunk loop fetches one chunk ahead (weights through registers into LDS, a barrier per chunk), which covers 384 cycles of a ~1.5 k-cycle load.
// With 64 output columns the accumulators are 32 registers, so this kernel can afford what the 256-wide one cannot: the raw activation words of THREE chunks in flight
// (48 registers), the weights by buffer LDS-DMA into a four-slot ring three chunks ahead (8 KB per slot: 33 KB per workgroup), ~110 registers = four workgroups per CU to
// cover the one barrier per chunk that is left.  The layer's K structure (eight 32-channel blocks x three taps of feature_agg, from the chain kernel's fragment image or
// fp32 rows) is a compile-time chunk table as in tgemm_mx_kernel; the product is the same three-term split-bf16 in the same order as tgemm_kernel's (bit-identical
// accumulators), the epilogue its NL_EPI_LNSLAB with MaxPool.
constexpr int TGC1_NRT = 2, TGC1_NW = 4, TGC1_D = 3, TGC1_NB = TGC1_D + 1, TGC1_NCH = 24;
#ifndef TGC1_KO   // knock-outs for timing experiments (results wrong): 1 = no cross terms (one matrix instruction per product instead of three), 2 = every tap reads the centre row, 4 = the side taps are not loaded at all
#define TGC1_KO 0
#endif
template <bool X3, bool F16 = false>   // F16: split-FP16 fragments (NlGemmSeg::frag == 3) against the layer's fp16 hi / lo weight stream (p_bst then points at it): three-term split-FP16
__global__ __launch_bounds__(64 * TGC1_NW, 4) void tgemm_conv1_kernel(const NlGemmArgs a, const char* __restrict__ p_bst, float* __restrict__ p_c, const float* __restrict__ p_zeros,
                                                                      const float* __restrict__ p_bias) {
  constexpr int NRT = TGC1_NRT, NW = TGC1_NW, D = TGC1_D, NB = TGC1_NB, NCH = TGC1_NCH;
  constexpr int PARTS = X3 ? 2 : 1;
  constexpr int CHB = 4 * NRT * 1024;              // bytes per chunk in the global stream (hi and lo parts are always stored)
  constexpr int SLOTB = PARTS * 2 * NRT * 1024;    // bytes per LDS slot
  constexpr int PPW = SLOTB / 1024 / NW;           // 1-KB pieces per wave and chunk
  static_assert(PPW * NW * 1024 == SLOTB, "pieces per wave");
  __shared__ uint4 lds_all[NB * SLOTB / 16 + NRT * 8 * 2 + 8];
  float* sbias = reinterpret_cast<float*>(lds_all + NB * SLOTB / 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31;
  const int tile = blockIdx.x * NW + wave;
  const int m = tile * 32 + j;
  const bool mok = m < a.M;
  const int q = m / a.So, t = m - q * a.So;

  for (int i = tid; i < NRT * 32; i += 64 * NW) sbias[i] = (p_bias && i < a.N) ? p_bias[i] : 0.f;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p_bst, 0, 0x7fffffff, 0x00020000);
  const unsigned lane16 = lane * 16;
  auto stage_piece = [&](auto Cc, auto Ic) __attribute__((always_inline)) {   // piece i of chunk c -> slot c % NB (compile-time: see tgemm_mx_kernel)
    constexpr int c = decltype(Cc)::value, slot = c % NB, i = decltype(Ic)::value;
    const int pp = wave + NW * i;
    unsigned so = (unsigned)c * CHB + pp * 1024;
    asm volatile("" : "+s"(so));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lds_all + (slot * SLOTB) / 16 + pp * 64), 16, lane16, so, 0, 0);
  };
  auto stage = [&](auto Cc) __attribute__((always_inline)) {
    tg_static_for<PPW>([&](auto Ic) __attribute__((always_inline)) { stage_piece(Cc, Ic); });
  };
  // activations: three per-tap row pointers worked out once (tgemm_mx_kernel)
  const NlGemmSeg& g0 = a.seg[0];
  const int frg = __builtin_amdgcn_readfirstlane(g0.frag);
  const int fr0 = frg == 3 ? 1 : frg;   // (3: the fragment image in split-FP16 — same layout, F16 instantiation)
  const int cstride0 = fr0 == 1 ? 1024 : 32;
  const float* P0[3]; int okm[3];
#pragma unroll
  for (int tp = 0; tp < 3; ++tp) {
    const int i = t + tp - 1;
    const bool ok = mok && i >= 0 && i < a.Li;
    const int row = q * a.Li + i;
    okm[tp] = ok ? 1 : 0;
    const float* f = fr0 == 1 ? g0.ptr + ((size_t)(row >> 5) * (g0.k >> 4)) * 512 + ((row & 31) + 32 * hh) * 4 : g0.ptr + (size_t)row * g0.ld + (fr0 == 2 ? 4 : 8) * hh;
    P0[tp] = ok ? f : p_zeros;
  }
  auto act_off = [](int fr, int pc) __attribute__((always_inline)) { return fr == 1 ? 256 * pc : 16 * (pc >> 1) + (fr == 2 ? 8 : 4) * (pc & 1); };
  float4 raw[D][4];   // the raw words of the chunks in flight (ring: chunk c in raw[c % D])
  auto load_act = [&](auto Cc) __attribute__((always_inline)) {
    constexpr int c = decltype(Cc)::value, tp = (TGC1_KO & 2) ? 1 : c % 3, cb = c / 3;   // (TGC1_KO & 2: every tap reads the centre row — timing experiment, results wrong)
    const float* p = P0[tp] + okm[tp] * (cb * cstride0);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) raw[c % D][pc] = *(const float4*)(p + act_off(fr0, pc));
  };
  auto load_act_piece = [&](auto Cc, auto Pc) __attribute__((always_inline)) {
    constexpr int c = decltype(Cc)::value, tp = (TGC1_KO & 2) ? 1 : c % 3, cb = c / 3, pc = decltype(Pc)::value;
    const float* p = P0[tp] + okm[tp] * (cb * cstride0);
    raw[c % D][pc] = *(const float4*)(p + act_off(fr0, pc));
  };

  tg_f32x16 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

  // prologue: chunks 0 .. D - 1 on their way
  tg_static_for<D>([&](auto Cc) __attribute__((always_inline)) { stage(Cc); load_act(Cc); });

#if defined(TG_TRACE) && defined(TG_TRACE_C1)   // (tools/tgmx_trace.py conv1, a -DTG_TRACE -DTG_TRACE_C1 build: phases of chunks 8 .. 13 of block 0, wave 0)
#define TGC1_T(k) do { if (blockIdx.x == 0 && wave == 0 && g >= 8 && g < 14) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) tg_trace[5 * (g - 8) + (k)] = t_; } } while (0)
#else
#define TGC1_T(k)
#endif
  tg_static_for<NCH>([&](auto Gc) __attribute__((always_inline)) {
    constexpr int g = decltype(Gc)::value;
    TGC1_T(0);
    // chunk g has landed when at most the later chunks' operations are in flight (each chunk: PPW DMA pieces + 4 row loads per wave; vmcnt retires in order)
    constexpr int nlater = (g + D - 1 < NCH ? D - 1 : NCH - 1 - g);
    // (TGC1_KO & 4, timing only: the side taps are not loaded — the centre chunk's words stand in for them: a third of the row-load instructions)
    constexpr int later = (TGC1_KO & 4) ? nlater * PPW + 4 * ((g + 1 < NCH && g + 1 <= g + nlater && (g + 1) % 3 == 1) + (g + 2 <= g + nlater && (g + 2) % 3 == 1)) : nlater * (PPW + 4);
    tg_wait_vmcnt<later>();
    TGC1_T(1);
    __syncthreads();   // every wave's pieces of chunk g are in LDS; every wave has left slot (g + D) % NB = (g - 1) % NB
    TGC1_T(2);
    tg_bf16x8 bh[2], bl[2];
    {
      const float4 (&rw4)[4] = raw[(TGC1_KO & 4) ? 1 : g % D];
      if (fr0 == 1) {   // fragment image: [k-step 0: hi | lo | k-step 1: hi | lo] (wave-uniform branch around vector moves only)
        bh[0] = __builtin_bit_cast(tg_bf16x8, rw4[0]); bl[0] = __builtin_bit_cast(tg_bf16x8, rw4[1]);
        bh[1] = __builtin_bit_cast(tg_bf16x8, rw4[2]); bl[1] = __builtin_bit_cast(tg_bf16x8, rw4[3]);
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const float v[8] = {rw4[2 * ks].x, rw4[2 * ks].y, rw4[2 * ks].z, rw4[2 * ks].w, rw4[2 * ks + 1].x, rw4[2 * ks + 1].y, rw4[2 * ks + 1].z, rw4[2 * ks + 1].w};
          if constexpr (F16) tg_split8_f16(v, bh[ks], bl[ks]); else tg_split8<X3>(v, bh[ks], bl[ks]);
        }
      }
    }
    // chunk g + D's memory instructions (PPW weight pieces + 4 raw-word loads per wave) go out BETWEEN the matrix instructions below (late round 6: issued in one go
    // behind the barrier they cost the wave 730-1 320 cycles of a 2-3 k-cycle chunk — sixteen waves per CU arrive at once; tools/tgmx_trace.py conv1)
    auto mem_slot = [&](auto Mc) __attribute__((always_inline)) {
      constexpr int m = decltype(Mc)::value;
      if constexpr (g + D < NCH) {
        if constexpr (m < PPW) stage_piece(std::integral_constant<int, g + D>{}, Mc);
        else if constexpr (m < PPW + 4 && (!(TGC1_KO & 4) || (g + D) % 3 == 1)) load_act_piece(std::integral_constant<int, g + D>{}, std::integral_constant<int, m - PPW>{});
      }
    };
    TGC1_T(3);
    const tg_bf16x8* L = reinterpret_cast<const tg_bf16x8*>(lds_all + ((g % NB) * SLOTB) / 16);
    tg_static_for<2 * NRT>([&](auto Tc) __attribute__((always_inline)) {
        constexpr int tt = decltype(Tc)::value, ks = tt / NRT, rt = tt % NRT;
        const tg_bf16x8 ah = L[((0 * 2 + ks) * NRT + rt) * 64 + lane];
        if (X3 && !(TGC1_KO & 1)) {
          const tg_bf16x8 al = L[((1 * 2 + ks) * NRT + rt) * 64 + lane];
          acc[rt] = tg_mfma<F16>(al, bh[ks], acc[rt]);
          acc[rt] = tg_mfma<F16>(ah, bl[ks], acc[rt]);
        }
        acc[rt] = tg_mfma<F16>(ah, bh[ks], acc[rt]);
        // two memory instructions behind each of the first matrix groups, then one (PPW + 4 = 5 or 6 of them over 2 NRT = 4 groups)
        mem_slot(std::integral_constant<int, (tt < 2 ? 2 * tt : 2 + tt)>{});
        if constexpr (tt < 2) mem_slot(std::integral_constant<int, 2 * tt + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    __builtin_amdgcn_sched_barrier(0);
    TGC1_T(4);
  });

  // ---- epilogue: LayerNorm over the ray's (128 x 64) slab + ELU + MaxPool(2) (tgemm_kernel's NL_EPI_LNSLAB, one ray per workgroup)
  __syncthreads();   // (sbias was written before the loop's first barrier; red lies behind it)
  float* red = sbias + NRT * 32;
  float s1 = 0.f;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = 32 * rt + 8 * gq + 4 * hh;
      const float4 b4 = *(const float4*)(sbias + n);
      nl_f32x2 p0, p1;
      nl_bias_sum4(acc[rt][4 * gq + 0], acc[rt][4 * gq + 1], acc[rt][4 * gq + 2], acc[rt][4 * gq + 3], b4, p0, p1, s1);
      acc[rt][4 * gq + 0] = p0[0]; acc[rt][4 * gq + 1] = p0[1]; acc[rt][4 * gq + 2] = p1[0]; acc[rt][4 * gq + 3] = p1[1];
    }
  s1 = wave_sum(s1);
  if (lane == 0) red[wave] = s1;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) tot += red[w];
  const float cnt = (float)a.So * (float)a.N;
  const float mean = tot / cnt;
  float s2;
  {
    nl_f32x2 sp = {0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) nl_sumsq_dev2(sp, acc[rt][r], acc[rt][r + 1], mean);
    s2 = sp[0] + sp[1];
  }
  s2 = wave_sum(s2);
  if (lane == 0) red[NW + wave] = s2;
  __syncthreads();
  float tot2 = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) tot2 += red[NW + w];
  const float rstd = 1.f / sqrtf(tot2 / cnt + a.ep_eps);
  const float* grow = a.ep_gamma + ((size_t)wave * NRT * 4 * 64 + lane) * 4;   // (a ray = the workgroup's four waves: wave = row tile inside the ray)
  const float* brow = a.ep_beta + ((size_t)wave * NRT * 4 * 64 + lane) * 4;
  const bool pool = a.ep_pool != 0;
  float* orow_p = p_c + (size_t)(pool ? q * (a.So / 2) + (t >> 1) : m) * a.ldc;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = 32 * rt + 8 * gq + 4 * hh;
      const float4 g4 = *(const float4*)(grow + (rt * 4 + gq) * 256), be4 = *(const float4*)(brow + (rt * 4 + gq) * 256);
      float4 v = nl_ln_elu4(acc[rt][4 * gq + 0], acc[rt][4 * gq + 1], acc[rt][4 * gq + 2], acc[rt][4 * gq + 3], mean, rstd, g4, be4);
      if (pool) {
        v.x = nl_max_lane_xor1(v.x); v.y = nl_max_lane_xor1(v.y);
        v.z = nl_max_lane_xor1(v.z); v.w = nl_max_lane_xor1(v.w);
      }
      if (mok && (!pool || !(j & 1))) *(float4*)(orow_p + n) = v;
    }
}


// ====================================================================================================================
// Per-sample chains around the neural-point branch (W = 256), one persistent workgroup per CU, a wave keeps its 32 rows:
//   chain (after the branch):  G = ELU(out_fc.2(t64))  (ibrnet.py:104-106)  ->  feature_agg = LayerNorm(fc(O) + G) * wscale  (ibrnet.py:
//     110-117, model.py:419-427)  ->  feat_mlp.0 + LeakyReLU (model.py:85-89)  and the feature_agg columns of rgb_blending_mlp.0 (model.py:532)
//   query (before the branch):  G as above  ->  Q = w_qs(G)  (model.py:391-396)
// G (N x W) is recomputed from out_fc's 64-wide hidden rows in both and never written or read; the LayerNorm output stays in the
// accumulator registers, is split to bf16 hi/lo in place and IS the B operand of the next two products (their weight streams are packed
// with K in accumulator order, like the fused neural-point kernel); feature_agg is written (the ray U-Net reads it), never read back.
// The K-outer chunk pipeline of tgemm_kernel otherwise, with the weights by buffer LDS-DMA into a 4-slot ring.
//
// Memory instructions are scheduled, not issued where the data becomes available.  A CU takes one 16-byte-per-lane row store per ~70
// cycles and one row load (lane = row: 64 cache lines) per ~64 (tools/ubench/vmem_issue.hip), and its 4 waves run in step: issued as
// bursts (32 stores after an epilogue, 24 loads for the next tile) every wave stood behind the queue for ~300 cycles per instruction,
// ~35 % of a tile, while the matrix pipes idled; HBM was never the limit (dropping every store at the bounds check changed nothing).
// With >= 12 MFMAs (or as much VALU work) between two of a wave's instructions the same traffic is free, so each one has a slot:
//   chain:  out_fc.2 / fc chunks 0..3: 4 attention-row loads each between their MFMAs | ELU epilogue, fc chunks 2, 3 and the first half
//           of the LayerNorm epilogue: the 24 feat_mlp.0-output stores the PREVIOUS tile left | second half of the LayerNorm epilogue
//           and feat_mlp.0's chunks: the 32 feature_agg stores + the NEXT tile's 8 hidden-row loads | feat_mlp.0's epilogue and the first
//           blend chunks: its first 8 output stores.  Rows wait in `fa` (feature_agg after the LayerNorm, feat_mlp.0's output later).
//   query:  out_fc.2 and w_qs chunks 0..3: the previous tile's 16 query-row stores | w_qs chunks 4..7: the next tile's hidden-row loads
struct NlChainArgs {
  const float* O; const float* T64; const float* wscale; const float* gamma; const float* beta; float eps;
  const char* wbase; unsigned off_g2, off_fc, off_f0, off_ba, off_q;   // weight streams as byte offsets into one packed blob
  const float* bias_g2; const float* bias_f0;
  float* FA; float* fth; float* blA; float* Q;
  int M;
};


enum { CK_G2 = 0, CK_FC, CK_F0, CK_BL, CK_Q, CK_NOP };
#ifdef CHAIN_TRACE   // debug build (tools/chain_trace.py): cycle counter of block 0, wave 0 at [kernel][tile][chunk][before wait | after barrier | end of slot]
__device__ unsigned long long chain_trace[2 * 4 * 96];
#define CHAIN_T(k)                                                                                                  \
  do {                                                                                                              \
    if (blockIdx.x == 0 && wave == 0 && trace_it < 4) {                                                             \
      const unsigned long long t_ = __builtin_readcyclecounter();                                                   \
      if (lane == 0) chain_trace[((QUERY ? 1 : 0) * 4 + trace_it) * 96 + 3 * c + (k)] = t_;                         \
    }                                                                                                               \
  } while (0)
#else
#define CHAIN_T(k)
#endif

// weight chunks (32 k each) per 128-row tile: chain = out_fc.2 0,1 | fc 2..5 | feat_mlp.0 6..13 | blend projection 14..21 | 2 empty;
// chain without feat_mlp.0 (FEAT = false: the f16mx render path since round 6, where feat_comp_mx_kernel runs that layer — and early termination): out_fc.2 0,1 |
// fc 2..5 | blend projection 6..13 | 2 empty — until late round 6 this program kept the eight feat_mlp.0 slots as barrier + weight DMA + row traffic with nothing to
// multiply: 256 KB of L2 -> LDS traffic and eight barriers per tile for nothing; their row traffic (24 feature_agg stores, the next tile's 8 hidden-row loads) now
// rides in the blend projection's slots;
// query = out_fc.2 0,1 | w_qs 2..9 | 2 empty.  (The ring slot of a chunk is c % 4 at compile time: programs are padded with empty
// chunks — barrier only — to a multiple of 4.)
template <bool QUERY, bool FEAT = true>
struct ChainGeo {
  static constexpr int NST = QUERY ? 2 : (FEAT ? 4 : 3);
  static constexpr int kind_at(int s) {
    return QUERY ? (s == 0 ? CK_G2 : CK_Q) : (s == 0 ? CK_G2 : s == 1 ? CK_FC : (FEAT ? (s == 2 ? CK_F0 : CK_BL) : CK_BL));
  }
  // (the blend projection: 8 k-chunks of 2 row tiles = 8 KB each; FOUR of them travel as one ring chunk — a full 32-KB slot, contiguous in the stream — so the
  // projection costs two barriers per tile instead of eight: late round 6)
  // (likewise w_qs: 8 k-chunks of 4 row tiles = 16 KB each, TWO to a ring chunk: the query program is 2 + 4 chunks + 2 empty instead of 2 + 8 + 2)
  static constexpr int nch_k(int k) { return k == CK_G2 ? 2 : k == CK_FC ? 4 : k == CK_BL ? 2 : k == CK_Q ? 4 : 8; }
  static constexpr int nrt_k(int k) { return k == CK_NOP ? 0 : 8; }
  static constexpr int start(int s) { int c = 0; for (int i = 0; i < s; ++i) c += nch_k(kind_at(i)); return c; }
  static constexpr int NREAL = start(NST), NCH = (NREAL + 3) / 4 * 4;
  static constexpr int cm(int c) { return ((c % NCH) + NCH) % NCH; }
  static constexpr int spos(int c) { int s = 0; for (int i = 1; i < NST; ++i) if (cm(c) >= start(i)) s = i; return s; }
  static constexpr int kind(int c) { return cm(c) >= NREAL ? CK_NOP : kind_at(spos(c)); }
  static constexpr int idx(int c) { return cm(c) - start(spos(c)); }
  static constexpr int nrt(int c) { return nrt_k(kind(c)); }
  static constexpr bool last(int c) { return kind(c) != CK_NOP && idx(c) == nch_k(kind(c)) - 1; }
  // VMEM operations other than LDS-DMA pieces issued in chunk c's slot after its own pieces went out (the memory schedule above, per
  // lane-instruction).  Waits count them: vmcnt retires in issue order.
  static constexpr int post(int c, bool feat) {
    const int k = kind(c), i = idx(c);
    if (QUERY) return k == CK_G2 ? 4 : k == CK_Q ? 4 : 0;
    if (k == CK_G2) return 4 + (i == 1 && feat ? 8 : 0);                    // attention rows | ELU epilogue: previous tile's rows 8..15
    if (k == CK_FC) return (i < 2 ? 4 : feat ? 4 : 0) + (i == 3 ? 1 + 8 + (feat ? 8 : 0) : 0);   // + scale, LayerNorm epilogue
    if (k == CK_F0) return 4 + (i == 7 && feat ? 6 : 0);                    // 3 feature_agg + 1 hidden-row | epilogue: own rows 0..5
    if (k == CK_BL) return (feat ? (i == 0 ? 2 : 0) : 16) + (i == 1 ? 4 : 0);   // own rows 6, 7 (no feat_mlp.0: 4 x (3 feature_agg + 1 hidden-row)) | blend projection rows
    return 0;
  }
};

// FRAGOUT (chain program only; round 5): feature_agg leaves the kernel as the split-bf16 B fragments the LayerNorm epilogue builds anyway (Xh / Xl: feat_mlp.0's
// operand) in the fragment-native layout of NlGemmSeg::frag — 32 coalesced 1-KB stores per tile instead of 32 row stores — and the ray U-Net's conv1 / conv_out
// (three taps each) load them as they are: no fp32 row loads (lane = row: 64 cache lines per instruction), no hi / lo split per tap.  Same bytes, same values.
// F16FRAG (late round 6; the f16mx render path at W = 256, S = 128, where every consumer of the image multiplies in fp16-based arithmetic): the fragments are split-FP16
// (hi = f16(v), lo = f16(v - hi)) instead of split-bf16 — the same 4 bytes per value in the same places.  conv_out and feat_comp_mx_kernel then take the hi plane AS their
// f16 operand and turn the lo plane into the residual fp6 image with one instruction (~45 instead of ~170 vector instructions per 64-k slab and wave), conv1 multiplies
// it as three-term split-FP16 (2^-22 per product instead of 2^-16), and so does the blend projection in here (its fp16 weight stream: `off_ba` then points at it).
template <bool X3, bool FEAT, bool QUERY, bool FRAGOUT = false, bool F16FRAG = false>
__global__ __launch_bounds__(256, 1) void sample_chain_kernel(const NlChainArgs a, const int ntiles) {
  static_assert(!(FRAGOUT && QUERY), "the query program has no feature_agg");
  static_assert(!F16FRAG || (FRAGOUT && X3 && !FEAT && !QUERY), "split-FP16 fragments: the three-term chain program without feat_mlp.0");
  using Geo = ChainGeo<QUERY, FEAT>;
  if constexpr (F16FRAG) __builtin_amdgcn_s_setreg(1473, 1);   // MODE.FP16_OVFL: the split-FP16 rows saturate instead of overflowing to inf (as their consumers' conversions did)
  constexpr int NW = 4, PARTS = X3 ? 2 : 1, NCH = Geo::NCH, NB = 4;
  constexpr int SLOT16 = PARTS * 2 * 8 * 64;   // 16-B units per ring slot (sized for 8 row tiles)
  __shared__ uint4 lds_all[NB * SLOT16 + 4 * 64];
  tg_bf16x8 (*ring)[SLOT16] = reinterpret_cast<tg_bf16x8 (*)[SLOT16]>(lds_all);
  float* stab = reinterpret_cast<float*>(lds_all + NB * SLOT16);   // gamma | beta | feat_mlp.0 bias | out_fc.2 bias
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31;
  int tile = (int)nl_xcd_block();
  if (tile >= ntiles) return;
  // stores go through buffer descriptors: rows past M (and the deferred stores of the tile before the first) carry an out-of-range
  // offset and are dropped, so every store instruction is always issued and the vmcnt bookkeeping below is exact
  constexpr unsigned DROP = 0x80000000u;
  // (FRAGOUT: whole 32-row tiles of 32 KB; a tile past the last row is out of range and dropped like a row past M otherwise)
  const __amdgpu_buffer_rsrc_t rFA = __builtin_amdgcn_make_buffer_rsrc((void*)(QUERY ? a.Q : a.FA), 0, FRAGOUT ? ((a.M + 31) >> 5) * 32768 : a.M * (QUERY ? 512 : 1024),
                                                                       0x00020000);
  const __amdgpu_buffer_rsrc_t rFT = __builtin_amdgcn_make_buffer_rsrc((void*)(FEAT ? a.fth : a.FA), 0, a.M * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBL = __builtin_amdgcn_make_buffer_rsrc((void*)a.blA, 0, a.M * 128, 0x00020000);
  for (int i = tid; i < 256; i += 256) {
    if (!QUERY) { stab[i] = a.gamma[i]; stab[256 + i] = a.beta[i]; stab[512 + i] = a.bias_f0 ? a.bias_f0[i] : 0.f; }
    stab[768 + i] = a.bias_g2[i];
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
      // (the blend projection's ring chunk = four 8-KB k-chunks [hi 4 KB | lo 4 KB]: contiguous in three-term mode; single-bf16 mode takes the four hi parts)
      //  w_qs's ring chunk = two 16-KB k-chunks [hi 8 KB | lo 8 KB]: likewise)
      unsigned s2 = so + ((kd == CK_Q && !X3) ? (i / 2) * 16384 + (i % 2) * 4096 : i * ((kd == CK_BL && !X3) ? 8192 : 4096));
      asm volatile("" : "+s"(s2));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lw + (c % NB) * SLOT16 + i * 256), 16, wvoff, s2, 0, 0);
    });
  };

  tg_f32x16 acc[8];
  tg_f32x16 qacc[QUERY ? 4 : 1];           // query: w_qs accumulates beside G, whose row tiles are converted while it runs
  tg_bf16x8 Xh[16], Xl[16];
  tg_f32x4 oraw[QUERY ? 1 : 16];           // attention output rows: fc's B operand, 8 k-steps x 8 floats per lane
  tg_f32x4 traw[8];                        // out_fc hidden rows: out_fc.2's B operand, 4 k-steps x 8 floats per lane
  float fa[QUERY ? 4 : 8][16];             // rows waiting for their scheduled stores (chain: feature_agg, then feat_mlp.0's output; query: Q)
  auto row_of = [&](int t, int& m, int& mm, bool& mok) __attribute__((always_inline)) {
    m = t * 128 + 32 * wave + j;
    mok = m < a.M;
    mm = mok ? m : a.M - 1;
  };
  // (the indices below are compile-time constants after unrolling: every caller sits in an unrolled loop or a static_for)
  auto load_traw = [&](int i, int mm) __attribute__((always_inline)) {
    traw[i] = *reinterpret_cast<const tg_f32x4*>(a.T64 + (size_t)mm * 64 + 8 * hh + 16 * (i >> 1) + 4 * (i & 1));
  };
  auto load_oraw = [&](int i, int mm) __attribute__((always_inline)) {
    oraw[i % (QUERY ? 1 : 16)] = *reinterpret_cast<const tg_f32x4*>(a.O + (size_t)mm * 128 + 8 * hh + 16 * (i >> 1) + 4 * (i & 1));
  };
  // row store q (0..31 | 0..15) of `fa`: 16 B of row tile q / 4 at columns 8 (q % 4) + 4 hh
  auto store_fa = [&](int q, __amdgpu_buffer_rsrc_t rsrc, unsigned rowoff) __attribute__((always_inline)) {
    const int rt = q >> 2, gq = q & 3, rtc = rt % (QUERY ? 4 : 8);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tg_u32x4, tg_f32x4{fa[rtc][4 * gq], fa[rtc][4 * gq + 1], fa[rtc][4 * gq + 2], fa[rtc][4 * gq + 3]}), rsrc,
                                           rowoff + (32 * rt + 8 * gq) * 4, 0, 0);
  };

  // feature_agg store q (0..31) of the current tile: fp32 row piece q of `fa`, or (FRAGOUT) fragment q & 3 = [hi k-step 0 | hi k-step 1 | lo 0 | lo 1] of row tile q >> 2
  unsigned fragoff = 0;   // byte offset of this lane's 16 bytes in the tile's first fragment block
  auto store_FA = [&](int q, unsigned rowoff) __attribute__((always_inline)) {
    if constexpr (FRAGOUT) {
      const int rt = q >> 2, w = q & 3, sI = w & 1, part = w >> 1;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tg_u32x4, part ? Xl[2 * rt + sI] : Xh[2 * rt + sI]), rFA,
                                             fragoff + (unsigned)(((2 * rt + sI) * 2 + part) * 1024), 0, 0);
    } else store_fa(q, rFA, rowoff);
  };
  // one chunk (32 k): 2 k-steps x NRT row tiles x (3 | 1) MFMAs out of ring slot `slot`; filler(step) runs between the MFMA groups
  auto compute = [&](auto Nc, int slot, const tg_bf16x8 (&bh)[2], const tg_bf16x8 (&bl)[2], auto& dst, auto&& filler, int sub = 0) __attribute__((always_inline)) {
    constexpr int NRT = decltype(Nc)::value, nt = 2 * NRT;
    constexpr bool F16 = F16FRAG && NRT == 2;   // the blend projection (the only two-row-tile product of the chain program) on the split-FP16 rows
    const tg_bf16x8* L = ring[slot] + sub * (PARTS * 2 * NRT * 64);   // sub: the k-chunk inside a ring chunk that carries several (the blend projection's four)
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
      if constexpr (F16) {
        dst[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tg_f16x8, al[tt % 3]), __builtin_bit_cast(tg_f16x8, bh[ks]), dst[rt], 0, 0, 0);
        dst[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tg_f16x8, ah[tt % 3]), __builtin_bit_cast(tg_f16x8, bl[ks]), dst[rt], 0, 0, 0);
        dst[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tg_f16x8, ah[tt % 3]), __builtin_bit_cast(tg_f16x8, bh[ks]), dst[rt], 0, 0, 0);
      } else {
      if (X3) {
        dst[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tt % 3], bh[ks], dst[rt], 0, 0, 0);
        dst[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tt % 3], bl[ks], dst[rt], 0, 0, 0);
      }
      dst[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tt % 3], bh[ks], dst[rt], 0, 0, 0);
      }
      filler(tt);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto no_fill = [](int) __attribute__((always_inline)) {};
  // G = ELU(out_fc.2 + bias) of row tile rt, in 7 steps: 4 x 4 values | 2 x bf16 split into k-steps 2 rt, 2 rt + 1 (query only) | clear
  auto g2_epilogue_step = [&](auto Rt, int step) __attribute__((always_inline)) {
    constexpr int rt = decltype(Rt)::value;
    if (step < 4) {
      const int gq = step;
      const float4 b4 = *(const float4*)(stab + 768 + 32 * rt + 8 * gq + 4 * hh);
      const nl_f32x2 e0 = nl_elu_fast2(nl_f32x2{acc[rt][4 * gq + 0], acc[rt][4 * gq + 1]} + nl_f32x2{b4.x, b4.y});
      const nl_f32x2 e1 = nl_elu_fast2(nl_f32x2{acc[rt][4 * gq + 2], acc[rt][4 * gq + 3]} + nl_f32x2{b4.z, b4.w});
      acc[rt][4 * gq + 0] = e0[0]; acc[rt][4 * gq + 1] = e0[1]; acc[rt][4 * gq + 2] = e1[0]; acc[rt][4 * gq + 3] = e1[1];
    } else if (QUERY && step < 6) {
      const int sI = step - 4;
      const float u[8] = {acc[rt][8 * sI], acc[rt][8 * sI + 1], acc[rt][8 * sI + 2], acc[rt][8 * sI + 3],
                          acc[rt][8 * sI + 4], acc[rt][8 * sI + 5], acc[rt][8 * sI + 6], acc[rt][8 * sI + 7]};
      tg_split8<X3>(u, Xh[2 * rt + sI], Xl[2 * rt + sI]);
    } else if (QUERY && step == 6) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
    }
  };
  auto zero_acc = [&](auto N0, auto N1) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = decltype(N0)::value; rt < decltype(N1)::value; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
  };
  using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
  using I8 = std::integral_constant<int, 8>;

  // ---------------------------------------------------------------- pipeline start
  dma_chunk(I0{}); dma_chunk(std::integral_constant<int, 1>{}); dma_chunk(I2{});
  zero_acc(I0{}, I8{});
#pragma unroll
  for (int rt = 0; rt < (QUERY ? 4 : 1); ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[rt][r] = 0.f;
#pragma unroll
  for (int rt = 0; rt < (QUERY ? 4 : 8); ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) fa[rt][r] = 0.f;
  int m_c, mm_c; bool mok_c;
  row_of(tile, m_c, mm_c, mok_c);
#pragma unroll
  for (int i = 0; i < 8; ++i) load_traw(i, mm_c);
#ifdef CHAIN_TRACE
  int trace_it = 0;
#endif
  unsigned prev_row = DROP;   // byte offset of this lane's row in the previous tile's deferred stores
  tg_wait_vmcnt<0>();   // the first pass of the counted waits below assumes nothing older is in flight
  __syncthreads();      // stab

  for (;;) {
    const int tile_next = tile + (int)gridDim.x;
    int m_n, mm_n; bool mok_n;
    row_of(tile_next < ntiles ? tile_next : tile, m_n, mm_n, mok_n);
    const unsigned row1k = mok_c ? (unsigned)m_c * 1024u + 16u * hh : DROP;   // this lane's row in the (N, 256) outputs
    fragoff = (unsigned)(tile * 4 + wave) * 32768u + (unsigned)lane * 16u;
    tg_static_for<NCH>([&](auto Cc) __attribute__((always_inline)) {
      constexpr int c = decltype(Cc)::value, kd = Geo::kind(c), g = Geo::idx(c);
      // chunk c must have landed: younger operations are the pieces of chunks c+1, c+2 and the other traffic issued since chunk c-3
      constexpr int younger = ppw(c + 1) + ppw(c + 2) + Geo::post(c - 3, FEAT) + Geo::post(c - 2, FEAT) + Geo::post(c - 1, FEAT);
      CHAIN_T(0);
#ifdef CHAIN_WAIT0
      tg_wait_vmcnt<0>();
#else
      tg_wait_vmcnt<(younger < 63 ? younger : 63)>();
#endif
      __builtin_amdgcn_s_barrier();
      CHAIN_T(1);
      dma_chunk(std::integral_constant<int, c + 3>{});   // its slot held chunk c-1, which every wave has left
      // The counted waits assume program order = issue order: nothing below may be scheduled in front of these pieces (a row store
      // hoisted above them made the blend chunk's wait one short in bf16 mode, where a chunk is ONE piece per wave: run-to-run
      // different colours)
      __builtin_amdgcn_sched_barrier(0);
      // ---- this slot's share of the row traffic: operation k of the slot goes out after MFMA group 4 k + 1 (always issued: the wait
      // counts stay exact)
      auto mem_op = [&](int k) __attribute__((always_inline)) {
        if constexpr (QUERY) {
          if constexpr (kd == CK_G2) store_fa(4 * g + k, rFA, prev_row);                       // previous tile's rows 0..7
          // (w_qs: q_mem below, per k-chunk)
        } else {
          if constexpr (kd == CK_G2) load_oraw(4 * g + k, mm_c);
          else if constexpr (kd == CK_FC && g < 2) load_oraw(8 + 4 * g + k, mm_c);
          else if constexpr (kd == CK_FC) { if constexpr (FEAT) store_fa(16 + 4 * (g - 2) + k, rFT, prev_row); }   // previous tile's rows 16..23
          else if constexpr (kd == CK_F0) { if (k < 3) store_FA(8 + 3 * g + k, row1k); else load_traw(g, mm_n); }
        }
      };
      auto q_mem = [&](int gq, int k) __attribute__((always_inline)) {   // the row traffic of w_qs k-chunk gq (0 .. 7), operation k (0, 1)
        if (gq < 4) store_fa(8 + 2 * gq + k, rFA, prev_row);                                  // previous tile's rows 8..15
        else load_traw(2 * (gq - 4) + k, mm_n);
      };
      auto bl_mem = [&](int kc) __attribute__((always_inline)) {   // the row traffic of blend k-chunk kc (0 .. 7), in one go in front of its 12 MFMAs
        if constexpr (FEAT) { if (kc < 2) store_fa(6 + kc, rFT, row1k); }
        else {
#pragma unroll
          for (int k = 0; k < 3; ++k) store_FA(8 + 3 * kc + k, row1k);
          load_traw(kc, mm_n);
        }
      };
      auto mem_fill = [&](int tt) __attribute__((always_inline)) { if ((tt & 3) == 1) mem_op(tt >> 2); };
      __builtin_amdgcn_sched_barrier(0);

      tg_bf16x8 bh[2], bl[2];
      if constexpr (kd == CK_FC || kd == CK_G2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          tg_f32x4 u0, u1;
          if constexpr (kd == CK_FC) { u0 = oraw[(4 * g + 2 * ks) % (QUERY ? 1 : 16)]; u1 = oraw[(4 * g + 2 * ks + 1) % (QUERY ? 1 : 16)]; }
          else { u0 = traw[4 * g + 2 * ks]; u1 = traw[4 * g + 2 * ks + 1]; }
          const float v[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
          tg_split8<X3>(v, bh[ks], bl[ks]);
        }
        compute(I8{}, c % NB, bh, bl, acc, mem_fill);
      } else if constexpr (kd != CK_NOP) {
        if constexpr (kd == CK_F0) { bh[0] = Xh[2 * g]; bh[1] = Xh[2 * g + 1]; bl[0] = Xl[2 * g]; bl[1] = Xl[2 * g + 1]; }
        if constexpr (kd == CK_F0) { if constexpr (FEAT) compute(I8{}, c % NB, bh, bl, acc, mem_fill); }
        else if constexpr (kd == CK_BL) {
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            const int kc = 4 * g + sub;
            bl_mem(kc);
            __builtin_amdgcn_sched_barrier(0);
            bh[0] = Xh[2 * kc]; bh[1] = Xh[2 * kc + 1]; bl[0] = Xl[2 * kc]; bl[1] = Xl[2 * kc + 1];
            compute(I2{}, c % NB, bh, bl, acc, no_fill, sub);
          }
        }
        else {   // w_qs: the two k-chunks of this ring chunk; k-chunk gq multiplies G's row tile gq, row tile gq + 1 is converted between its MFMAs
          tg_static_for<2>([&](auto Sc) __attribute__((always_inline)) {
            constexpr int sub = decltype(Sc)::value, gq = 2 * g + sub;
            bh[0] = Xh[2 * gq]; bh[1] = Xh[2 * gq + 1]; bl[0] = Xl[2 * gq]; bl[1] = Xl[2 * gq + 1];
            compute(I4{}, c % NB, bh, bl, qacc, [&](int tt) __attribute__((always_inline)) {
              if constexpr (gq + 1 < 8) g2_epilogue_step(std::integral_constant<int, (gq + 1) % 8>{}, tt);
              if ((tt & 3) == 1) q_mem(gq, tt >> 2);
            }, sub);
          });
        }
      }

      if constexpr (kd == CK_G2 && Geo::last(c)) {
        if constexpr (QUERY) {
#pragma unroll
          for (int step = 0; step < 7; ++step) g2_epilogue_step(I0{}, step);
        } else {   // G stays in the accumulators: fc continues on it (the residual of ibrnet.py:112)
          tg_static_for<8>([&](auto Rt) __attribute__((always_inline)) {
#pragma unroll
            for (int step = 0; step < 4; ++step) g2_epilogue_step(Rt, step);
            if constexpr (FEAT) store_fa(8 + decltype(Rt)::value, rFT, prev_row);
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      }
      if constexpr (kd == CK_FC && Geo::last(c)) {
        // ---- (fc + residual) -> LayerNorm(row) * aggregation scale -> feature_agg: into `fa` for the scheduled stores, and split
        // to bf16 as the next products' B operand
        // (the element-wise arithmetic of this epilogue on pairs: packed fp32 instructions, the same operations per element)
        nl_f32x2 sp1 = {0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
          for (int r = 0; r < 16; r += 4) sp1 += nl_f32x2{acc[rt][r], acc[rt][r + 1]} + nl_f32x2{acc[rt][r + 2], acc[rt][r + 3]};
        float s1 = sp1[0] + sp1[1];
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 / 256.f;
        nl_f32x2 sp2 = {0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
          for (int r = 0; r < 16; r += 2) nl_sumsq_dev2(sp2, acc[rt][r], acc[rt][r + 1], mean);
        float s2 = sp2[0] + sp2[1];
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = 1.f / sqrtf(s2 / 256.f + a.eps);
        const float sc = a.wscale[mm_c];
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
          float (&v)[16] = fa[rt % (QUERY ? 4 : 8)];
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int n = 32 * rt + 8 * gq + 4 * hh;
            const float4 g4 = *(const float4*)(stab + n), be4 = *(const float4*)(stab + 256 + n);
            const nl_f32x2 mm = {mean, mean}, rr = {rstd, rstd}, ss = {sc, sc};
            const nl_f32x2 p0 = ((nl_f32x2{acc[rt][4 * gq + 0], acc[rt][4 * gq + 1]} - mm) * rr * nl_f32x2{g4.x, g4.y} + nl_f32x2{be4.x, be4.y}) * ss;
            const nl_f32x2 p1 = ((nl_f32x2{acc[rt][4 * gq + 2], acc[rt][4 * gq + 3]} - mm) * rr * nl_f32x2{g4.z, g4.w} + nl_f32x2{be4.z, be4.w}) * ss;
            v[4 * gq + 0] = p0[0]; v[4 * gq + 1] = p0[1]; v[4 * gq + 2] = p1[0]; v[4 * gq + 3] = p1[1];
          }
#pragma unroll
          for (int sI = 0; sI < 2; ++sI) {   // accumulator registers 8 s .. 8 s + 7 of row tile rt = k-step 2 rt + s in accumulator order
            const float u[8] = {v[8 * sI], v[8 * sI + 1], v[8 * sI + 2], v[8 * sI + 3], v[8 * sI + 4], v[8 * sI + 5], v[8 * sI + 6], v[8 * sI + 7]};
            if constexpr (F16FRAG) tg_split8_f16_pk(u, Xh[2 * rt + sI], Xl[2 * rt + sI]); else tg_split8<X3>(u, Xh[2 * rt + sI], Xl[2 * rt + sI]);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
          // row tiles 0..3: the previous tile's last rows (they sit in fa[6], fa[7], overwritten in iterations 6, 7); 4..7: this tile's first
          if (rt < 4) { if constexpr (FEAT) { store_fa(24 + 2 * rt, rFT, prev_row); store_fa(25 + 2 * rt, rFT, prev_row); } }
          else { store_FA(2 * (rt - 4), row1k); store_FA(2 * (rt - 4) + 1, row1k); }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (kd == CK_F0 && Geo::last(c)) {
        // ---- feat_mlp.0's rows (bias + LeakyReLU) replace feature_agg in `fa` (its last store went out in front of this chunk)
        if constexpr (FEAT) {
#pragma unroll
          for (int rt = 0; rt < 8; ++rt) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int n = 32 * rt + 8 * gq + 4 * hh;
              const float4 b4 = *(const float4*)(stab + 512 + n);
              float (&v)[16] = fa[rt % (QUERY ? 4 : 8)];
              const nl_f32x2 l0 = nl_lrelu2(nl_f32x2{acc[rt][4 * gq], acc[rt][4 * gq + 1]} + nl_f32x2{b4.x, b4.y});
              const nl_f32x2 l1 = nl_lrelu2(nl_f32x2{acc[rt][4 * gq + 2], acc[rt][4 * gq + 3]} + nl_f32x2{b4.z, b4.w});
              v[4 * gq] = l0[0]; v[4 * gq + 1] = l0[1]; v[4 * gq + 2] = l1[0]; v[4 * gq + 3] = l1[1];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
            if (rt >= 2) store_fa(rt - 2, rFT, row1k);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if constexpr (kd == CK_BL && Geo::last(c)) {
        const unsigned brow = mok_c ? (unsigned)m_c * 128u + 16u * hh : DROP;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tg_u32x4, tg_f32x4{acc[0][4 * gq], acc[0][4 * gq + 1], acc[0][4 * gq + 2], acc[0][4 * gq + 3]}), rBL,
                                                 brow + 32 * gq, 0, 0);
        zero_acc(I0{}, I2{});
      }
      if constexpr (kd == CK_Q && Geo::last(c)) {   // (the previous tile's query rows left `fa` in chunks 0..7)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { fa[rt % (QUERY ? 4 : 8)][r] = qacc[rt % (QUERY ? 4 : 1)][r]; qacc[rt % (QUERY ? 4 : 1)][r] = 0.f; }
      }
      CHAIN_T(2);
    });
#ifdef CHAIN_TRACE
    ++trace_it;
#endif
    prev_row = QUERY ? (mok_c ? (unsigned)m_c * 512u + 16u * hh : DROP) : row1k;
    m_c = m_n; mm_c = mm_n; mok_c = mok_n;
    tile = tile_next;
    if (tile >= ntiles) break;
  }
  // the last tile's deferred rows
  if constexpr (QUERY) {
#pragma unroll
    for (int q = 0; q < 16; ++q) store_fa(q, rFA, prev_row);
  } else if constexpr (FEAT) {
#pragma unroll
    for (int q = 8; q < 32; ++q) store_fa(q, rFT, prev_row);
  }
  tg_wait_vmcnt<0>();   // LDS-DMA prefetched past the last tile must land before the LDS goes to another workgroup
}

}  // namespace

// stream layout helpers (also used by the packer in abi.hip)
int nl_tgemm_nrt(int N) { return N <= 64 ? 2 : (N <= 128 ? 4 : 8); }
size_t nl_tgemm_mx_image_bytes(int Kpad) { return (size_t)((Kpad / 32 + 1) / 2) * TGMX_IMG; }   // fp6 images + scales of a 256-column layer (pack_tgemm_mx6_kernel)
size_t nl_tgemm_stream_bytes(int Kpad, int N) { return (size_t)(Kpad / 32) * 4 * nl_tgemm_nrt(N) * 1024; }

bool nl_tgemm_supported(const NlGemmArgs& a, int precision) {
  if (precision == NL_PREC_F16X3_INTERNAL && (a.epi != NL_EPI_NONE || a.tile_map)) return false;   // plain products only
  if (a.act == NL_ACT_LRELU_MASK && (a.epi != NL_EPI_NONE || a.So > 0 || (!a.ep_maskin && (!a.ep_res || (a.ep_ldres & 3) || (((size_t)a.ep_res) & 15))))) return false;
  if ((a.ep_maskout || a.ep_maskin) && (a.epi != NL_EPI_NONE || a.So > 0 || a.tile_map)) return false;
  if (a.ep_tab && (a.epi != NL_EPI_NONE || a.So > 0 || a.tile_map || a.act == NL_ACT_LRELU_MASK || !a.ep_tabidx || a.ep_tabK <= 0 || (a.ep_ldtab & 3) ||
                   (((size_t)a.ep_tab) & 15)))
    return false;
  if (a.tile_map && (a.epi != NL_EPI_NONE || a.So > 0 || !a.tile_count)) return false;
  if (precision == NL_PREC_F32 || !a.Bst || a.N > 256 || (a.N & 3) || (a.ldc & 3) || (((size_t)a.C) & 15) || a.M <= 0 || !a.zeros) return false;
  if (a.epi == NL_EPI_LNROW && (a.N != 32 * nl_tgemm_nrt(a.N) || a.So > 0 || !a.ep_res || (a.ep_ldres & 3) || (((size_t)a.ep_res) & 15))) return false;
  if (a.epi == NL_EPI_LNSLAB && ((a.So != 192 && a.So != 96 && a.So != 128 && a.So != 64 && a.So != 32 && a.So != 16) || a.M % a.So || a.Li != a.So || a.ostride != 1 || a.ooff != 0 || !a.ep_gamma || !a.ep_beta)) return false;
  for (int s = 0; s < a.nseg; ++s) {
    const NlGemmSeg& g = a.seg[s];
    if (!g.vec || (g.k & 31) || g.rdiv > 1 || g.ld < g.k || g.ntap < 1) return false;
    if (g.frag && precision == NL_PREC_F16X3_INTERNAL) return false;   // (split-bf16 fragments / accumulator K order: the bf16 modes only)
  }
  return true;
}

// f16mx form of an LNSLAB launch (conv_out): 256-wide, one ray = 128 rows, whole 32-k chunks from 16-byte-aligned sources, both weight images present
bool nl_tgemm_mx_supported(const NlGemmArgs& a, int precision) {
  if (precision != NL_PREC_BF16X3 || !a.Bsh_mx || !a.Bmx || a.epi != NL_EPI_LNSLAB || a.So != 128 || a.Li != 128 || a.N != 256 || a.ep_pool || a.tile_map) return false;
  // conv_out's K structure at W = 256 (the kernel's chunk table is a compile-time constant): [feature_agg: 8 blocks x 3 taps | x2: 1 block x 3 taps]
  if (a.nseg != 2 || a.Kpad != 864 || a.seg[0].k != 256 || a.seg[0].ntap != 3 || a.seg[0].ioff != 0 || a.seg[1].k != 32 || a.seg[1].ntap != 3 || a.seg[1].ioff != 0 ||
      a.seg[1].frag != 0)
    return false;
  return nl_tgemm_supported(a, precision);
}

// conv1 of the ray U-Net at W = 256, S = 128 (tgemm_conv1_kernel): one segment of 256 channels x 3 taps, 64 columns, one ray = 128 rows per workgroup
bool nl_tgemm_conv1_supported(const NlGemmArgs& a, int precision) {
  if ((precision != NL_PREC_BF16X3 && precision != NL_PREC_BF16) || a.epi != NL_EPI_LNSLAB || a.So != 128 || a.Li != 128 || a.N != 64 || a.tile_map || a.ep_sig_w || !a.C) return false;
  if (a.nseg != 1 || a.Kpad != 768 || a.seg[0].k != 256 || a.seg[0].ntap != 3 || a.seg[0].ioff != 0) return false;
  return nl_tgemm_supported(a, precision);
}

int nl_tgemm_launch(const NlGemmArgs& a, int precision, hipStream_t st) {
  const bool x3 = precision == NL_PREC_BF16X3;
  const int nrt = nl_tgemm_nrt(a.N);
#ifndef NL_NO_TGEMM_CONV1
  if (nl_tgemm_conv1_supported(a, precision)) {
    const dim3 grid((unsigned)nl_cdiv(a.M, 32 * TGC1_NW));
    if (a.seg[0].frag == 3) {
      if (!x3 || !a.Bsh16) return NL_ERR_UNSUPPORTED;
      hipLaunchKernelGGL((tgemm_conv1_kernel<true, true>), grid, dim3(64 * TGC1_NW), 0, st, a, (const char*)a.Bsh16, a.C, a.zeros, a.bias);
    } else if (x3) hipLaunchKernelGGL(tgemm_conv1_kernel<true>, grid, dim3(64 * TGC1_NW), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias);
    else hipLaunchKernelGGL(tgemm_conv1_kernel<false>, grid, dim3(64 * TGC1_NW), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias);
    return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
  }
#endif
  if (nl_tgemm_mx_supported(a, precision)) {
    hipLaunchKernelGGL(tgemm_mx_kernel, dim3((unsigned)nl_cdiv(a.M, 32 * TGMX_NW)), dim3(64 * TGMX_NW), 0, st, a, (const char*)a.Bsh_mx, (const char*)a.Bmx, a.C, a.zeros, a.bias);
    return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
  }
  for (int sI = 0; sI < a.nseg; ++sI) if (a.seg[sI].frag == 3) return NL_ERR_UNSUPPORTED;   // split-FP16 fragments: tgemm_conv1_kernel / tgemm_mx_kernel only
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
  if (precision == NL_PREC_F16X3_INTERNAL) {
    dim3 grid((unsigned)nl_cdiv(a.M, 32 * 4));
    if (nrt == 8) hipLaunchKernelGGL((tgemm_kernel<8, 4, true, NL_EPI_NONE, true>), grid, dim3(256), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias);
    else if (nrt == 4) hipLaunchKernelGGL((tgemm_kernel<4, 4, true, NL_EPI_NONE, true>), grid, dim3(256), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias);
    else hipLaunchKernelGGL((tgemm_kernel<2, 4, true, NL_EPI_NONE, true>), grid, dim3(256), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias);
    return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
  }
  if (a.epi == NL_EPI_LNSLAB && (a.So == 192 || a.So == 96)) {   // six waves = one ray of 192 rows or two of 96
#define NL_TG6(NRT, X3) hipLaunchKernelGGL((tgemm_kernel<NRT, 6, X3, NL_EPI_LNSLAB>), dim3((unsigned)nl_cdiv(a.M, 192)), dim3(384), 0, st, a, (const char*)a.Bst, a.C, a.zeros, a.bias)
    if (nrt == 8) { if (x3) NL_TG6(8, true); else NL_TG6(8, false); }
    else if (nrt == 4) { if (x3) NL_TG6(4, true); else NL_TG6(4, false); }
    else { if (x3) NL_TG6(2, true); else NL_TG6(2, false); }
#undef NL_TG6
    return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
  }
  if (nrt == 8) { if (x3) NL_TG(8, 4, true); else NL_TG(8, 4, false); }
  else if (nrt == 4) { if (x3) NL_TG(4, 4, true); else NL_TG(4, 4, false); }
  else { if (x3) NL_TG(2, 4, true); else NL_TG(2, 4, false); }
#undef NL_TG
  return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}

#ifdef TG_TRACE
extern "C" __attribute__((visibility("default"))) int nl_debug_tg_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_trace), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif
#ifdef CHAIN_TRACE
extern "C" __attribute__((visibility("default"))) int nl_debug_chain_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(chain_trace), sizeof(unsigned long long) * 2 * 4 * 96) == hipSuccess ? 0 : -1;
}
#endif

namespace {
int chain_grid(int ntiles, dim3* grid) {
  const int num_cu = nl_persistent_cus();
  if (num_cu < 0) return num_cu;
  *grid = dim3(ntiles < num_cu ? nl_xcd_grid(ntiles) : num_cu);
  return NL_OK;
}
}  // namespace

// feature_agg (+ feat_mlp.0's rows when fth != null, + the blend projection) from the attention rows O and out_fc's hidden rows T64
int nl_launch_sample_chain(const float* O, const float* T64, const float* wscale, const float* gamma, const float* beta, float eps, const void* wbase,
                           size_t off_g2, const float* bias_g2, size_t off_fc, size_t off_f0, size_t off_ba, const float* bias_f0, float* FA, float* fth,
                           float* blA, int64_t M, int precision, hipStream_t st, bool frag_out, bool frag_f16) {
  if (M <= 0) return NL_OK;
  if ((int64_t)(M + 31) * 1024 > 0x7fffffffll) return NL_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  const int ntiles = (int)nl_cdiv(M, 128);
  dim3 grid;
  if (chain_grid(ntiles, &grid) != NL_OK) return NL_ERR_HIP;
  NlChainArgs a{};
  a.O = O; a.T64 = T64; a.wscale = wscale; a.gamma = gamma; a.beta = beta; a.eps = eps; a.wbase = (const char*)wbase;
  a.off_g2 = (unsigned)off_g2; a.off_fc = (unsigned)off_fc; a.off_f0 = (unsigned)off_f0; a.off_ba = (unsigned)off_ba;
  a.bias_g2 = bias_g2; a.bias_f0 = bias_f0; a.FA = FA; a.fth = fth; a.blA = blA; a.M = (int)M;
  const bool x3 = precision == NL_PREC_BF16X3;
#define NL_CHAIN(FR)                                                                                                            \
  do {                                                                                                                          \
    if (x3 && fth) hipLaunchKernelGGL((sample_chain_kernel<true, true, false, FR>), grid, dim3(256), 0, st, a, ntiles);         \
    else if (x3) hipLaunchKernelGGL((sample_chain_kernel<true, false, false, FR>), grid, dim3(256), 0, st, a, ntiles);          \
    else if (fth) hipLaunchKernelGGL((sample_chain_kernel<false, true, false, FR>), grid, dim3(256), 0, st, a, ntiles);         \
    else hipLaunchKernelGGL((sample_chain_kernel<false, false, false, FR>), grid, dim3(256), 0, st, a, ntiles);                 \
  } while (0)
  if (frag_f16) {   // split-FP16 fragments (off_ba: the blend projection's fp16 stream)
    if (!x3 || fth || !frag_out) return NL_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((sample_chain_kernel<true, false, false, true, true>), grid, dim3(256), 0, st, a, ntiles);
  } else if (frag_out) NL_CHAIN(true); else NL_CHAIN(false);
#undef NL_CHAIN
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
  if (precision == NL_PREC_BF16X3) hipLaunchKernelGGL((sample_chain_kernel<true, false, true>), grid, dim3(256), 0, st, a, ntiles);
  else hipLaunchKernelGGL((sample_chain_kernel<false, false, true>), grid, dim3(256), 0, st, a, ntiles);
  return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}

// feat_mlp.0 + LeakyReLU + compositing along the ray (feat_comp_mx_kernel): W = 256, rays of 32 .. 256 samples whose waves divide a workgroup of 8 or 6
bool nl_feat_comp_mx_supported(int W, int S, int64_t N) {
  if (W != 256 || S < 32 || (S & 31) || N <= 0 || (N & 31) || N % S || N * 256 > 0x7fffffffll * 4) return false;
  const int wpr = S >> 5;
  return (8 % wpr) == 0 || (6 % wpr) == 0;
}
int nl_launch_feat_comp_mx(const float* fa_frag, const float* wts, int64_t N, int S, const void* bsh, const void* bmx, const float* bias, float* hc, hipStream_t st,
                           const float* w2, int npad, int C, const float* wsum, float* feat, bool frag_f16) {
  if (!nl_feat_comp_mx_supported(256, S, N) || !fa_frag || !wts || !bsh || !bmx || !bias || !hc || (((size_t)wts) & 15)) return NL_ERR_UNSUPPORTED;
  if (w2 && (npad <= 0 || npad > 192 || C <= 0 || C > npad || !wsum || !feat)) return NL_ERR_UNSUPPORTED;   // (two thread sets of npad in a workgroup of 384 / 512)
  const int wpr = S >> 5, nw = (8 % wpr) == 0 ? 8 : 6;
  const int ngroups = (int)nl_cdiv(N, 32 * nw);
  dim3 grid;
  if (chain_grid(ngroups, &grid) != NL_OK) return NL_ERR_HIP;   // one workgroup per CU (130 KB of LDS), XCD-aware when there are fewer groups than CUs
  if (nw == 8) hipLaunchKernelGGL(feat_comp_mx_kernel<8>, grid, dim3(64 * 8), 0, st, fa_frag, wts, (int)N, S, (const char*)bsh, (const char*)bmx, bias, hc, ngroups, w2, npad, C, wsum,
                                    feat, frag_f16 ? 1 : 0);
  else hipLaunchKernelGGL(feat_comp_mx_kernel<6>, grid, dim3(64 * 6), 0, st, fa_frag, wts, (int)N, S, (const char*)bsh, (const char*)bmx, bias, hc, ngroups, w2, npad, C, wsum, feat, frag_f16 ? 1 : 0);
  return hipPeekAtLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}
