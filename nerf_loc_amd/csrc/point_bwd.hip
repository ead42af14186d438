// Way back through the neural-point branch's per-(sample, neighbour) rows with frozen weights (pose refinement: pose_optimizer.py:131-160 differentiates the
// render w.r.t. the rays only) — the mirror image of point_fused2.hip:
//
//   d kv (N x 8, 256)  ->  . [Wk; Wv]           * LeakyReLU'(layer 3)      ibrnet.py:98-99, model.py:63-71
//                      ->  . base_mlp.4.weight  * LeakyReLU'(layer 2)
//                      ->  . base_mlp.2.weight  * LeakyReLU'(layer 1)
//                      ->  . base_mlp.0.weight[:, F:F+90]  =  d (posenc | ray_diff_fc) rows (N x 8, 96)      model.py:394-409
//
// The staged way back ran these as four streaming GEMMs, each reading and writing an (N x 8, W) fp32 matrix in HBM (2 KB per row and product: 4.5 GB per 512-ray
// step, 1.1 ms).  Here the chain is OUTPUT-STATIONARY like the forward kernel: a wave keeps 32 rows, the gradient of layer L (128 VGPRs as split-bf16 B fragments) and
// of layer L-1 (being produced) are register resident, the transposed weights stream through a 4-slot LDS ring by buffer LDS-DMA (one chunk = one 32-feature output
// tile x all K), the LeakyReLU masks are the sign BITS the forward left (tgemm's ep_maskin layout: 16 bytes per lane and layer), and the epilogue of tile rt - 1, the
// DMA pieces of the chunk three ahead and the next tile's input rows are issued in the MFMA shadow (`fill`).  Per row: 1 KB read, 384 B written.
// Arithmetic: three-term split-bf16 like every product of the way back (fp32's exponent range: gradients are small numbers).
#include <string.h>
#include <utility>
#include "common.h"

typedef __bf16 pb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pb_f32x16 __attribute__((ext_vector_type(16)));
typedef float pb_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int pb_u32x4 __attribute__((ext_vector_type(4)));

namespace {

#ifdef PB_TRACE
__device__ unsigned long long pb_trace[256];   // debug (tools/pb_trace.py): cycle counter at every region start of four tiles (block 0, wave 0)
#endif
constexpr int PB_NBUF = 4;   // LDS ring slots; chunk c lives in slot c % 4, chunks c+1 .. c+3 are in flight while c is consumed

template <int NRT>
struct BGeo {
  static constexpr int W = 32 * NRT;
  static constexpr int KS0 = 16;             // layer 0 of the way back: K = 256 (d k | d v)
  static constexpr int KSL = 2 * NRT;        // k-steps of the wide layers (K = W)
  static constexpr int KSM = KS0 > KSL ? KS0 : KSL;
  static constexpr int NC = 3 * NRT + 4;     // chunks (= output row tiles) per tile: L0' | L1' | L2' | the 96 encode columns (3 tiles) + one all-zero tile (NC % 4 == 0)
  static constexpr int SLOT = 2 * KSM * 64;  // ring slot in uint4 (hi and lo parts)
  static constexpr int cm(int g) { return ((g % NC) + NC) % NC; }
  static constexpr int layer(int g) { return cm(g) < NRT ? 0 : cm(g) < 2 * NRT ? 1 : cm(g) < 3 * NRT ? 2 : 3; }
  static constexpr int rt(int g) { return cm(g) < 3 * NRT ? cm(g) % NRT : cm(g) - 3 * NRT; }
  static constexpr int nks(int g) { return layer(g) == 0 ? KS0 : KSL; }
  static constexpr int ppw(int g) { return 2 * nks(g) / 4; }   // LDS-DMA pieces (1 KB) per wave
  static constexpr int gkb(int g) {                            // offset of chunk g in the global stream, in KB
    int o = 0;
    for (int i = 0; i < cm(g); ++i) o += 2 * nks(i);
    return o;
  }
  static constexpr int cumks(int g) {
    int o = 0;
    for (int i = 0; i < g; ++i) o += nks(i);
    return o;
  }
  static constexpr int STREAM_KB = 2 * (NRT * KS0 + (2 * NRT + 4) * KSL);
  // micro-steps of a finished chunk's epilogue: layers: 8 pairs x (mask + hi | lo); encode columns: 4 row stores; the zero tile: none
  static constexpr int epi_steps(int c) { return layer(c) < 3 ? 16 : (rt(c) < 3 ? 4 : 0); }
  static constexpr int RL = 4;   // A-fragment register ring (3 k-steps are live)
  static constexpr int rpos(int runks) { return runks % RL; }
  static_assert(NC % PB_NBUF == 0 && cumks(NC) % RL == 0, "ring positions must be tile-periodic");
};

template <int N>
__device__ __forceinline__ void pb_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
template <int... Is, class F>
__device__ __forceinline__ void pb_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void pb_static_for(F&& f) {
  pb_static_for_impl(std::make_integer_sequence<int, (N > 0 ? N : 0)>{}, static_cast<F&&>(f));
}
__device__ __forceinline__ unsigned pb_cvt_pk_bf16(float a, float b) {   // low half = bf16(a), high half = bf16(b), round to nearest even
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// lo word of a pair: bf16(v - float(hi))
__device__ __forceinline__ unsigned pb_lo2(float v0, float v1, unsigned hi) {
  unsigned lo; float t0, t1;
  asm("v_lshlrev_b32 %1, 16, %5\n\tv_and_b32 %2, 0xffff0000, %5\n\tv_sub_f32 %1, %3, %1\n\tv_sub_f32 %2, %4, %2\n\tv_cvt_pk_bf16_f32 %0, %1, %2"
      : "=&v"(lo), "=&v"(t0), "=&v"(t1) : "v"(v0), "v"(v1), "v"(hi));
  return lo;
}

struct PbArgs {
  const float* q; const float* kv; const float* go; float* gq;   // ATT: query (N, 128), kept k | v rows (NK, 256), d attention output (N, 128) -> d query (N, 128)
  unsigned q_bytes;
  const float* gkv;          // !ATT: (NK, 256) d k | d v rows
  const unsigned* mk[3];     // sign bits of the forward layers 1..3 ([32-row tile][lane][4 dwords])
  const uint4* wstream;
  float* gx;                 // (NK, 96)
  unsigned in_bytes, mk_bytes, out_bytes;
  int ntiles;
};

// ATT: the tile's d kv rows are not read but made — the attention's way back (ibrnet.py:89-108; backward.hip: attn_backward_kernel) runs in the prologue: lane (row,
// half) holds exactly the 16 dims per head that its B-operand k-slots want, the softmax over a sample's 8 neighbours is three DPP steps, d query is written from here.
template <int NRT, bool ATT>
__global__ __launch_bounds__(256, 1) void point_bwd_chain_kernel(const PbArgs a) {
  using GG = BGeo<NRT>;
  constexpr int NC = GG::NC, SLOT = GG::SLOT;
  __shared__ uint4 lds_all[PB_NBUF * SLOT];
  // (the ring's upper 64 KB through one opaque base whose offsets fit ds_read's immediate: point_fused2.hip)
  typedef __attribute__((address_space(3))) uint4 lds_u4;
  lds_u4* lds_hi = (lds_u4*)lds_all + 4096 + (threadIdx.x & 63);
  asm volatile("" : "+v"(lds_hi));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31;
  const unsigned nwg = gridDim.x;
  int tile = (int)nl_xcd_block();
  if (tile >= a.ntiles) return;

  // every access of the tile loop goes through buffer instructions (point_fused2.hip: exact vmcnt bookkeeping by the compiler, scalar piece offsets, out-of-range
  // offsets read zero / are dropped: rows past the end need no branch)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wstream, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc((void*)(ATT ? a.kv : a.gkv), 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void*)a.q, 0, ATT ? (int)a.q_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rGO = __builtin_amdgcn_make_buffer_rsrc((void*)a.go, 0, ATT ? (int)a.q_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rGQ = __builtin_amdgcn_make_buffer_rsrc((void*)a.gq, 0, ATT ? (int)a.q_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.mk[0], 0, (int)a.mk_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.mk[1], 0, (int)a.mk_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.mk[2], 0, (int)a.mk_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc((void*)a.gx, 0, (int)a.out_bytes, 0x00020000);
  const unsigned wvoff = wave * 1024 + lane * 16;   // piece p = 4 i + wave of a chunk
  uint4* lw = lds_all + wave * 64;
  unsigned soff = 0;                                // running stream offset of the next piece (scalar)
  auto dma_piece = [&](auto Cc, auto Ic) __attribute__((always_inline)) {
    constexpr int c = GG::cm(decltype(Cc)::value), i = decltype(Ic)::value;
    constexpr int want = GG::gkb(c) * 1024 + i * 4096;
    constexpr int prev = i == 0 ? GG::gkb(c - 1) * 1024 + (GG::ppw(c - 1) - 1) * 4096 : want - 4096;   // gkb wraps: chunk -1 = NC-1
    soff += (unsigned)(want - prev);
    asm volatile("" : "+s"(soff));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lw + (c % PB_NBUF) * SLOT + i * 256), 16, wvoff, soff, 0, 0);
  };

  // ---------------------------------------------------------------- register state
  // gradients as B fragments (hi / lo), ping-pong between layers; X[1] also takes the tile's input rows (free while the encode columns are multiplied out of X[0]).
  // Scalar dwords, assembled into a 4-dword operand at the MFMA (point_fused2.hip: vector-typed storage keeps both halves alive across the tile loop)
  unsigned Xh[2][GG::KSM][4], Xl[2][GG::KSM][4];
  pb_u32x4 frh[GG::RL], frl[GG::RL];   // A-fragment ring, position = (running k-step) % RL
  pb_f32x16 acc[2];                    // accumulator of chunk c = acc[c & 1]: region G accumulates one while the epilogue of G-1 drains the other
  pb_f32x4 raw[4][2];                  // !ATT: input rows in flight: k-step e of the next tile sits in raw[e & 3] until it is split three events later
  // ATT: this lane's 16 dims of a head: query, d output (per sample), k, v (per row) — two sets: head h + 1 is in flight (a whole region ahead: the loads are lane = row
  // gathers with HBM latency; issued half a region ahead they cost 6 000 idle cycles per head) while head h is worked on.  (The allocator finds the second set in the
  // k-steps of X[1] that the later heads have not written yet.)
  pb_f32x4 qv[2][4], gov[2][4], kk[2][4], vv[2][4];
  float asc = 0.f, agp = 0.f, amx = 0.f, aee = 0.f, aat = 0.f, ags = 0.f, agsq = 0.f;
  unsigned qoff = 0, gqoff = 0, pn_qoff = 0, pn_gqoff = 0;
  pb_u32x4 mkw[3];                     // the tile's sign bits of forward layers 3, 2, 1 (= way-back layers 0, 1, 2)
  float ev0 = 0.f, ev1 = 0.f;
  unsigned ehi = 0;
  unsigned inoff = 0, pn_inoff = 0, outoff = 0;
  int pn_tile = tile;
  const pb_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto mfma = [](const pb_u32x4& x, const pb_u32x4& y, const pb_f32x16& c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pb_bf16x8, x), __builtin_bit_cast(pb_bf16x8, y), c, 0, 0, 0);
  };
  auto frag4 = [](const unsigned (&d)[4]) __attribute__((always_inline)) { return pb_u32x4{d[0], d[1], d[2], d[3]}; };

  // ---------------------------------------------------------------- tile prologue: the tile's 128 d kv rows -> X[1]
  // Lane (row j, half hh) holds k-slots 8 hh .. 8 hh + 7 of every k-step: columns 16 ks + 8 hh .. + 7 of its row, two 16-byte loads.  19 events: event e loads
  // k-step e (e < 16) and splits k-step e - 3 (e >= 3); the next tile's events are spread over the four encode-column regions, the first tile runs them back to back.
  constexpr int EV = 19;
  auto pro_event = [&](auto Ec, unsigned off) __attribute__((always_inline)) {
    constexpr int e = decltype(Ec)::value;
    if constexpr (e >= 3) {
      constexpr int k = e - 3;
      pb_static_for<4>([&](auto Dc) __attribute__((always_inline)) {
        constexpr int d = decltype(Dc)::value;
        const float v0 = raw[k & 3][d >> 1][2 * (d & 1)], v1 = raw[k & 3][d >> 1][2 * (d & 1) + 1];
        const unsigned h = pb_cvt_pk_bf16(v0, v1);
        Xh[1][k][d] = h;
        Xl[1][k][d] = pb_lo2(v0, v1, h);
      });
    }
    if constexpr (e < 16) {
      raw[e & 3][0] = __builtin_bit_cast(pb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rIn, off + 64u * e, 0, 0));
      raw[e & 3][1] = __builtin_bit_cast(pb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rIn, off + 64u * e + 16u, 0, 0));
    }
  };
  // ATT: 13 events per head h (= encode-column region h): 0 the 16 loads | 1-4 scores, softmax, its way back | 5-8 d k / d v -> k-steps 2 h + u / 8 + 2 h + u of X[1] |
  // 9-12 d query (summed over the sample's 8 rows) -> HBM
  // load l (0..15) of head h.  One at a time between the MFMAs: a burst of lane = row gathers parks the wave at issue
  // (the CU's address unit takes ~32 cycles per k / v instruction and its four waves issue in step) — 16 back to back cost 5 000 idle cycles per head
  auto att_load = [&](auto Hc, auto Lc, unsigned qo, unsigned ko) __attribute__((always_inline)) {
    // order: k quads 0..3 | v quads 0..3 | query | d output — the four quads of a head's k (v) are the four 32-byte quarters of the SAME 128-byte line of every row: issued
    // next to each other the line is fetched once (32 KB of vector L1 hold one head's k + v lines of the CU's four waves, nothing more)
    constexpr int h = decltype(Hc)::value, l = decltype(Lc)::value, i = l & 3, w = l < 4 ? 2 : l < 8 ? 3 : l < 12 ? 0 : 1;
    constexpr unsigned o = 128u * h + 64u * (i >> 1) + 16u * (i & 1);
    if constexpr (w == 0) qv[h & 1][i] = __builtin_bit_cast(pb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rQ, qo + o, 0, 0));
    else if constexpr (w == 1) gov[h & 1][i] = __builtin_bit_cast(pb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rGO, qo + o, 0, 0));
    else if constexpr (w == 2) kk[h & 1][i] = __builtin_bit_cast(pb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rIn, ko + o, 0, 0));
    else vv[h & 1][i] = __builtin_bit_cast(pb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rIn, ko + 512u + o, 0, 0));
  };
  auto att_event = [&](auto Hc, auto Ec, unsigned qo, unsigned ko, unsigned gqo) __attribute__((always_inline)) {
    constexpr int h = decltype(Hc)::value, e = decltype(Ec)::value;
    constexpr float itemp = 1.0f / 5.656854249492381f;   // temperature sqrt(d_k) (ibrnet.py:84)
    if constexpr (e == 0) {   // (the first tile: all 16 at once)
      pb_static_for<16>([&](auto Lc) __attribute__((always_inline)) { att_load(Hc, Lc, qo, ko); });
    } else if constexpr (e == 1) {
      float x = 0.f, y = 0.f;
      pb_static_for<16>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int i = decltype(Ic)::value >> 2, c = decltype(Ic)::value & 3;
        x = fmaf(qv[h & 1][i][c], kk[h & 1][i][c], x); y = fmaf(gov[h & 1][i][c], vv[h & 1][i][c], y);
      });
      asc = x; agp = y;
    } else if constexpr (e == 2) {
      asc += __shfl_xor(asc, 32, 64); agp += __shfl_xor(agp, 32, 64);
      asc *= itemp;
      amx = nl_max8(asc);
    } else if constexpr (e == 3) {
      aee = expf(asc - amx);
      aat = aee / nl_sum8(aee);
    } else if constexpr (e == 4) {
      ags = aat * (agp - nl_sum8(aat * agp));   // softmax backward -> d score
      agsq = ags * itemp;
    } else if constexpr (e < 9) {
      constexpr int kind = (e - 5) >> 1, u = (e - 5) & 1, ks = (kind ? 8 : 0) + 2 * h + u;
      pb_static_for<4>([&](auto Dc) __attribute__((always_inline)) {
        constexpr int d = decltype(Dc)::value, i = 2 * u + (d >> 1), c0 = 2 * (d & 1);
        const float v0 = kind ? aat * gov[h & 1][i][c0] : agsq * qv[h & 1][i][c0], v1 = kind ? aat * gov[h & 1][i][c0 + 1] : agsq * qv[h & 1][i][c0 + 1];
        const unsigned hw = pb_cvt_pk_bf16(v0, v1);
        Xh[1][ks][d] = hw;
        Xl[1][ks][d] = pb_lo2(v0, v1, hw);
      });
    } else {
      constexpr int i = e - 9;
      const float x0 = nl_sum8(ags * kk[h & 1][i][0]) * itemp, x1 = nl_sum8(ags * kk[h & 1][i][1]) * itemp, x2 = nl_sum8(ags * kk[h & 1][i][2]) * itemp, x3 = nl_sum8(ags * kk[h & 1][i][3]) * itemp;
      __builtin_amdgcn_raw_buffer_store_b128(pb_u32x4{__float_as_uint(x0), __float_as_uint(x1), __float_as_uint(x2), __float_as_uint(x3)}, rGQ,
                                             gqo + 128u * h + 64u * (i >> 1) + 16u * (i & 1), 0, 0);
    }
  };
  auto row_off = [&](int t) __attribute__((always_inline)) { return (unsigned)(t * 128 + wave * 32 + j); };
  auto set_att_off = [&](int t, unsigned& io, unsigned& qo, unsigned& gqo) __attribute__((always_inline)) {
    const unsigned r = row_off(t);
    io = r * 1024u + 32u * hh;
    qo = (r >> 3) * 512u + 32u * hh;               // the row's sample
    gqo = (r & 7u) ? 0x80000000u : qo;             // the sample's first row stores d query (the others: out of range = dropped)
  };
  auto load_mask = [&](auto Lc, int t) __attribute__((always_inline)) {   // way-back layer L reads the sign bits of forward layer 3 - L
    constexpr int L = decltype(Lc)::value;
    mkw[L] = __builtin_bit_cast(pb_u32x4, __builtin_amdgcn_raw_buffer_load_b128(L == 0 ? rM2 : L == 1 ? rM1 : rM0, (unsigned)(t * 4 + wave) * 1024u + lane * 16u, 0, 0));
  };

  // A fragments of k-step t of chunk G (t >= nks(G): k-step t - nks(G) of chunk G+1) -> ring position of that running k-step
  auto read_frag = [&](auto Gc, auto Tc, auto Pc) __attribute__((always_inline)) {
    constexpr int G = decltype(Gc)::value, t = decltype(Tc)::value, part = decltype(Pc)::value;
    constexpr int c = t >= GG::nks(G) ? GG::cm(G + 1) : GG::cm(G), ks = t >= GG::nks(G) ? t - GG::nks(G) : t;
    constexpr int pos = GG::rpos(GG::cumks(GG::cm(G)) + t);
    constexpr int li = (c % PB_NBUF) * SLOT + (part * GG::nks(c) + ks) * 64;
    pb_u32x4 v;
    if constexpr (li >= 4096) v = __builtin_bit_cast(pb_u32x4, lds_hi[li - 4096]);
    else v = __builtin_bit_cast(pb_u32x4, lds_all[li + lane]);
    if (part == 0) frh[pos] = v; else frl[pos] = v;
  };

  // ---------------------------------------------------------------- epilogue micro-steps of chunk C, run inside region C+1
  auto epi_step = [&](auto Cc, auto Ec) __attribute__((always_inline)) {
    constexpr int C = GG::cm(decltype(Cc)::value), E = decltype(Ec)::value;
    constexpr int L = GG::layer(C), RT = GG::rt(C), AB = C & 1;
    if constexpr (L < 3) {
      constexpr int p = E / 2, sub = E % 2;
      constexpr int out = L & 1;   // L0' -> X[0], L1' -> X[1], L2' -> X[0]
      constexpr int fo = 2 * RT + (p >> 2), d = p & 3;
      if constexpr (sub == 0) {   // LeakyReLU' from the forward's sign bits: bit 16 (RT & 1) + r of dword RT >> 1 <-> accumulator register r of row tile RT
        const unsigned w = mkw[L][RT >> 1] >> (16 * (RT & 1) + 2 * p);
        ev0 = acc[AB][2 * p] * ((w & 1u) ? 1.f : 0.01f);
        ev1 = acc[AB][2 * p + 1] * ((w & 2u) ? 1.f : 0.01f);
        ehi = pb_cvt_pk_bf16(ev0, ev1);
        Xh[out][fo][d] = ehi;
      } else Xl[out][fo][d] = pb_lo2(ev0, ev1, ehi);
    } else if constexpr (RT < 3) {   // encode columns 32 RT + 8 g + 4 hh .. + 3 of this lane's row
      constexpr int g = E;
      const float x0 = acc[AB][4 * g], x1 = acc[AB][4 * g + 1], x2 = acc[AB][4 * g + 2], x3 = acc[AB][4 * g + 3];
      __builtin_amdgcn_raw_buffer_store_b128(pb_u32x4{__float_as_uint(x0), __float_as_uint(x1), __float_as_uint(x2), __float_as_uint(x3)}, rOut,
                                             outoff + (unsigned)(32 * RT + 8 * g + 4 * hh) * 4u, 0, 0);
    }
  };

  // everything that is issued in the shadow of MFMA slot K of region G
  auto fill = [&](auto Gc, auto Kc) __attribute__((always_inline)) {
    constexpr int G = decltype(Gc)::value, K = decltype(Kc)::value;
    constexpr int NKS = GG::nks(G), NS = 3 * NKS, NSD = 3 * (NKS - 1);
    // LDS-DMA pieces of chunk G+3 (its slot held chunk G-1, which every wave left behind at the previous barrier)
    if constexpr (K < NSD) {
      constexpr int ND = GG::ppw(G + 3), d0 = K * ND / NSD, d1 = (K + 1) * ND / NSD;
      pb_static_for<d1 - d0>([&](auto Ic) __attribute__((always_inline)) { dma_piece(std::integral_constant<int, G + 3>{}, std::integral_constant<int, d0 + decltype(Ic)::value>{}); });
    }
    // sign bits: two regions before their layer's first epilogue
    if constexpr (K == 0 && G == NC - 2) load_mask(std::integral_constant<int, 0>{}, pn_tile);
    if constexpr (K == 0 && G == NRT - 2) load_mask(std::integral_constant<int, 1>{}, tile);
    if constexpr (K == 0 && G == 2 * NRT - 2) load_mask(std::integral_constant<int, 2>{}, tile);
    // the next tile's input rows, under the encode-column regions
    if constexpr (G >= 3 * NRT && !ATT) {
      if constexpr (K == 0 && G == 3 * NRT) pn_inoff = row_off(pn_tile) * 1024u + 32u * hh;
      constexpr int s = (G - 3 * NRT) * NS + K, T = 4 * NS, e0 = s * EV / T, e1 = (s + 1) * EV / T;
      pb_static_for<e1 - e0>([&](auto Ic) __attribute__((always_inline)) { pro_event(std::integral_constant<int, e0 + decltype(Ic)::value>{}, pn_inoff); });
    }
    if constexpr (ATT) {   // head h is worked on under encode-column region h; its 16 loads went out one by one under the region before
      if constexpr (G >= 3 * NRT - 1 && G < 3 * NRT + 3) {
        constexpr int hl = G - (3 * NRT - 1);   // the head whose loads this region issues
        if constexpr (K == 0 && hl == 0) set_att_off(pn_tile, pn_inoff, pn_qoff, pn_gqoff);
        // (in the region's FIRST half: the arithmetic of this head starts half a region into the next one — a whole region of latency cover for the last load)
        if constexpr (K < NS / 2) {
          constexpr int l0 = K * 16 / (NS / 2), l1 = (K + 1) * 16 / (NS / 2);
          pb_static_for<l1 - l0>([&](auto Ic) __attribute__((always_inline)) {
            att_load(std::integral_constant<int, hl>{}, std::integral_constant<int, l0 + decltype(Ic)::value>{}, pn_qoff, pn_inoff);
          });
        }
      }
      if constexpr (G >= 3 * NRT) {
        constexpr int h = G - 3 * NRT, S0 = NS / 2, H2 = NS - S0;
        if constexpr (K >= S0) {
          constexpr int e0 = 1 + (K - S0) * 12 / H2, e1 = 1 + (K - S0 + 1) * 12 / H2;
          pb_static_for<e1 - e0>([&](auto Ic) __attribute__((always_inline)) {
            att_event(std::integral_constant<int, h>{}, std::integral_constant<int, e0 + decltype(Ic)::value>{}, pn_qoff, pn_inoff, pn_gqoff);
          });
        }
      }
    }
    // epilogue of the previous chunk (a layer's last row tile is finished inside the first region of the NEXT layer, which consumes the fragments it produces
    // in its last two k-steps: layer epilogues end one k-step early)
    {
      constexpr int NE = GG::epi_steps(G - 1), NSE = GG::layer(G - 1) < 3 ? NSD : NS;
      if constexpr (NE > 0 && K < NSE) {
        constexpr int e0 = K * NE / NSE, e1 = (K + 1) * NE / NSE;
        pb_static_for<e1 - e0>([&](auto Ec) __attribute__((always_inline)) { epi_step(std::integral_constant<int, G - 1>{}, std::integral_constant<int, e0 + decltype(Ec)::value>{}); });
      }
    }
  };

  // ---------------------------------------------------------------- one region = one output row tile accumulated over all K
  int trace_it = 0;
  (void)trace_it;
  auto region = [&](auto Gc) __attribute__((always_inline)) {
    constexpr int G = decltype(Gc)::value;
#ifdef PB_TRACE
    if (blockIdx.x == 0 && wave == 0 && trace_it < 4) {
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) pb_trace[trace_it * 64 + G] = t;
    }
#endif
    constexpr int L = GG::layer(G), NKS = GG::nks(G), AB = G & 1, CK = GG::cumks(G);
    pb_static_for<NKS>([&](auto Kc) __attribute__((always_inline)) {
      constexpr int ks = decltype(Kc)::value, pos = GG::rpos(CK + ks);
      if constexpr (ks == NKS - 1) {
        // chunk G+1 must have landed (only the pieces of G+2, G+3 may still fly) and every wave must be through with chunk G's slot reads
        pb_wait_vmcnt<GG::ppw(G + 2) + GG::ppw(G + 3)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      const pb_u32x4 bh = frag4(Xh[(L + 1) & 1][ks]), bl = frag4(Xl[(L + 1) & 1][ks]);
      pb_static_for<3>([&](auto Mc) __attribute__((always_inline)) {
        constexpr int m = decltype(Mc)::value, K = 3 * ks + m;
        // A fragments two k-steps ahead; the first two of the next chunk wait for the barrier of the last group
        if constexpr (ks + 2 < NKS) {
          if constexpr (m == 0) read_frag(Gc, std::integral_constant<int, ks + 2>{}, std::integral_constant<int, 0>{});
          if constexpr (m == 1) read_frag(Gc, std::integral_constant<int, ks + 2>{}, std::integral_constant<int, 1>{});
        } else if constexpr (ks == NKS - 1) {
          if constexpr (m == 0) {
            read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 0>{});
            read_frag(Gc, std::integral_constant<int, NKS>{}, std::integral_constant<int, 1>{});
          }
          if constexpr (m == 1) {
            read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 0>{});
            read_frag(Gc, std::integral_constant<int, NKS + 1>{}, std::integral_constant<int, 1>{});
          }
        }
        if constexpr (m == 0) acc[AB] = mfma(frl[pos], bh, ks == 0 ? zero16 : acc[AB]);
        else if constexpr (m == 1) acc[AB] = mfma(frh[pos], bl, acc[AB]);
        else acc[AB] = mfma(frh[pos], bh, acc[AB]);
        fill(Gc, std::integral_constant<int, K>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---------------------------------------------------------------- pipeline start
  soff = (unsigned)(GG::gkb(NC - 1) * 1024 + (GG::ppw(NC - 1) - 1) * 4096);   // the "previous piece" of the very first one
  pb_static_for<3>([&](auto Cc) __attribute__((always_inline)) {
    pb_static_for<GG::ppw(decltype(Cc)::value)>([&](auto Ic) __attribute__((always_inline)) { dma_piece(Cc, Ic); });
  });
  load_mask(std::integral_constant<int, 0>{}, tile);
  if constexpr (ATT) {
    set_att_off(tile, inoff, qoff, gqoff);
    pb_static_for<4>([&](auto Hc) __attribute__((always_inline)) {
      pb_static_for<13>([&](auto Ec) __attribute__((always_inline)) { att_event(Hc, Ec, qoff, inoff, gqoff); });
    });
  } else {
    inoff = row_off(tile) * 1024u + 32u * hh;
    pb_static_for<EV>([&](auto Ec) __attribute__((always_inline)) { pro_event(Ec, inoff); });
  }
  pb_wait_vmcnt<GG::ppw(1) + GG::ppw(2)>();   // conservative: the prologue's own loads are younger than every piece of chunk 0
  __builtin_amdgcn_s_barrier();
  read_frag(std::integral_constant<int, NC - 1>{}, std::integral_constant<int, GG::nks(NC - 1)>{}, std::integral_constant<int, 0>{});
  read_frag(std::integral_constant<int, NC - 1>{}, std::integral_constant<int, GG::nks(NC - 1) + 1>{}, std::integral_constant<int, 0>{});
  read_frag(std::integral_constant<int, NC - 1>{}, std::integral_constant<int, GG::nks(NC - 1)>{}, std::integral_constant<int, 1>{});
  read_frag(std::integral_constant<int, NC - 1>{}, std::integral_constant<int, GG::nks(NC - 1) + 1>{}, std::integral_constant<int, 1>{});

  for (;;) {
    pn_tile = tile + (int)nwg;   // rows past the end read zero: the last tile prepares a tile that is never computed
    outoff = row_off(tile) * 384u;
    pb_static_for<NC>(region);
#ifdef PB_TRACE
    if (blockIdx.x == 0 && wave == 0 && trace_it < 4) {
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) pb_trace[trace_it * 64 + NC] = t;
    }
    ++trace_it;
#endif
    tile = pn_tile;
    if (tile >= a.ntiles) break;
  }
  pb_wait_vmcnt<0>();   // LDS-DMA prefetched for a tile that does not exist must land before the LDS is handed to another workgroup
}

__device__ __forceinline__ unsigned short pb_f2bf(float x) {
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ int pb_m(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }   // accumulator register -> row of the 32x32 tile

// Stream: chunk (layer, rt) = [part hi/lo][k-step][lane][8 bf16] in A-fragment order (lane: out row 32 rt + (lane & 31), k-slots 8 (lane >> 5) + t).
// Layer 0's K runs over the d kv columns in natural order (the prologue loads them so); layers 1..3 take K in ACCUMULATOR order of the layer before
// (k-step 2 rt' + u, slot t of half hh <-> feature 32 rt' + m(8 u + t, hh)), so that a finished accumulator IS the next B operand.
__global__ void pack_point_bwd_stream_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                             const float* __restrict__ wk, const float* __restrict__ wv, unsigned short* __restrict__ out, int NRT, int F) {
  const int W = 32 * NRT, KSL = 2 * NRT;
  const long long n0 = (long long)NRT * 16 * 512, n1 = (long long)NRT * KSL * 512, n3 = (long long)4 * KSL * 512;
  const long long total = n0 + 2 * n1 + n3;   // (chunk, k-step, lane, t) elements of one part
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int layer; long long r = e;
  if (r < n0) layer = 0; else if ((r -= n0) < n1) layer = 1; else if ((r -= n1) < n1) layer = 2; else { r -= n1; layer = 3; }
  const int nks = layer == 0 ? 16 : KSL;
  const int t = (int)(r & 7), lane = (int)((r >> 3) & 63);
  const long long r2 = r >> 9;
  const int ks = (int)(r2 % nks), rt = (int)(r2 / nks);
  const int hh = lane >> 5, orow = 32 * rt + (lane & 31);
  float v = 0.f;
  if (layer == 0) {   // d H3[f] = sum_c d kv[c] . [Wk; Wv][c][f]
    const int c = 16 * ks + 8 * hh + t;
    v = c < 128 ? wk[(size_t)c * W + orow] : wv[(size_t)(c - 128) * W + orow];
  } else {
    const int fin = 32 * (ks >> 1) + pb_m(8 * (ks & 1) + t, hh);
    if (layer == 1) v = w3[(size_t)fin * W + orow];            // base_mlp.4.weight (H3 <- H2)
    else if (layer == 2) v = w2[(size_t)fin * W + orow];       // base_mlp.2.weight (H2 <- H1)
    else v = orow < 90 ? w1[(size_t)fin * (F + 90) + F + orow] : 0.f;   // base_mlp.0.weight's posenc | ray_diff_fc columns; 6 + 32 zero rows
  }
  long long base;   // chunk base in bf16 elements (a chunk holds 2 parts x nks x 512)
  if (layer == 0) base = (long long)rt * 2 * 16 * 512;
  else base = (long long)NRT * 2 * 16 * 512 + ((long long)(layer - 1) * NRT + rt) * 2 * KSL * 512;
  const long long in_part = ((long long)ks * 64 + lane) * 8 + t;
  const unsigned short h = pb_f2bf(v);
  out[base + in_part] = h;
  out[base + (long long)nks * 512 + in_part] = pb_f2bf(v - __uint_as_float(((unsigned int)h) << 16));
}


}  // namespace

#ifdef PB_TRACE
extern "C" __attribute__((visibility("default"))) int nl_debug_pb_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pb_trace), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1;
}
#endif
bool nl_point_bwd_chain_supported(int W) { return W == 128 || W == 256; }
size_t nl_point_bwd_stream_bytes(int W) {
  const int NRT = W / 32;
  return (size_t)2 * (NRT * 16 + (2 * NRT + 4) * 2 * NRT) * 1024 + 4096;
}
// w1: base_mlp.0.weight (W, F + 90); w2: base_mlp.2.weight; w3: base_mlp.4.weight; wk / wv: the k / v projections (128, W)
int nl_pack_point_bwd_stream(const float* w1, const float* w2, const float* w3, const float* wk, const float* wv, void* out, int W, int F, hipStream_t st) {
  if (!nl_point_bwd_chain_supported(W)) return NL_ERR_UNSUPPORTED;
  const int NRT = W / 32;
  const long long total = ((long long)NRT * 16 + (2LL * NRT + 4) * 2 * NRT) * 512;
  hipLaunchKernelGGL(pack_point_bwd_stream_kernel, dim3((unsigned)nl_cdiv(total, 256)), dim3(256), 0, st, w1, w2, w3, wk, wv, (unsigned short*)out, NRT, F);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

// gkv (NK, 256) -> gx (NK, 96): the four transposed products of the branch's rows with the forward's sign bits mk[0..2] (forward layers 1..3) in between.
// att (gkv == null): the attention's way back in front — q (N, 128), kv (NK = 8 N, 256), go (N, 128) -> gq (N, 128) and the d kv rows, which never leave the registers
int nl_launch_point_bwd_chain(const float* gkv, const unsigned* const* mk, const void* wstream, float* gx, int64_t NK, int W, hipStream_t st, const float* q,
                              const float* kv, const float* go, float* gq) {
  if (NK <= 0) return NL_OK;
  if (!nl_point_bwd_chain_supported(W) || NK * 1024 > 0x7fffffffll) return NL_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  const int g_pb_num_cu = nl_persistent_cus();
  if (g_pb_num_cu < 0) return g_pb_num_cu;
  PbArgs a;
  memset(&a, 0, sizeof(a));
  const bool att = gkv == nullptr;
  if (att && (!q || !kv || !go || !gq || (NK & 7))) return NL_ERR_BAD_ARG;
  a.q = q; a.kv = kv; a.go = go; a.gq = gq; a.q_bytes = (unsigned)(NK / 8 * 512);
  a.gkv = gkv; a.mk[0] = mk[0]; a.mk[1] = mk[1]; a.mk[2] = mk[2]; a.wstream = (const uint4*)wstream; a.gx = gx;
  a.in_bytes = (unsigned)(NK * 1024); a.mk_bytes = (unsigned)(nl_cdiv(NK, 32) * 1024); a.out_bytes = (unsigned)(NK * 384);
  a.ntiles = (int)nl_cdiv(NK, 128);
  const int nwg = a.ntiles < g_pb_num_cu ? (int)nl_xcd_grid(a.ntiles) : g_pb_num_cu;
  if (W == 256) {
    if (att) hipLaunchKernelGGL((point_bwd_chain_kernel<8, true>), dim3(nwg), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((point_bwd_chain_kernel<8, false>), dim3(nwg), dim3(256), 0, st, a);
  } else {
    if (att) hipLaunchKernelGGL((point_bwd_chain_kernel<4, true>), dim3(nwg), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((point_bwd_chain_kernel<4, false>), dim3(nwg), dim3(256), 0, st, a);
  }
  NL_LAUNCH_CHECK();
  return NL_OK;
}
