// The five inner layers of the ray U-Net as ONE kernel (round 6; SURVEY.md §8 row a13, conditional_nerf/ray_unet.py:55-69):
//   conv2 (64 -> 128, k 3) + LayerNorm([128, S/2]) + ELU + MaxPool(2)      c1 (S/2 x 64)   -> c2 (S/4 x 128)
//   conv3 (128 -> 128, k 3) + LayerNorm([128, S/4]) + ELU + MaxPool(2)     c2              -> c3 (S/8 x 128)
//   trans_conv3 (128 -> 128, k 3, stride 2) + LayerNorm([128, S/4]) + ELU  c3              -> x0 (S/4 x 128)
//   trans_conv2 on cat[c2, x0] (256 -> 64) + LayerNorm([64, S/2]) + ELU                    -> x1 (S/2 x 64)
//   trans_conv1 on cat[c1, x1] (128 -> 32) + LayerNorm([32, S]) + ELU                      -> x2 (S x 32)
// for S = 128.  As five tgemm launches these are 7.3 M MAC per ray on 64 ... 512 workgroups each, bound by their own ramp, their
// LayerNorm reductions and the round trip of every slab through L2: 0.36 ms of a 7.6-ms config-2 step, 0.13 ms of a 1.19-ms 512-ray
// shard (profiles/r6_*).  Here a workgroup of eight waves takes a PAIR of rays through all five layers:
//   * the slabs never leave the CU: every activation lives in LDS as the split-bf16 B operand it will be read as — a hi and a lo plane
//     per (slab, ray), rows = positions with one zero row in front and behind (the k = 3 halo and the m + 1 tap of the transposed
//     layers need no masking), row stride 2 C + 16 bytes (conflict-free 16-byte fragment reads: 16 lanes cover the 64 banks once);
//   * D^T = W . X^T like tgemm.hip: the weights are the A operand.  A (row tile, column tile) pair belongs to ONE wave, so a weight
//     fragment is private to its wave and goes from L2 straight into registers (the streams nl_pack_weights wrote for tgemm_kernel are
//     read as they are: 1 KB per fragment, coalesced), four 32-k chunks ahead in a register ring; nothing is staged through LDS;
//   * every layer keeps all eight waves busy: 16 / 8 / 8 / 8 / 8 tiles of 32 x 32 for the pair (the 16-position trans_conv3 input
//     of TWO rays is exactly one row tile);
//   * LayerNorm over a ray's whole slab = two block reductions per layer (mean, then centred sum of squares: torch's two-pass
//     form), ELU, MaxPool over neighbouring lanes (DPP), bf16 hi / lo split and the store into the next layer's planes in the epilogue.
// Same products in the same order as the tgemm launches they replace (same weight streams, same three-term split, k ascending): the results
// differ only by the summation order of the LayerNorm statistics.
#include <utility>
#include "common.h"

typedef __bf16 ui_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ui_bf16x4 __attribute__((ext_vector_type(4)));
typedef float ui_f32x16 __attribute__((ext_vector_type(16)));
typedef float ui_f32x2 __attribute__((ext_vector_type(2)));
#ifndef UI_PK
#define UI_PK 1   // the epilogue's element-wise arithmetic on pairs (v_pk_add_f32 / v_pk_mul_f32: the same operations in the same order per element, half the instructions); 0 = scalar
#endif
// ELU of a pair (nl_elu_fast per element; the scale by log2(e) and the -1 packed)
__device__ __forceinline__ ui_f32x2 ui_elu2(ui_f32x2 x) {
  const ui_f32x2 y = x * ui_f32x2{1.4426950408889634f, 1.4426950408889634f};
  ui_f32x2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
  e = e - ui_f32x2{1.f, 1.f};
  return ui_f32x2{__builtin_amdgcn_fmed3f(x[0], e[0], 0.f), __builtin_amdgcn_fmed3f(x[1], e[1], 0.f)};
}

namespace {

template <int... Is, class F>
__device__ __forceinline__ void ui_static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void ui_static_for(F&& f) { ui_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

// ---- LDS map (bytes).  Geometry A: 64 channels, 64 positions (c1, x1); geometry B: 128 channels, 32 positions (c2, x0); c3: 128 channels, 16 positions.
constexpr int STR64 = 2 * 64 + 16, STR128 = 2 * 128 + 16;
constexpr int PL_C1 = 66 * STR64, PL_C2 = 34 * STR128, PL_C3 = 18 * STR128;   // one plane (hi or lo) of one ray, halo rows included
constexpr int OFF_C1 = 0;
constexpr int OFF_C2 = OFF_C1 + 4 * PL_C1;
constexpr int OFF_C3 = OFF_C2 + 4 * PL_C2;
constexpr int OFF_X0 = OFF_C3 + 4 * PL_C3;
constexpr int OFF_RED = OFF_X0 + 4 * PL_C2;
constexpr int OFF_X1 = OFF_C2;                      // x1 (geometry A) is written when c2 and c3 are dead: it lies over them
constexpr int LDS_BYTES = OFF_RED + 256;
static_assert(4 * PL_C1 <= 4 * PL_C2 + 4 * PL_C3, "x1 must fit over c2 | c3");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

enum { S_C1 = 0, S_C2, S_C3, S_X0, S_X1 };
__host__ __device__ constexpr int slab_off(int s) { return s == S_C1 ? OFF_C1 : s == S_C2 ? OFF_C2 : s == S_C3 ? OFF_C3 : s == S_X0 ? OFF_X0 : OFF_X1; }
__host__ __device__ constexpr int slab_str(int s) { return (s == S_C1 || s == S_X1) ? STR64 : STR128; }
__host__ __device__ constexpr int slab_pl(int s) { return (s == S_C1 || s == S_X1) ? PL_C1 : s == S_C3 ? PL_C3 : PL_C2; }

// ---- the five layers.  LI: rows of the product per ray (= input positions); NRT: 32-column tiles; NCH: 32-k chunks; chunk c reads
// channel block blk(c) of slab src(c) at position t + ioff(c); OUT: destination slab (-1: global x2); POOL / TRANS (merged phases:
// columns [0, N/2) are output position 2 t, [N/2, N) position 2 t + 1)
template <int ID> struct Lyr;
template <> struct Lyr<0> {   // conv2
  static constexpr int LI = 64, NRT = 4, NCH = 6, OUT = S_C2; static constexpr bool POOL = true, TRANS = false;
  static constexpr int src(int) { return S_C1; }
  static constexpr int blk(int c) { return c / 3; }
  static constexpr int ioff(int c) { return c % 3 - 1; }
};
template <> struct Lyr<1> {   // conv3
  static constexpr int LI = 32, NRT = 4, NCH = 12, OUT = S_C3; static constexpr bool POOL = true, TRANS = false;
  static constexpr int src(int) { return S_C2; }
  static constexpr int blk(int c) { return c / 3; }
  static constexpr int ioff(int c) { return c % 3 - 1; }
};
template <> struct Lyr<2> {   // trans_conv3, merged phases: K = [c3[m] | c3[m + 1]]
  static constexpr int LI = 16, NRT = 8, NCH = 8, OUT = S_X0; static constexpr bool POOL = false, TRANS = true;
  static constexpr int src(int) { return S_C3; }
  static constexpr int blk(int c) { return c % 4; }
  static constexpr int ioff(int c) { return c / 4; }
};
template <> struct Lyr<3> {   // trans_conv2 on cat[c2, x0]: K = [c2[m] | x0[m] | c2[m + 1] | x0[m + 1]]
  static constexpr int LI = 32, NRT = 4, NCH = 16, OUT = S_X1; static constexpr bool POOL = false, TRANS = true;
  static constexpr int src(int c) { return (c / 4) & 1 ? S_X0 : S_C2; }
  static constexpr int blk(int c) { return c % 4; }
  static constexpr int ioff(int c) { return c / 8; }
};
template <> struct Lyr<4> {   // trans_conv1 on cat[c1, x1]: K = [c1[m] | x1[m] | c1[m + 1] | x1[m + 1]]
  static constexpr int LI = 64, NRT = 2, NCH = 8, OUT = -1; static constexpr bool POOL = false, TRANS = true;
  static constexpr int src(int c) { return (c / 2) & 1 ? S_X1 : S_C1; }
  static constexpr int blk(int c) { return c % 2; }
  static constexpr int ioff(int c) { return c / 4; }
};

#ifndef UI_DEPTH
#define UI_DEPTH 4
#endif
#ifndef UI_BPIPE
#define UI_BPIPE 1      // the B fragments (LDS) of chunk c + 1 are read before the matrix instructions of chunk c
#endif
#ifndef UI_LNPRE
#define UI_LNPRE 1      // bias and LayerNorm tables of the layer are fetched before its product, not in its epilogue
#endif
constexpr int DEPTH = UI_DEPTH;   // weight chunks in flight per wave (16 registers each)
// ring slot of chunk c of layer ID: the chunks of all five layers form ONE sequence through the ring (a layer's last chunks refill their slots with the next
// layer's first chunks), so a layer starts at the phase the chunk counts before it leave
__host__ __device__ constexpr int nch_of(int id) { return id == 0 ? Lyr<0>::NCH : id == 1 ? Lyr<1>::NCH : id == 2 ? Lyr<2>::NCH : id == 3 ? Lyr<3>::NCH : Lyr<4>::NCH; }
__host__ __device__ constexpr int ring_phase(int id) { int p = 0; for (int i = 0; i < id; ++i) p += nch_of(i); return p % DEPTH; }

#ifdef UI_TRACE   // debug build (tools/unet_inner_trace.py): cycle counter of one workgroup's wave 0 at [0 start | 1 staged | per layer: 2 + 4 l: product done, + 1: after the first
// statistics barrier, + 2: after the second, + 3: epilogue done and the layer's last barrier passed]
__device__ unsigned long long ui_trace[32];
#define UI_T(i) do { if (pair == UI_TRACE && wave == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) ui_trace[(i)] = t_; } } while (0)
#else
#define UI_T(i)
#endif

template <bool X3>
__global__ __launch_bounds__(512, 1) void unet_inner_kernel(const NlUnetInnerArgs a, const int npairs) {
  __shared__ uint4 lds_all[LDS_BYTES / 16];
  char* const lds = reinterpret_cast<char*>(lds_all);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31;
  const int pair = (int)nl_xcd_block();
  if (pair >= npairs) return;
  const int ray0 = 2 * pair;
  const bool two = ray0 + 1 < a.R;   // the last pair of an odd batch holds one ray: the second one is rendered on zero rows and not stored
  UI_T(0);

  // ---- halo rows of every slab that is written once (x1's are zeroed when it is written: it lies over c2 | c3)
  {
    auto zero_halo = [&](int s, int rows) __attribute__((always_inline)) {
      const int n16 = slab_str(s) / 16;   // 16-byte units per row
      for (int i = tid; i < 4 * 2 * n16; i += 512) {
        const int sp = i / (2 * n16), r = (i / n16) & 1, u = i % n16;
        *reinterpret_cast<uint4*>(lds + slab_off(s) + sp * slab_pl(s) + (r ? rows + 1 : 0) * slab_str(s) + 16 * u) = make_uint4(0u, 0u, 0u, 0u);
      }
    };
    zero_halo(S_C1, 64); zero_halo(S_C2, 32); zero_halo(S_C3, 16); zero_halo(S_X0, 32);
  }
  // ---- stage c1: fp32 rows -> split-bf16 planes (coalesced 16-byte loads: 2 rays x 64 rows x 16 float4)
  for (int i = tid; i < 2 * 64 * 16; i += 512) {
    const int ray = i >> 10, row = (i >> 4) & 63, c4 = i & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ray == 0 || two) v = *reinterpret_cast<const float4*>(a.c1 + ((size_t)(ray0 + ray) * 64 + row) * 64 + 4 * c4);
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned hw[2], lw[2];
    nl_split_bf16_pair(f[0], f[1], hw[0], lw[0]); nl_split_bf16_pair(f[2], f[3], hw[1], lw[1]);
    const ui_bf16x4 h = __builtin_bit_cast(ui_bf16x4, make_uint2(hw[0], hw[1])), l = __builtin_bit_cast(ui_bf16x4, make_uint2(lw[0], lw[1]));
    char* dst = lds + OFF_C1 + (ray * 2) * PL_C1 + (row + 1) * STR64 + 8 * c4;
    *reinterpret_cast<ui_bf16x4*>(dst) = h;
    if (X3) *reinterpret_cast<ui_bf16x4*>(dst + PL_C1) = l;
  }

  float* const red = reinterpret_cast<float*>(lds + OFF_RED);   // [2 statistics][8 waves][2 rays]

  ui_bf16x8 wreg[DEPTH][4];   // [ring slot][hi ks0, hi ks1, lo ks0, lo ks1]
  // weight fragments of chunk c of layer LT for this wave's column tile
  auto load_w = [&](auto LT, int c, int ct, ui_bf16x8 (&w)[4]) __attribute__((always_inline)) {
    using L = Lyr<decltype(LT)::value>;
    const ui_bf16x8* src = reinterpret_cast<const ui_bf16x8*>(a.w[decltype(LT)::value]) + (size_t)c * (4 * L::NRT * 64) + ct * 64 + lane;
    w[0] = src[(0 * 2 + 0) * L::NRT * 64];
    w[1] = src[(0 * 2 + 1) * L::NRT * 64];
#ifdef UI_KO_LO      // knock-out (timing only, wrong results): half of the weight bytes are not fetched
    if (X3) { w[2] = w[0]; w[3] = w[1]; }
#else
    if (X3) { w[2] = src[(1 * 2 + 0) * L::NRT * 64]; w[3] = src[(1 * 2 + 1) * L::NRT * 64]; }
#endif
  };
  {   // the first layer's first chunks are on their way while c1 is staged
    const int ct0 = wave % Lyr<0>::NRT;
    ui_static_for<DEPTH>([&](auto C) __attribute__((always_inline)) { load_w(std::integral_constant<int, 0>{}, decltype(C)::value, ct0, wreg[decltype(C)::value]); });
  }
  __syncthreads();
  UI_T(1);

  ui_static_for<5>([&](auto LT) __attribute__((always_inline)) {
    constexpr int ID = decltype(LT)::value;
    using L = Lyr<ID>;
    constexpr int RTP = 2 * L::LI / 32;          // row tiles of the pair
    constexpr int TPW = RTP * L::NRT / 8;         // tiles per wave (same column tile, consecutive row tiles)
    static_assert(RTP * L::NRT % 8 == 0 && TPW >= 1, "tiles must split over eight waves");
    constexpr int N = 32 * L::NRT;
    const int ct = wave % L::NRT, rt0 = (wave / L::NRT) * TPW;
    // this lane's row in each of its tiles: pair-row p = 32 (rt0 + ti) + j -> ray p / LI, position p % LI
    int rayl[TPW], tl[TPW];
    unsigned rowbase[TPW];   // byte offset of (ray, hi plane, row t + 1, 8 hh channels) relative to the slab's offset (both slabs of a layer share their geometry)
    constexpr int s0 = L::src(0);
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int p = 32 * (rt0 + ti) + j;
      rayl[ti] = p / L::LI; tl[ti] = p % L::LI;
      rowbase[ti] = (unsigned)(rayl[ti] * 2 * slab_pl(s0) + (tl[ti] + 1) * slab_str(s0) + 16 * hh);
    }
    ui_f32x16 acc[TPW];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;

    // bias and LayerNorm tables of this wave's tiles: on their way before the product starts (L2 latency hides behind it)
    float4 bias4[4], gpre[TPW][4], bpre[TPW][4];
    if (UI_LNPRE) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) bias4[gq] = *reinterpret_cast<const float4*>(a.bias[ID] + 32 * ct + 8 * gq + 4 * hh);
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti) {
        const int wq = ((32 * (rt0 + ti)) % L::LI) / 32;
        const float* gp = a.gl[ID] + ((size_t)((wq * L::NRT + ct) * 4) * 64 + lane) * 4;
        const float* bp = a.bl[ID] + ((size_t)((wq * L::NRT + ct) * 4) * 64 + lane) * 4;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { gpre[ti][gq] = *reinterpret_cast<const float4*>(gp + gq * 256); bpre[ti][gq] = *reinterpret_cast<const float4*>(bp + gq * 256); }
      }
    }
    // ---- the product: chunk by chunk, the weight ring DEPTH chunks ahead, the B fragments one chunk ahead
    ui_bf16x8 bfr[2][2][TPW][2];   // [buffer][k-step][tile][hi | lo]
    auto load_b = [&](auto C, ui_bf16x8 (&dst)[2][TPW][2]) __attribute__((always_inline)) {
      constexpr int c = decltype(C)::value;
      constexpr int s = L::src(c);
      static_assert(slab_str(s) == slab_str(s0) && slab_pl(s) == slab_pl(s0), "the slabs of one layer share a geometry");
      constexpr int cbase = slab_off(s) + L::ioff(c) * slab_str(s) + 2 * 32 * L::blk(c);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
          const char* bp = lds + rowbase[ti] + (cbase + 32 * ks);
          dst[ks][ti][0] = *reinterpret_cast<const ui_bf16x8*>(bp);
          if (X3) dst[ks][ti][1] = *reinterpret_cast<const ui_bf16x8*>(bp + slab_pl(s));
        }
    };
    if (UI_BPIPE) load_b(std::integral_constant<int, 0>{}, bfr[0]);
    ui_static_for<L::NCH>([&](auto C) __attribute__((always_inline)) {
      constexpr int c = decltype(C)::value;
      constexpr int slot = (c + ring_phase(ID)) % DEPTH;
      const ui_bf16x8 (&w)[4] = wreg[slot];
      if constexpr (UI_BPIPE) { if constexpr (c + 1 < L::NCH) load_b(std::integral_constant<int, c + 1>{}, bfr[(c + 1) & 1]); }
      else load_b(C, bfr[c & 1]);
      const ui_bf16x8 (&b)[2][TPW][2] = bfr[c & 1];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
#ifndef UI_KO_MFMA   // knock-out (timing only): two of the three matrix instructions are not issued
          if (X3) {
            acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2 + ks], b[ks][ti][0], acc[ti], 0, 0, 0);
            acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], b[ks][ti][1], acc[ti], 0, 0, 0);
          }
#endif
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], b[ks][ti][0], acc[ti], 0, 0, 0);
        }
      }
      // refill the slot: this layer's chunk c + DEPTH, or the NEXT layer's first chunks (they arrive during the LayerNorm)
      if constexpr (c + DEPTH < L::NCH) load_w(LT, c + DEPTH, ct, wreg[slot]);
      else if constexpr (ID + 1 < 5) {
        using LN_ = Lyr<ID + 1>;
        constexpr int cn = c + DEPTH - L::NCH;
        static_assert(cn < LN_::NCH && (cn + ring_phase(ID + 1)) % DEPTH == slot, "ring phase");
        load_w(std::integral_constant<int, ID + 1>{}, cn, wave % LN_::NRT, wreg[slot]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });

    UI_T(2 + 4 * ID);
    // ---- bias, LayerNorm statistics of each ray's whole slab (two-pass), ELU (+ MaxPool), split, store
    float s1[2] = {0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      float s = 0.f;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 b4 = UI_LNPRE ? bias4[gq] : *reinterpret_cast<const float4*>(a.bias[ID] + 32 * ct + 8 * gq + 4 * hh);
        if constexpr (UI_PK) {
          const ui_f32x2 p0 = ui_f32x2{acc[ti][4 * gq + 0], acc[ti][4 * gq + 1]} + ui_f32x2{b4.x, b4.y};
          const ui_f32x2 p1 = ui_f32x2{acc[ti][4 * gq + 2], acc[ti][4 * gq + 3]} + ui_f32x2{b4.z, b4.w};
          acc[ti][4 * gq + 0] = p0[0]; acc[ti][4 * gq + 1] = p0[1]; acc[ti][4 * gq + 2] = p1[0]; acc[ti][4 * gq + 3] = p1[1];
          const ui_f32x2 q = p0 + p1;
          s += q[0] + q[1];
        } else {
        acc[ti][4 * gq + 0] += b4.x; acc[ti][4 * gq + 1] += b4.y; acc[ti][4 * gq + 2] += b4.z; acc[ti][4 * gq + 3] += b4.w;
        s += (acc[ti][4 * gq + 0] + acc[ti][4 * gq + 1]) + (acc[ti][4 * gq + 2] + acc[ti][4 * gq + 3]);
        }
      }
      s1[0] += rayl[ti] == 0 ? s : 0.f;
      s1[1] += rayl[ti] == 0 ? 0.f : s;
    }
    s1[0] = wave_sum(s1[0]); s1[1] = wave_sum(s1[1]);
    if (lane == 0) { red[wave * 2 + 0] = s1[0]; red[wave * 2 + 1] = s1[1]; }
    __syncthreads();   // (every wave is past its last read of this layer's inputs: the epilogue below may overwrite dead slabs)
    UI_T(3 + 4 * ID);
    float mean[2] = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) { mean[0] += red[w * 2 + 0]; mean[1] += red[w * 2 + 1]; }
    const float cnt = (float)(L::LI * N);
    mean[0] /= cnt; mean[1] /= cnt;
    float s2[2] = {0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const float m = rayl[ti] == 0 ? mean[0] : mean[1];
      float s = 0.f;
      if constexpr (UI_PK) {
        ui_f32x2 sp = {0.f, 0.f};
        const ui_f32x2 mm = {m, m};
#pragma unroll
        for (int r = 0; r < 16; r += 2) { const ui_f32x2 d = ui_f32x2{acc[ti][r], acc[ti][r + 1]} - mm; sp += d * d; }
        s = sp[0] + sp[1];
      } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[ti][r] - m; s += d * d; }
      }
      s2[0] += rayl[ti] == 0 ? s : 0.f;
      s2[1] += rayl[ti] == 0 ? 0.f : s;
    }
    s2[0] = wave_sum(s2[0]); s2[1] = wave_sum(s2[1]);
    if (lane == 0) { red[16 + wave * 2 + 0] = s2[0]; red[16 + wave * 2 + 1] = s2[1]; }
    __syncthreads();
    UI_T(4 + 4 * ID);
    float var[2] = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) { var[0] += red[16 + w * 2 + 0]; var[1] += red[16 + w * 2 + 1]; }
    const float rstd0 = 1.f / sqrtf(var[0] / cnt + a.eps), rstd1 = 1.f / sqrtf(var[1] / cnt + a.eps);

    if constexpr (L::OUT == S_X1) {   // x1's halo rows (its planes lie over c2 | c3, which every wave has finished reading)
      if (tid < 4 * 2 * (STR64 / 16)) {
        const int sp = tid / (2 * (STR64 / 16)), r = (tid / (STR64 / 16)) & 1, u = tid % (STR64 / 16);
        *reinterpret_cast<uint4*>(lds + OFF_X1 + sp * PL_C1 + (r ? 65 : 0) * STR64 + 16 * u) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const float m = rayl[ti] == 0 ? mean[0] : mean[1], rs = rayl[ti] == 0 ? rstd0 : rstd1;
      const int wq = ((32 * (rt0 + ti)) % L::LI) / 32;   // row tile inside the ray (the lane-major tables' first index)
      const float* gp = a.gl[ID] + ((size_t)((wq * L::NRT + ct) * 4) * 64 + lane) * 4;
      const float* bp = a.bl[ID] + ((size_t)((wq * L::NRT + ct) * 4) * 64 + lane) * 4;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 g4 = UI_LNPRE ? gpre[ti][gq] : *reinterpret_cast<const float4*>(gp + gq * 256);
        const float4 be4 = UI_LNPRE ? bpre[ti][gq] : *reinterpret_cast<const float4*>(bp + gq * 256);
        float v[4];
        if constexpr (UI_PK) {
          const ui_f32x2 mm = {m, m}, rr = {rs, rs};
          const ui_f32x2 p0 = ui_elu2((ui_f32x2{acc[ti][4 * gq + 0], acc[ti][4 * gq + 1]} - mm) * rr * ui_f32x2{g4.x, g4.y} + ui_f32x2{be4.x, be4.y});
          const ui_f32x2 p1 = ui_elu2((ui_f32x2{acc[ti][4 * gq + 2], acc[ti][4 * gq + 3]} - mm) * rr * ui_f32x2{g4.z, g4.w} + ui_f32x2{be4.z, be4.w});
          v[0] = p0[0]; v[1] = p0[1]; v[2] = p1[0]; v[3] = p1[1];
        } else {
        v[0] = nl_elu_fast((acc[ti][4 * gq + 0] - m) * rs * g4.x + be4.x);
        v[1] = nl_elu_fast((acc[ti][4 * gq + 1] - m) * rs * g4.y + be4.y);
        v[2] = nl_elu_fast((acc[ti][4 * gq + 2] - m) * rs * g4.z + be4.z);
        v[3] = nl_elu_fast((acc[ti][4 * gq + 3] - m) * rs * g4.w + be4.w);
        }
        if constexpr (L::POOL) {   // positions 2 p, 2 p + 1 are neighbouring lanes
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = nl_max_lane_xor1(v[e]);
        }
        const int n = 32 * ct + 8 * gq + 4 * hh;          // first of this lane's four output columns
        constexpr int CO = L::TRANS ? N / 2 : N;
        const int phase = L::TRANS ? (n >= CO ? 1 : 0) : 0;
        const int ch = n - phase * CO;
        const int pos = L::POOL ? (tl[ti] >> 1) : (L::TRANS ? 2 * tl[ti] + phase : tl[ti]);
        if constexpr (L::OUT < 0) {
          if (rayl[ti] == 0 || two)
            *reinterpret_cast<float4*>(a.x2 + ((size_t)(ray0 + rayl[ti]) * 128 + pos) * 32 + ch) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          if (!L::POOL || !(j & 1)) {
            unsigned hw[2], lw[2];
            nl_split_bf16_pair(v[0], v[1], hw[0], lw[0]); nl_split_bf16_pair(v[2], v[3], hw[1], lw[1]);
            const ui_bf16x4 h = __builtin_bit_cast(ui_bf16x4, make_uint2(hw[0], hw[1])), l = __builtin_bit_cast(ui_bf16x4, make_uint2(lw[0], lw[1]));
            constexpr int so = L::OUT >= 0 ? L::OUT : 0;
            char* dst = lds + slab_off(so) + (rayl[ti] * 2) * slab_pl(so) + (pos + 1) * slab_str(so) + 2 * ch;
            *reinterpret_cast<ui_bf16x4*>(dst) = h;
            if (X3) *reinterpret_cast<ui_bf16x4*>(dst + slab_pl(so)) = l;
          }
        }
      }
    }
    if constexpr (ID + 1 < 5) __syncthreads();   // the next layer's operands are in place; `red` may be written again
    UI_T(5 + 4 * ID);
  });
}

}  // namespace

#ifdef UI_TRACE
extern "C" __attribute__((visibility("default"))) int nl_debug_unet_inner_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ui_trace), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
#endif

bool nl_unet_inner_supported(int S, int precision) { return S == 128 && (precision == NL_PREC_BF16X3 || precision == NL_PREC_BF16); }

int nl_launch_unet_inner(const NlUnetInnerArgs& a, int precision, hipStream_t st) {
  if (a.R <= 0) return NL_OK;
  const int npairs = (a.R + 1) / 2;
  dim3 grid(nl_xcd_grid(npairs));
  if (precision == NL_PREC_BF16X3) hipLaunchKernelGGL(unet_inner_kernel<true>, grid, dim3(512), 0, st, a, npairs);
  else if (precision == NL_PREC_BF16) hipLaunchKernelGGL(unet_inner_kernel<false>, grid, dim3(512), 0, st, a, npairs);
  else return NL_ERR_UNSUPPORTED;
  NL_LAUNCH_CHECK();
  return NL_OK;
}
