// Multi-view front end of the fused render path, round 4 (SURVEY.md §8 rows a4, a7 — VERDICT r3 item 1): the visibility-weighted statistics of the
// multi-view taps AND out_fc.0 (393 -> 64, ELU) in one kernel, for the reference's feature width C = 192.
//
//   reference arithmetic: ibrnet.py:169-231 (projection, zeros / align_corners = True taps of rgb + features), multiview_aggregator.py:199-221
//   (weights = vis / (sum vis + 1e-8), weighted mean / variance over the views, out_fc.0 + ELU).
//
// What mv_stats8_kernel (mvagg.hip) waits on is its taps: every (sample, view) fetches four 768-byte texel rows although consecutive samples of a ray
// land in the same texel cell of a support view ~90 % of the time (the feature maps have a quarter of the image resolution), and then it writes the 416-wide
// statistics row that a separate GEMM launch reads back (1.7 GB per config-2 batch).  Here
//   * a wave walks FOUR CONSECUTIVE SAMPLES, view by view, with lanes = channels (3 per lane): the four texel rows of the current cell stay in 12 registers
//     and are re-fetched only when the cell changes (a wave-uniform compare of the packed cell word) — 0.3 instead of 1 row set per (sample, view);
//   * the statistics are accumulated in one pass (sum w x, sum w x^2: 24 registers for the four samples) and leave the lane as bf16 hi / lo halves of a
//     32-sample staging tile in LDS — the B operand of out_fc.0, whose A fragments (64 x 384, split-bf16) live in REGISTERS (96 per wave: each of the eight
//     waves of the persistent workgroup owns one 16-row slice of the layer and one half of the samples), so no weight is streamed per tile;
//   * the nine remaining statistics (colour mean / variance, depth-difference mean / variance, mean weight) and the bias are nine FMAs per output on the
//     vector unit (lane = output unit) and enter the MFMA as the accumulator's initial value;
//   * the per-(sample, view) scalar work (projection, tap weights, colour taps, weight normalisation) runs with lane = (sample, view), 64 pairs per wave.
// Output: the 64-wide hidden rows t64 (N x 64), the valid flags and the tapped colours + visibility (N x V x 4) for the blend kernel — the statistics row
// and the blend layer's per-(sample, view) rows (bl1: 0.7 GB written and read back per batch) no longer exist on this path: blend_taps_kernel (heads.hip)
// recomputes its taps.
#include "mvdec.h"

namespace {
using namespace nlmv;

typedef __bf16 mf_bf16x8 __attribute__((ext_vector_type(8)));
typedef float mf_f32x4 __attribute__((ext_vector_type(4)));
typedef float mf_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned mf_u32x4 __attribute__((ext_vector_type(4)));

#ifndef MF_KO
#define MF_KO 0   // knock-out bits for timing experiments (results are garbage): 1 no texel loads in phase B, 2 no phase B, 4 no colour taps in phase A, 8 no MFMA phase, 16 no staging writes, 64 every texel fetch from texel 0 (all cache hits), 128 phase A's inputs computed instead of loaded (what fetching them a round ahead could return at most)
#endif
#ifndef MF_PK
#define MF_PK 1   // phase B's channel pairs on the packed-fp32 instructions (round 6); 0 = one scalar FMA per channel
#endif
#ifndef MF_AHEAD
#define MF_AHEAD 0   // 1 = phase A's inputs fetched across the previous round's matrix phase: measured 870 -> 920 us (the five registers spill loop invariants to scratch, whose reloads drain vmcnt: round 6); 0 = loaded where they are used
#endif
#ifndef MF_CPL
#define MF_CPL 3
#endif
constexpr int MF_CPL_ = MF_CPL;    // channels per lane in phase B: 3 = all 64 lanes, 12-byte loads; 4 = lanes 0..47, 16-byte loads
constexpr int MF_C = 192;          // feature channels (3 per lane)
constexpr int MF_KS = 12;          // k-steps of 32: [mean 192 | variance 192]
constexpr int MF_LD = 400;         // staging row stride in halves (384 + pad): 200 dwords = 8 (mod 64) — the MFMA phase's ds_read_b128 (lane = (row, k-quarter), 16-lane
                                   // groups {0-3, 12-15, 20-27} ...) then start on 16 distinct 4-bank slots; 392 (= 4 mod 64) put rows c and c - 1 of neighbouring quarters on one slot:
                                   // a 2-way conflict in every group (SQ_LDS_BANK_CONFLICT 2.2e8 per launch in round 4; no measurable time: the phase hides behind the tap loads)
#ifndef MF_NW
#define MF_NW 8
#endif
constexpr int MF_NWAVES = MF_NW;   // waves per workgroup: 8 (two per SIMD) or 4 (one per SIMD: leaves half of the register file to co-resident kernels — the exact KNN on the side stream)
constexpr int MF_NS = 4 * MF_NWAVES;   // samples per round (4 per wave)
constexpr int MF_SLOT = 8;         // dwords per (sample, view): packed cell, 4 tap weights, view weight, 2 unused
constexpr int MF_LDS_BYTES = 2 * MF_NS * MF_LD * 2 + MF_NS * 64 * 4 + MF_NWAVES * 4 * 16 * MF_SLOT * 4 + MF_NS * 4;

__device__ __forceinline__ float mf_sum16(float v) {   // sum over an aligned group of 16 lanes (a DPP row), result in all 16
  v = nl_sum8(v);
  v += nl_dpp<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float mf_rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// A fragments of out_fc.0 for v_mfma_f32_16x16x32_bf16: [n-tile 4][k-step 12][part hi / lo][lane 64] x 8 bf16; lane: row n = 16 nt + (lane & 15), k = 32 ks + 8 (lane >> 4) + t,
// k < 192: mean of feature channel k (column 3 + k of the layer), k >= 192: its variance (column F + 3 + k - 192).  w9 [64][12]: the nine other columns + the bias.
__global__ void pack_mv_front_kernel(const float* __restrict__ w /*(64, 393)*/, const float* __restrict__ b, unsigned short* __restrict__ out, float* __restrict__ w9) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int F = MF_C + 3, LDW = 2 * F + 3;
  auto f2bf = [](float x) { unsigned u = __float_as_uint(x); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
  if (e < 4 * MF_KS * 64 * 8) {
    const int t = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) % MF_KS, nt = (e >> 9) / MF_KS;
    const int n = 16 * nt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + t;
    const int col = k < MF_C ? 3 + k : F + 3 + (k - MF_C);
    const float v = w[n * LDW + col];
    const unsigned short h = f2bf(v);
    const size_t base = ((size_t)(nt * MF_KS + ks) * 2) * 512 + lane * 8 + t;
    out[base] = h;
    out[base + 512] = f2bf(v - __uint_as_float(((unsigned)h) << 16));
  }
  if (e < 64 * 12) {
    const int n = e / 12, i = e % 12;
    float v = 0.f;
    if (i < 3) v = w[n * LDW + i];
    else if (i < 6) v = w[n * LDW + F + (i - 3)];
    else if (i < 9) v = w[n * LDW + 2 * F + (i - 6)];
    else if (i == 9) v = b[n];
    w9[e] = v;
  }
}

__global__ __launch_bounds__(64 * MF_NWAVES, 1) void mv_front_kernel(const NlViews vw, const float* __restrict__ viewsdev, const float* __restrict__ images,
                                                          const float* __restrict__ feat /*(V,h,w,192)*/, const float* __restrict__ xyz, int N,
                                                          const float* __restrict__ vis_in, const float* __restrict__ dd_in,
                                                          const mf_u32x4* __restrict__ wpack, const float* __restrict__ w9g, float* __restrict__ t64,
                                                          int* __restrict__ valid_s, float* __restrict__ rgbv, int nrounds, int rounds_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mf_lds[];
  unsigned short* st_hi = reinterpret_cast<unsigned short*>(mf_lds);
  unsigned short* st_lo = st_hi + MF_NS * MF_LD;
  float* partial = reinterpret_cast<float*>(st_lo + MF_NS * MF_LD);   // [32][64]
  float* slots = partial + MF_NS * 64;                                  // [8 waves][4 samples][16 views][MF_SLOT]
  float* wsumS = slots + MF_NWAVES * 4 * 16 * MF_SLOT;                  // [MF_NS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int V = vw.V;
  const int HW = vw.H * vw.Wimg;
  const size_t fmap = (size_t)vw.h * vw.w;

  // ---- resident: this wave's slice of out_fc.0 (n-tile nt) as A fragments, the small columns' row of output unit `lane`
  const int nt = wave & 3, half = wave >> 2;   // (4 waves: every wave one n-tile, one half of 16 samples)
  mf_u32x4 wa[MF_KS][2];
#pragma unroll
  for (int ks = 0; ks < MF_KS; ++ks) {
    wa[ks][0] = wpack[((nt * MF_KS + ks) * 2 + 0) * 64 + lane];
    wa[ks][1] = wpack[((nt * MF_KS + ks) * 2 + 1) * 64 + lane];
  }
  float w9[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) w9[i] = w9g[lane * 12 + i];

  const int lb = (int)nl_xcd_block();
  const int r_begin = lb * rounds_per_block, r_end = min(nrounds, r_begin + rounds_per_block);
  float* myslots = slots + (size_t)wave * 4 * 16 * MF_SLOT;

  // the round's per-(sample, view) inputs (position, visibility, depth difference) are fetched while the PREVIOUS round's out_fc.0 phase runs (round 6): phase A otherwise
  // starts with a global-memory latency at two waves per SIMD, and in that phase the accumulators and texel rows of phase B are dead, so five registers cost nothing
  auto load_in = [&](int round, float& X, float& Y, float& Z, float& vis, float& dd) __attribute__((always_inline)) {
    const int s = lane >> 4, v = lane & 15;
    const int n = round * MF_NS + wave * 4 + s;
    const int nn = n < N ? n : N - 1;
    const int vl = v < V ? v : 0;
    X = xyz[3 * (size_t)nn]; Y = xyz[3 * (size_t)nn + 1]; Z = xyz[3 * (size_t)nn + 2];
    const unsigned vo = (unsigned)vl * (unsigned)N + (unsigned)nn;   // V N < 2^31 (nl_mv_front_supported): a 32-bit offset from the scalar base, no hoisted 64-bit lane pointer
    vis = vis_in[vo]; dd = dd_in[vo];
  };
  float nX = 0.f, nY = 0.f, nZ = 0.f, nvis = 0.f, ndd = 0.f;
  if (MF_AHEAD && r_begin < r_end) load_in(r_begin, nX, nY, nZ, nvis, ndd);
  for (int round = r_begin; round < r_end; ++round) {
    const int n0 = round * MF_NS + wave * 4;
    // ---------------------------------------------------------------- phase A: lane = (sample s, view v)
    unsigned vmask;   // views in which at least one of the wave's four samples has a non-zero weight
    {
      const int s = lane >> 4, v = lane & 15;
      const int n = n0 + s;
      const bool live = n < N;
      const int nn = live ? n : N - 1;
      const bool vact = v < V;
      const int vl = vact ? v : 0;
      float X, Y, Z, vis_l = 0.f, dd_l = 0.f;
      if (MF_KO & 128) { X = 0.25f + 1e-6f * (float)nn; Y = 0.5f - 1e-6f * (float)nn; Z = 1.f + 2e-6f * (float)nn; }   // (knock-out 128: phase A's position / visibility / depth-difference inputs without memory)
      else if (MF_AHEAD) { X = nX; Y = nY; Z = nZ; vis_l = nvis; dd_l = ndd; }
      else load_in(round, X, Y, Z, vis_l, dd_l);
      const float4 p0 = *(const float4*)(viewsdev + 12 * vl), p1 = *(const float4*)(viewsdev + 12 * vl + 4), p2 = *(const float4*)(viewsdev + 12 * vl + 8);
      const float cx = fmaf(p0.z, Z, fmaf(p0.y, Y, p0.x * X)) + p0.w;
      const float cy = fmaf(p1.z, Z, fmaf(p1.y, Y, p1.x * X)) + p1.w;
      const float cz = fmaf(p2.z, Z, fmaf(p2.y, Y, p2.x * X)) + p2.w;
      const float zc = fmaxf(cz, 1e-8f);
      float px = cx / zc, py = cy / zc;
      px = fminf(fmaxf(px, -1e6f), 1e6f);
      py = fminf(fmaxf(py, -1e6f), 1e6f);
      // (the two bounds as SCALAR registers: as hoisted vector registers they were spilled to scratch and reloaded — with a full vmcnt(0) drain — three times per round)
      const float wm1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)vw.Wimg - 1.f)));
      const float hm1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)vw.H - 1.f)));
      const bool m1 = vact && (px <= wm1) && (px >= 0.f) && (py <= hm1) && (py >= 0.f) && (cz > 0.f);
      const unsigned long long bm = __ballot(m1);
      const int cnt1 = __popc((unsigned)(bm >> (16 * s)) & 0xffffu);
      const float xn = 2.f * px / (float)(vw.Wimg - 1) - 1.f;
      const float yn = 2.f * py / (float)(vw.H - 1) - 1.f;
      const Taps tf = make_taps<true, false>(xn, yn, vw.w, vw.h);
      const Taps ti = make_taps<true, false>(xn, yn, vw.Wimg, vw.H);
      // colour taps (zeros padding: masked weights, clamped offsets)
      float rgb[3] = {0.f, 0.f, 0.f};
      {
        int oi[4];
        unpack_taps(pack_taps(ti, vw.Wimg, vw.H), vw.Wimg, oi);
        const float i0 = (ti.mn && ti.mw) ? ti.nw : 0.f, i1 = (ti.mn && ti.me) ? ti.ne : 0.f, i2 = (ti.ms && ti.mw) ? ti.sw : 0.f, i3 = (ti.ms && ti.me) ? ti.se : 0.f;
        const float* ib = images + (size_t)vl * 3 * HW;
#pragma unroll
        for (int c = 0; c < ((MF_KO & 4) ? 0 : 3); ++c) {
          const float* pl = ib + (size_t)c * HW;
          rgb[c] = fmaf(pl[oi[3]], i3, fmaf(pl[oi[2]], i2, fmaf(pl[oi[1]], i1, pl[oi[0]] * i0)));
        }
      }
      const unsigned vo = (unsigned)vl * (unsigned)N + (unsigned)nn;   // V N < 2^31 (nl_mv_front_supported): a 32-bit offset from the scalar base, no hoisted 64-bit lane pointer
      const float vis = (MF_KO & 128) ? (vact ? 0.125f + 1e-7f * (float)vo : 0.f) : (vact ? vis_l : 0.f);
      const float dd = (MF_KO & 128) ? 0.01f : (vact ? dd_l : 0.f);
      const float vsum = mf_sum16(vis);
      const float wgt = vis / (vsum + 1e-8f);
      vmask = 0;
      {
        const unsigned long long nz = __ballot(wgt != 0.f);
        vmask = (unsigned)((nz | (nz >> 16) | (nz >> 32) | (nz >> 48)) & 0xffffull);
      }
      // statistics of the four per-view scalars (colour, depth difference) over the views: two passes inside the row of 16 lanes
      float q[4] = {rgb[0], rgb[1], rgb[2], dd}, mean[4], var[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mean[i] = mf_sum16(wgt * q[i]);
        const float d = q[i] - mean[i];
        var[i] = mf_sum16(wgt * (d * d));
      }
      const float wsum = mf_sum16(wgt);
      float* sl = myslots + (size_t)(s * 16 + v) * MF_SLOT;
      *(float4*)sl = make_float4(__uint_as_float(pack_taps(tf, vw.w, vw.h)), (tf.mn && tf.mw) ? tf.nw : 0.f, (tf.mn && tf.me) ? tf.ne : 0.f, (tf.ms && tf.mw) ? tf.sw : 0.f);
      *(float2*)(sl + 4) = make_float2((tf.ms && tf.me) ? tf.se : 0.f, wgt);
      {   // (the slot's address re-derived per round from an opaque copy of s: hoisted, it was one vector register too many — a scratch reload with a full drain per round)
        int s_o = s;
        asm volatile("" : "+v"(s_o));
        if (v == 0) wsumS[wave * 4 + s_o] = wsum;
      }
      if (live && vact) *(float4*)(rgbv + ((size_t)n * V + v) * 4) = make_float4(rgb[0], rgb[1], rgb[2], vis);
      if (live && v == 0) valid_s[n] = cnt1 > 1 ? 1 : 0;
      // the nine small columns + bias of out_fc.0 for the wave's four samples: lane = output unit
      const float wm = wsum / (float)V;
#pragma unroll
      for (int sI = 0; sI < 4; ++sI) {
        const int src = 16 * sI;
        float a = w9[9];
        a = fmaf(w9[0], mf_rl(mean[0], src), a); a = fmaf(w9[1], mf_rl(mean[1], src), a); a = fmaf(w9[2], mf_rl(mean[2], src), a);
        a = fmaf(w9[3], mf_rl(var[0], src), a); a = fmaf(w9[4], mf_rl(var[1], src), a); a = fmaf(w9[5], mf_rl(var[2], src), a);
        a = fmaf(w9[6], mf_rl(mean[3], src), a); a = fmaf(w9[7], mf_rl(var[3], src), a); a = fmaf(w9[8], mf_rl(wm, src), a);
        partial[(wave * 4 + sI) * 64 + lane] = a;
      }
    }
    __builtin_amdgcn_wave_barrier();   // the slots are wave-private: LDS ordering within the wave is all phase B needs

    // ---------------------------------------------------------------- phase B: lane = channels 3 lane .. 3 lane + 2, view by view over the four samples
    constexpr int CPL = MF_CPL_;
    float a1[4][CPL], a2[4][CPL];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < CPL; ++j) a1[s][j] = a2[s][j] = 0.f;
    const bool lact = CPL * lane < MF_C;
    const unsigned lch = lact ? (unsigned)(CPL * lane) : 0u;   // (idle lanes re-read channel group 0: no exec-masked loads)
    // The eight waves of a workgroup walk neighbouring samples: started on the same view they would miss on the same cold texel rows at the same moment
    // (eight requests to L2 per row, all waiting for the slowest).  Every wave starts its view loop at a different view — a function of the wave's position in
    // its RAY, so that a sample's summation order does not depend on how the batch was chunked — and finds most rows already fetched by a neighbour.
    const int rot = (int)(((unsigned)n0 % (unsigned)(vw.qS > 0 ? vw.qS : 1)) >> 2) % V;
    for (int vi = 0; vi < ((MF_KO & 2) ? 0 : V); ++vi) {
      int v = vi + rot;
      v = v >= V ? v - V : v;
      if (!((vmask >> v) & 1u)) continue;   // weight exactly 0 for all four samples: nothing of this view reaches a statistic (wave-uniform)
      const float* fb = feat + (size_t)v * fmap * MF_C;
      float T[4][CPL];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) T[k][j] = 0.f;
      unsigned cur = 0xffffffffu;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* sl = myslots + (size_t)(s * 16 + v) * MF_SLOT;
        const float4 c4 = *(const float4*)sl;      // (every lane reads the same address: an LDS broadcast)
        const float2 c2 = *(const float2*)(sl + 4);
        const unsigned cell = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(c4.x));
        if (cell != cur && !(MF_KO & 1)) {   // wave-uniform: the four texel rows of the new cell
          cur = cell;
          int o[4];
          unpack_taps((MF_KO & 64) ? (cell & 0xc0000000u) : cell, vw.w, o);   // (knock-out 64: every fetch from texel 0 of the view — all hits)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float* p = fb + ((unsigned)o[k] * (unsigned)MF_C + lch);
            if constexpr (CPL == 4) { const float4 t4 = *(const float4*)p; T[k][0] = t4.x; T[k][1] = t4.y; T[k][2] = t4.z; T[k][3] = t4.w; }
            else { T[k][0] = p[0]; T[k][1] = p[1]; T[k][2] = p[2]; }
          }
        }
        const float w0 = c4.y, w1 = c4.z, w2 = c4.w, w3 = c2.x, wg = c2.y;
        if constexpr (CPL == 3 && MF_PK) {
          // channels 0, 1 as one packed operation each (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: the same roundings in the same order, two thirds of the instructions)
          const mf_f32x2 t0 = {T[0][0], T[0][1]}, t1 = {T[1][0], T[1][1]}, t2 = {T[2][0], T[2][1]}, t3 = {T[3][0], T[3][1]};
          const mf_f32x2 W0 = {w0, w0}, W1 = {w1, w1}, W2 = {w2, w2}, W3 = {w3, w3}, WG = {wg, wg};
          const mf_f32x2 x = __builtin_elementwise_fma(t3, W3, __builtin_elementwise_fma(t2, W2, __builtin_elementwise_fma(t1, W1, t0 * W0)));
          const mf_f32x2 t = WG * x;
          mf_f32x2 A1 = {a1[s][0], a1[s][1]}, A2 = {a2[s][0], a2[s][1]};
          A1 += t;
          A2 = __builtin_elementwise_fma(t, x, A2);
          a1[s][0] = A1[0]; a1[s][1] = A1[1]; a2[s][0] = A2[0]; a2[s][1] = A2[1];
          const float xs = fmaf(T[3][2], w3, fmaf(T[2][2], w2, fmaf(T[1][2], w1, T[0][2] * w0)));
          const float ts = wg * xs;
          a1[s][2] += ts;
          a2[s][2] = fmaf(ts, xs, a2[s][2]);
        } else {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const float x = fmaf(T[3][j], w3, fmaf(T[2][j], w2, fmaf(T[1][j], w1, T[0][j] * w0)));
          const float t = wg * x;
          a1[s][j] += t;
          a2[s][j] = fmaf(t, x, a2[s][j]);
        }
        }
      }
    }
    if (MF_AHEAD && round + 1 < r_end) load_in(round + 1, nX, nY, nZ, nvis, ndd);   // (in flight across the staging writes, both barriers and the matrix phase)
    // every wave is through with the previous round's staging tile (its MFMA phase ended at a barrier) — this round's rows may be written
#pragma unroll
    for (int s = 0; s < ((MF_KO & 16) ? 0 : 4); ++s) {
      const float Ws = wsumS[wave * 4 + s];
      const int row = wave * 4 + s;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        if (!lact) break;
        const float m = a1[s][j];
        // sum w (x - m)^2 = sum w x^2 - m^2 (2 - sum w).  One pass: its absolute error is ~2e-7 (m^2 + var) where the reference's two-pass form (ibrnet.py:8-12) has
        // ~1e-7 var — below the split product that consumes it (1e-5 |m|) for |m| < 50; the difference can come out a rounding error below zero where the
        // views agree exactly (var = 0): clamped, as the two-pass value is a sum of squares (tests: the "+off4" rows of tools/scale_sweep.py)
        const float vr = fmaxf(a2[s][j] - m * m * (2.f - Ws), 0.f);
        const __bf16 mh = (__bf16)m, vh = (__bf16)vr;
        const __bf16 ml = (__bf16)(m - (float)mh), vl2 = (__bf16)(vr - (float)vh);
        const int k = CPL * lane + j;
        st_hi[row * MF_LD + k] = __builtin_bit_cast(unsigned short, mh);
        st_lo[row * MF_LD + k] = __builtin_bit_cast(unsigned short, ml);
        st_hi[row * MF_LD + MF_C + k] = __builtin_bit_cast(unsigned short, vh);
        st_lo[row * MF_LD + MF_C + k] = __builtin_bit_cast(unsigned short, vl2);
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- out_fc.0 on the staging tile: wave = (n-tile nt, sample half), 12 k-steps x 3 MFMAs
    {
      const int col = lane & 15, kq = lane >> 4;
      const int srow = 16 * half + col;
      mf_f32x4 acc = *(const mf_f32x4*)(partial + srow * 64 + 16 * nt + 4 * kq);
#pragma unroll
      for (int ks = 0; ks < ((MF_KO & 8) ? 0 : MF_KS); ++ks) {
        const mf_u32x4 bh = *(const mf_u32x4*)(st_hi + srow * MF_LD + 32 * ks + 8 * kq);
        const mf_u32x4 bl = *(const mf_u32x4*)(st_lo + srow * MF_LD + 32 * ks + 8 * kq);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, wa[ks][1]), __builtin_bit_cast(mf_bf16x8, bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, wa[ks][0]), __builtin_bit_cast(mf_bf16x8, bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, wa[ks][0]), __builtin_bit_cast(mf_bf16x8, bh), acc, 0, 0, 0);
      }
      const int ng = round * MF_NS + srow;
      // (the round's rows from a SCALAR base + a small lane offset: a hoisted 64-bit lane pointer was the last value this kernel spilled to scratch)
      float* trow = t64 + (size_t)__builtin_amdgcn_readfirstlane(round * MF_NS) * 64;
      if (ng < N) *(float4*)(trow + (srow * 64 + 16 * nt + 4 * kq)) = make_float4(nl_elu(acc[0]), nl_elu(acc[1]), nl_elu(acc[2]), nl_elu(acc[3]));
    }
    __syncthreads();   // the staging tile and the partial rows are free again
  }
}

}  // namespace

size_t nl_mv_front_pack_bytes() { return (size_t)4 * MF_KS * 2 * 64 * 16 + 64 * 12 * 4; }

int nl_pack_mv_front(const float* w_outfc0, const float* b_outfc0, void* out, hipStream_t st) {
  float* w9 = reinterpret_cast<float*>((char*)out + (size_t)4 * MF_KS * 2 * 64 * 16);
  hipLaunchKernelGGL(pack_mv_front_kernel, dim3((4 * MF_KS * 64 * 8 + 255) / 256), dim3(256), 0, st, w_outfc0, b_outfc0, (unsigned short*)out, w9);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

bool nl_mv_front_supported(int C, int V, int64_t N) { return C == MF_C && V >= 1 && V <= NL_MAX_VIEWS && N > 0 && N <= 0x7fffffffll / 32; }   // (int sample indices; rounds of 32)

// statistics + out_fc.0 of N samples: t64 (N, 64) hidden rows (post-ELU), valid_s (N), rgbv (N, V, 4) = tapped colours + visibility
int nl_launch_mv_front(const NlViews& vw, const float* viewsdev, const float* images, const float* feat, const float* xyz, int64_t N, const float* vis_in,
                       const float* dd_in, const void* pack, float* t64, int* valid_s, float* rgbv, hipStream_t st) {
  if (N <= 0) return NL_OK;
  const int g_mf_cus = nl_persistent_cus();
  if (g_mf_cus < 0) return g_mf_cus;
  // (per device, and the call is a table look-up: set every time rather than remembered per process)
  if (hipFuncSetAttribute((const void*)mv_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS_BYTES) != hipSuccess) return NL_ERR_HIP;
  const int nrounds = (int)nl_cdiv(N, MF_NS);
  const int blocks = nrounds < g_mf_cus ? (int)nl_xcd_grid(nrounds) : g_mf_cus;
  const int rpb = (int)nl_cdiv(nrounds, blocks);
  const float* w9 = reinterpret_cast<const float*>((const char*)pack + (size_t)4 * MF_KS * 2 * 64 * 16);
  hipLaunchKernelGGL(mv_front_kernel, dim3(blocks), dim3(64 * MF_NWAVES), MF_LDS_BYTES, st, vw, viewsdev, images, feat, xyz, (int)N, vis_in, dd_in, (const mf_u32x4*)pack, w9, t64,
                     valid_s, rgbv, nrounds, rpb);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
