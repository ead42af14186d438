// Fused neural-point branch — the MFMA roofline kernel.
//
//   per (sample, neighbour) row:  [feature(195) | posenc(63) | ray_diff_fc(27)]            model.py:394-409
//       -> base_mlp 285->W->W->W (LeakyReLU)                                                model.py:63-71
//       -> k/v projections W->128+128                                                       ibrnet.py:98-99
//   per sample: 4-head attention of the (precomputed) query over its 8 neighbours -> O (N,128)  ibrnet.py:28-45,104
//
// MI355X design
//   * Transposed MFMA: D^T = W * X^T, i.e. the WEIGHTS are the A operand and the activations the B operand of
//     v_mfma_f32_32x32x16_bf16.  A wave owns 32 rows (= 4 samples x 8 neighbours) for the whole chain; the C/D
//     fragment of layer L (lane = row, registers = 16 of every 32 features) is converted in registers to bf16
//     and IS the B fragment of layer L+1 — the K order of every packed weight matrix is permuted offline to the
//     accumulator's register order, so activations never touch LDS or HBM between the four layers.
//   * Weights stream from L2 into LDS with LDS-DMA (global_load_lds_dwordx4), double-buffered in 2-k-step chunks
//     whose LDS image is already in A-fragment order (one conflict-free ds_read_b128 per fragment), shared by
//     the 4 waves of the workgroup.
//   * bf16x3 mode (parity): activations and weights are split hi+lo; D += Alo*Bhi + Ahi*Blo + Ahi*Bhi (fp32 acc).
//   * Layer-1 operands are assembled in registers: pre-split bf16 feature rows are gathered with 16-B loads that
//     are already B fragments; positional encoding / ray_diff_fc are computed per lane in fp32 and split.
//   * One wave per SIMD (the kernel lives in the 512-register file): 128 accumulators + up to 152 operand regs.
#include <utility>
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


namespace {

constexpr int L1_KSTEPS = 6;    // 4 posenc + 2 ray_diff_fc k-steps (K = 96); the 195 feature columns come from the per-frame table T
constexpr int L1_CHUNKS = 3;

// LDS-DMA of 16 B per lane; the immediate offset OFF is added to BOTH addresses, so four consecutive 1-KB pieces share one
// scalar base and one M0 value
template <int OFF>
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, OFF, 0);
}

template <bool X3>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    __bf16 h = (__bf16)v[t];
    hi[t] = h;
    if (X3) lo[t] = (__bf16)(v[t] - (float)h);
  }
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
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

constexpr int NBUF = 4;   // LDS ring: chunk g lives in buffer g % 4, three chunks of LDS-DMA in flight

// Pointers are direct __restrict__ kernel arguments (not struct members) so that wave-uniform weight reads
// become scalar loads instead of occupying VGPRs.
struct PfScalars { int dir_stride, dir_div, N, M; float inv_span; };
struct PfView {   // what the body calls `a.` : plain locals, nothing escapes
  const float* xyz; const float* dir; int dir_stride, dir_div; const int* idx; const float* Q; float* O;
  const float* ptt; const float* sp_xyz; const float* sp_dir; const uint4* wstream;
  const float* bias; const float* rd_w; int N, M; float inv_span;
};

template <int NRT, bool X3>
__global__ __launch_bounds__(256, 1) void point_fused_kernel(
    const float* __restrict__ p_xyz, const float* __restrict__ p_dir, const int* __restrict__ p_idx, const float* __restrict__ p_Q,
    float* __restrict__ p_O, const float* __restrict__ p_ptt, const float* __restrict__ p_sp_xyz,
    const float* __restrict__ p_sp_dir, const uint4* __restrict__ p_wstream, const float* __restrict__ p_bias,
    const float* __restrict__ p_rd_w, const PfScalars sc, unsigned* __restrict__ p_logit_amax) {
  const float* __restrict__ rd_w = p_rd_w;
  const PfView a = {p_xyz, p_dir, sc.dir_stride, sc.dir_div, p_idx, p_Q, p_O, p_ptt, p_sp_xyz, p_sp_dir, p_wstream,
                    p_bias, p_rd_w, sc.N, sc.M, sc.inv_span};
  // ONE __shared__ object (a second one makes hipcc drain vmcnt(0) before every ds_read of an LDS-DMA pipeline):
  // 4 x 32 KB ring of A fragments [part hi/lo][k-step 0/1][row tile][lane], then 3 KB of biases
  __shared__ uint4 lds_all[NBUF * 2048 + 192];
  uint4 (*lds)[2048] = reinterpret_cast<uint4 (*)[2048]>(lds_all);
  float* sbias = reinterpret_cast<float*>(lds_all + NBUF * 2048);
  constexpr int W = 32 * NRT;
  constexpr int PARTS = X3 ? 2 : 1;
  constexpr int NC = L1_CHUNKS + 3 * NRT;   // chunks of the whole chain: L1 | L2 | L3 | KV
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, j = lane & 31, kk = j & 7;
  const int n = nl_xcd_block() * 16 + wave * 4 + (j >> 3);
  const bool live = n < a.N;
  const int nn = live ? n : a.N - 1;
  const char* wptr = (const char*)a.wstream;

  auto ort_of = [](int g) constexpr { return g < L1_CHUNKS + 2 * NRT ? NRT : 8; };
  auto glds_of = [=](int g) constexpr { return g < NC ? PARTS * 2 * (g < L1_CHUNKS + 2 * NRT ? NRT : 8) / 4 : 0; };   // LDS-DMA instructions per wave
  // LDS-DMA chunk g (2 k-steps): global image == LDS image, hi parts first.
  auto stage = [&](int g) __attribute__((always_inline)) {
    const int ort = ort_of(g);
    const int nkb = PARTS * 2 * ort;
    // wave w moves the contiguous pieces [w * nkb/4, (w+1) * nkb/4): groups of four share base + M0 through the immediate offset
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4) {
      if (4 * q4 < nkb / 4) {
        const int i = wave * (nkb / 4) + 4 * q4;
        const char* gp = wptr + (size_t)i * 1024 + lane * 16;
        uint4* lp = &lds[g % NBUF][i * 64];
        glds16<0>(gp, lp);
        if (4 * q4 + 1 < nkb / 4) glds16<1024>(gp, lp);
        if (4 * q4 + 2 < nkb / 4) glds16<2048>(gp, lp);
        if (4 * q4 + 3 < nkb / 4) glds16<3072>(gp, lp);
      }
    }
    wptr += (size_t)4 * ort * 1024;
  };

  // ---------------------------------------------------------------- layer-1 operands
  // feature columns: gathered row of the per-frame table T (already W1_feat . feature + b1, accumulator order) -> acc init;
  // k-steps 0-3 positional encoding, 4-5 ray_diff_fc are assembled below
  bf16x8 fh[2 * NRT > L1_KSTEPS ? 2 * NRT : L1_KSTEPS], fl[2 * NRT > L1_KSTEPS ? 2 * NRT : L1_KSTEPS];
  const bool have = live && kk < a.M && a.M > 0;   // knn_gather zero-fills k >= M (knn_utils.py:211-220)
  const int id = a.idx[(size_t)nn * 8 + kk];
  f32x16 acc[8];
  {
    const float* trow = a.ptt + (size_t)(have ? id : a.M) * W + 16 * hh;   // row M holds the bias alone
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 t4 = *(const float4*)(trow + 32 * rt + 4 * g);
        acc[rt][4 * g] = t4.x; acc[rt][4 * g + 1] = t4.y; acc[rt][4 * g + 2] = t4.z; acc[rt][4 * g + 3] = t4.w;
      }
  }
  for (int i = tid; i < 3 * W; i += 256) sbias[i] = a.bias[i];
  __builtin_amdgcn_sched_barrier(0);
  stage(0); stage(1); stage(2);
  {
    const float qx = a.xyz[3 * (size_t)nn], qy = a.xyz[3 * (size_t)nn + 1], qz = a.xyz[3 * (size_t)nn + 2];
    const float px = have ? a.sp_xyz[3 * (size_t)id] : 0.f, py = have ? a.sp_xyz[3 * (size_t)id + 1] : 0.f, pz = have ? a.sp_xyz[3 * (size_t)id + 2] : 0.f;
    const float off[3] = {(qx - px) * a.inv_span, (qy - py) * a.inv_span, (qz - pz) * a.inv_span};
    // ---- ray direction difference (model.py:396-399) -> ray_diff_fc (model.py:36-39): k-steps 4, 5
    {
      const size_t dr = (size_t)(nn / a.dir_div) * a.dir_stride;
      float dx, dy, dz;
      if (a.dir) { dx = a.dir[dr]; dy = a.dir[dr + 1]; dz = a.dir[dr + 2]; }
      else {
        const int i0 = a.idx[(size_t)nn * 8];
        const bool ok = live && a.M > 0;
        dx = ok ? a.sp_dir[4 * (size_t)i0] : 0.f; dy = ok ? a.sp_dir[4 * (size_t)i0 + 1] : 0.f; dz = ok ? a.sp_dir[4 * (size_t)i0 + 2] : 0.f;
      }
      const float ndx = have ? a.sp_dir[4 * (size_t)id] : 0.f, ndy = have ? a.sp_dir[4 * (size_t)id + 1] : 0.f, ndz = have ? a.sp_dir[4 * (size_t)id + 2] : 0.f;
      float r0 = dx - ndx, r1 = dy - ndy, r2 = dz - ndz;
      const float nr = sqrtf(r0 * r0 + r1 * r1 + r2 * r2) + 1e-8f;
      r0 /= nr; r1 /= nr; r2 /= nr;
      const float r3 = dx * ndx + dy * ndy + dz * ndz;
      float hid[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float s = rd_w[64 + i];
        s = fmaf(rd_w[i * 4 + 0], r0, s); s = fmaf(rd_w[i * 4 + 1], r1, s);
        s = fmaf(rd_w[i * 4 + 2], r2, s); s = fmaf(rd_w[i * 4 + 3], r3, s);
        hid[i] = nl_lrelu(s);
      }
      // outputs with wave-uniform (scalar-loaded) weights; each half keeps its 8 of every 16 slots
      const float* __restrict__ w2 = rd_w + 80;
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float r[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int o = 16 * qs + 8 * h2 + t;
            if (o < 27) {
              float s = w2[27 * 16 + o];
#pragma unroll
              for (int i = 0; i < 16; ++i) s = fmaf(w2[o * 16 + i], hid[i], s);
              r[h2] = nl_lrelu(s);
            } else r[h2] = 0.f;
          }
          v[t] = hh ? r[1] : r[0];
        }
        split8<X3>(v, fh[4 + qs], fl[4 + qs]);
      }
    }
    // ---- positional encoding (utils.py:5-35): sin/cos(off * 2^f), f = 0..9: k-steps 0..3.
    // Octaves come from the double-angle recurrence in fp64 started from one accurate evaluation per axis (abs
    // error < 1e-12, i.e. correctly rounded in fp32).  Pair order is axis-major, pi = 10*axis + f, so the
    // recurrence streams straight into fragments: slot pair (2*t2, 2*t2+1) of half hh in k-step qs holds pair
    // pi = 8*qs + 4*hh + t2;  pi = 30 -> (x, y);  pi = 31 -> (z, 0).
    {
      double s = 0.0, c = 1.0;
#pragma unroll
      for (int qs = 0; qs < 4; ++qs) {
        float v0[8], v1[8];   // candidates for hh = 0 / 1
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int pi = 8 * qs + u;
          float ps, pc;
          if (pi < 30) {
            const int ax = pi / 10, f = pi - 10 * ax;
            if (f == 0) sincos_d((double)off[ax], s, c);
            ps = (float)s; pc = (float)c;
            const double s2 = 2.0 * s * c;
            c = fma(-2.0 * s, s, 1.0);
            s = s2;
          } else if (pi == 30) { ps = off[0]; pc = off[1]; }
          else { ps = off[2]; pc = 0.f; }
          if (u < 4) { v0[2 * u] = ps; v0[2 * u + 1] = pc; } else { v1[2 * (u - 4)] = ps; v1[2 * (u - 4) + 1] = pc; }
        }
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = hh ? v1[t] : v0[t];
        split8<X3>(v, fh[qs], fl[qs]);
      }
    }
  }

  auto init_acc = [&](const float* bias, int ort) __attribute__((always_inline)) {   // bias lives in LDS
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
      if (rt < ort) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 b = bias ? *(const float4*)(bias + 32 * rt + 8 * g + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
          acc[rt][4 * g] = b.x; acc[rt][4 * g + 1] = b.y; acc[rt][4 * g + 2] = b.z; acc[rt][4 * g + 3] = b.w;
        }
      }
  };

  // one chunk: up to 2 k-steps x ort row tiles x (3 | 1) MFMAs.  A fragments are read from LDS two row tiles
  // ahead of their MFMAs and the scheduler is fenced per row tile, so at most 3 fragment pairs are live.
  auto compute = [&](int buf, int ort, int nks, const bf16x8 bh0, const bf16x8 bl0, const bf16x8 bh1, const bf16x8 bl1) __attribute__((always_inline)) {
    const int nt = nks * ort;   // (k-step, row tile) pairs in issue order: t = ks * ort + rt
    auto ldA = [&](int t, bf16x8& ah, bf16x8& al) __attribute__((always_inline)) {
      const int ks = t / ort, rt = t - ks * ort;
      ah = __builtin_bit_cast(bf16x8, lds[buf][((0 * 2 + ks) * ort + rt) * 64 + lane]);
      if (X3) al = __builtin_bit_cast(bf16x8, lds[buf][((1 * 2 + ks) * ort + rt) * 64 + lane]);
    };
    bf16x8 ah[3], al[3];
    ldA(0, ah[0], al[0]);
    if (nt > 1) ldA(1, ah[1], al[1]);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t < nt) {
        if (t + 2 < nt) ldA(t + 2, ah[(t + 2) % 3], al[(t + 2) % 3]);
        const int ks = t / ort, rt = t - ks * ort;
        const bf16x8 bh = ks ? bh1 : bh0;
        const bf16x8 bl = ks ? bl1 : bl0;
        if (X3) {
          acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t % 3], bh, acc[rt], 0, 0, 0);
          acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t % 3], bl, acc[rt], 0, 0, 0);
        }
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t % 3], bh, acc[rt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // LeakyReLU + bf16 split of the finished layer: C/D registers -> next layer's B fragments
  auto epilogue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
          // LeakyReLU = max(x, 0.01 x): packed multiply, and a bare v_max_f32 (fmaxf() adds a canonicalising v_max x,x per
          // element, which is one instruction in six of this epilogue; NaN ordering is irrelevant here)
          typedef float pf_f32x2 __attribute__((ext_vector_type(2)));
          const pf_f32x2 x2 = {acc[rt][8 * s + t], acc[rt][8 * s + t + 1]};
          const pf_f32x2 y2 = x2 * 0.01f;
          asm("v_max_f32 %0, %1, %2" : "=v"(v[t]) : "v"(x2[0]), "v"(y2[0]));
          asm("v_max_f32 %0, %1, %2" : "=v"(v[t + 1]) : "v"(x2[1]), "v"(y2[1]));
        }
        split8<X3>(v, fh[2 * rt + s], fl[2 * rt + s]);
      }
  };

  // ---------------------------------------------------------------- the whole chain as one chunk pipeline
  static_for<NC>([&](auto G) __attribute__((always_inline)) {
    constexpr int g = decltype(G)::value;
    // chunk g must have landed: only the LDS-DMA of chunks g+1, g+2 (issued later) may still be in flight
    wait_vmcnt<glds_of(g + 1) + glds_of(g + 2)>();
    __builtin_amdgcn_s_barrier();
    if constexpr (g + 3 < NC) stage(g + 3);   // its buffer held chunk g-1, which every wave finished before the barrier
    constexpr int layer = g < L1_CHUNKS ? 0 : (g - L1_CHUNKS) / NRT + 1;
    constexpr int c = g < L1_CHUNKS ? g : (g - L1_CHUNKS) % NRT;
    constexpr int nks = (layer == 0 && 2 * c + 1 >= L1_KSTEPS) ? 1 : 2;
    compute(g % NBUF, ort_of(g), nks, fh[2 * c], fl[2 * c], fh[2 * c + 1], fl[2 * c + 1]);
    constexpr bool last = layer == 0 ? (c == L1_CHUNKS - 1) : (c == NRT - 1);
    if constexpr (last && layer < 3) {
      epilogue();
      if constexpr (layer < 2) init_acc(sbias + (layer + 1) * W, NRT); else init_acc(nullptr, 8);
    }
  });

  // ---------------------------------------------------------------- attention over the 8 neighbours of each sample
  // acc[h] = k-projection of head h, acc[4+h] = v-projection; register r <-> dim i = (r&3) + 8*(r>>2) + 4*hh
  const float inv_temp = 1.0f / 5.656854249492381f;
  const float* qrow = a.Q + (size_t)nn * 128;
  float att[4];
  float lmax = 0.f;   // conditioning indicator (nl_frame_diagnostics): the largest |attention logit| scored; NaN counts as +inf
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    float p = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 q4 = *(const float4*)(qrow + 32 * h + 8 * g + 4 * hh);
      p = fmaf(q4.x * inv_temp, acc[h][4 * g], p);
      p = fmaf(q4.y * inv_temp, acc[h][4 * g + 1], p);
      p = fmaf(q4.z * inv_temp, acc[h][4 * g + 2], p);
      p = fmaf(q4.w * inv_temp, acc[h][4 * g + 3], p);
    }
    p += __shfl_xor(p, 32, 64);
    lmax = (p == p) ? fmaxf(lmax, fabsf(p)) : __builtin_inff();
    const float mx = nl_max8(p);          // over the 8 neighbours (lanes) of a sample
    const float e = expf(p - mx);
    att[h] = e / nl_sum8(e);
  }
#pragma unroll
  for (int h = 0; h < 4; ++h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float o[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        o[t] = nl_sum8(att[h] * acc[4 + h][4 * g + t]);
      }
      if (live && kk == 0) *(float4*)(a.O + (size_t)n * 128 + 32 * h + 8 * g + 4 * hh) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  if (p_logit_amax) {   // one atomic per wave at most: the running maximum is read first (L2-resident) and only a larger value is written
    const float m = wave_max(live ? lmax : 0.f);
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > *(volatile const unsigned*)p_logit_amax) atomicMax(p_logit_amax, __float_as_uint(m));
  }
}

// ---------------------------------------------------------------------------------------------------- packing
__device__ __forceinline__ unsigned short pf_f2bf(float x) {
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// Weight stream: L1 (10 chunks) | L2 (NRT) | L3 (NRT) | KV (NRT); chunk = [hi ks0][hi ks1][lo ks0][lo ks1], each
// ORT x 64 lanes x 8 bf16 in A-fragment order (lane: out row = 32*rt + (lane&31), k slots 8*(lane>>5) + t).
__global__ void pack_point_stream_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                         const float* __restrict__ wk, const float* __restrict__ wv, unsigned short* __restrict__ out,
                                         int NRT, int F) {
  const int W = 32 * NRT;
  const long long per_l1 = (long long)L1_CHUNKS * 2 * NRT * 512, per_lw = (long long)NRT * 2 * NRT * 512, per_kv = (long long)NRT * 2 * 8 * 512;
  const long long total = per_l1 + 2 * per_lw + per_kv;   // number of (k-step, rt, lane, t) elements
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int layer; long long r = e;
  if (r < per_l1) layer = 0; else if ((r -= per_l1) < per_lw) layer = 1; else if ((r -= per_lw) < per_lw) layer = 2; else { r -= per_lw; layer = 3; }
  const int ort = layer == 3 ? 8 : NRT;
  const int t = (int)(r & 7), lane = (int)((r >> 3) & 63);
  long long r2 = r >> 9;
  const int rt = (int)(r2 % ort); r2 /= ort;
  const int ks = (int)(r2 & 1), chunk = (int)(r2 >> 1);
  const int q = 2 * chunk + ks, hh = lane >> 5, orow = 32 * rt + (lane & 31);
  float v = 0.f;
  if (layer == 0) {   // k-steps 0-3 positional encoding, 4-5 ray_diff_fc (feature columns live in the per-frame table T)
    int col = -1;
    if (q < 4) {
      const int pi = 8 * q + 4 * hh + (t >> 1), comp = t & 1;
      if (pi < 30) { const int ax = pi / 10, f = pi - 10 * ax; col = F + 3 + 6 * f + (comp ? 3 : 0) + ax; }
      else if (pi == 30) col = F + comp;
      else col = comp == 0 ? F + 2 : -1;
    } else if (q < 6) { const int o = 16 * (q - 4) + 8 * hh + t; col = o < 27 ? F + 63 + o : -1; }
    if (col >= 0) v = w1[(size_t)orow * (F + 90) + col];
  } else {
    const int fin = 32 * (q >> 1) + 16 * (q & 1) + (t & 3) + 8 * (t >> 2) + 4 * hh;
    if (layer == 1) v = w2[(size_t)orow * W + fin];
    else if (layer == 2) v = w3[(size_t)orow * W + fin];
    else v = orow < 128 ? wk[(size_t)orow * W + fin] : wv[(size_t)(orow - 128) * W + fin];
  }
  // destination
  long long chunk_base;   // in bf16 elements
  const long long c_l = (long long)4 * NRT * 512, c_kv = (long long)4 * 8 * 512;
  if (layer == 0) chunk_base = chunk * c_l;
  else if (layer == 1) chunk_base = L1_CHUNKS * c_l + chunk * c_l;
  else if (layer == 2) chunk_base = (L1_CHUNKS + NRT) * c_l + chunk * c_l;
  else chunk_base = (L1_CHUNKS + 2 * NRT) * c_l + chunk * c_kv;
  const long long in_chunk = ((long long)(ks * ort + rt) * 64 + lane) * 8 + t;
  const unsigned short h = pf_f2bf(v);
  const float hf = __uint_as_float(((unsigned int)h) << 16);
  out[chunk_base + in_chunk] = h;
  out[chunk_base + (long long)2 * ort * 512 + in_chunk] = pf_f2bf(v - hf);
}

// B operand + bias of the per-frame table GEMM  T[m][c'] = sum_k feat[m][k] * W1[f(c')][k] + b1[f(c')], where column c' =
// 32*rt + 16*hh + r is the accumulator-order slot of output feature f(c') = 32*rt + (r&3) + 8*(r>>2) + 4*hh
__global__ void pack_ptt_kernel(const float* __restrict__ w1 /*(W, F+90)*/, const float* __restrict__ b1, int W, int F, int Kpad, int Npad,
                                float* __restrict__ B32 /*[Kpad][Npad]*/, float* __restrict__ bias /*[Npad]*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * W) return;
  const int k = i / W, c = i - k * W;
  const int rt = c >> 5, hh = (c >> 4) & 1, r = c & 15;
  const int f = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * hh;
  B32[(size_t)k * Npad + c] = w1[(size_t)f * (F + 90) + k];
  if (k == 0) bias[c] = b1[f];
}

// sum_k of the normalised aggregation weights (model.py:419-427 with correlation == 1/K), one lane per sample
__global__ void wscale_kernel(const int* __restrict__ idx, const float* __restrict__ d2, const float* __restrict__ conf, int N, int K,
                              int M, float* __restrict__ wscale) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float w[NL_KNN_MAX_K];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) {
    w[k] = 0.f;
    if (k < K) {
      const float dist = sqrtf(d2[(size_t)n * K + k]);
      const float c = k < M ? conf[idx[(size_t)n * K + k]] : 0.f;
      float x = 1.f / fmaxf(dist, 1e-8f);
      x = x * (1.f / (float)K);
      x = x * c;
      w[k] = x;
      sum += x;
    }
  }
  const float den = fmaxf(sum, 1e-8f);
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < NL_KNN_MAX_K; ++k) tot += w[k] / den;
  wscale[n] = tot;
}

}  // namespace

size_t nl_point_stream_bytes(int W) {
  const int NRT = W / 32;
  return ((size_t)(L1_CHUNKS + 2 * NRT) * 4 * NRT + (size_t)NRT * 32) * 1024;
}

int nl_pack_point_stream(const float* w1, const float* w2, const float* w3, const float* wk, const float* wv, void* out, int W, int F,
                         hipStream_t st) {
  const int NRT = W / 32;
  const long long total = (long long)L1_CHUNKS * 2 * NRT * 512 + 2LL * NRT * 2 * NRT * 512 + (long long)NRT * 2 * 8 * 512;
  hipLaunchKernelGGL(pack_point_stream_kernel, dim3((unsigned)nl_cdiv(total, 256)), dim3(256), 0, st, w1, w2, w3, wk, wv,
                     (unsigned short*)out, NRT, F);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_pack_ptt(const float* w1, const float* b1, int W, int F, int Kpad, int Npad, float* B32, float* bias, hipStream_t st) {
  hipLaunchKernelGGL(pack_ptt_kernel, dim3((unsigned)nl_cdiv((int64_t)F * W, 256)), dim3(256), 0, st, w1, b1, W, F, Kpad, Npad, B32, bias);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_wscale(const int* idx, const float* d2, const float* conf, int64_t N, int K, int64_t M, float* wscale, hipStream_t st) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(wscale_kernel, dim3((unsigned)nl_cdiv(N, 256)), dim3(256), 0, st, idx, d2, conf, (int)N, K,
                     (int)(M > 0x7fffffff ? 0x7fffffff : M), wscale);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

bool nl_point_fused_supported(int W, int precision) {
  return precision != NL_PREC_F32 && (W == 64 || W == 128 || W == 256);
}

int nl_launch_point_fused(const NlPointFusedArgs& a, int W, int precision, hipStream_t st) {
  if (a.N <= 0) return NL_OK;
  dim3 grid(nl_xcd_grid(nl_cdiv(a.N, 16)));
  const bool x3 = precision == NL_PREC_BF16X3;
#define NL_PF(NRT)                                                                                           \
  do {                                                                                                       \
    const PfScalars sc{a.dir_stride, a.dir_div, a.N, a.M, a.inv_span};                                       \
    if (x3) hipLaunchKernelGGL((point_fused_kernel<NRT, true>), grid, dim3(256), 0, st, a.xyz, a.dir, a.idx, a.Q, a.O, a.ptt, \
                               a.sp_xyz, a.sp_dir, a.wstream, a.bias, a.rd_w, sc, a.logit_amax);             \
    else hipLaunchKernelGGL((point_fused_kernel<NRT, false>), grid, dim3(256), 0, st, a.xyz, a.dir, a.idx, a.Q, a.O, a.ptt, \
                            a.sp_xyz, a.sp_dir, a.wstream, a.bias, a.rd_w, sc, a.logit_amax);                \
  } while (0)
  if (W == 256) NL_PF(8);
  else if (W == 128) NL_PF(4);
  else if (W == 64) NL_PF(2);
  else return NL_ERR_UNSUPPORTED;
#undef NL_PF
  NL_LAUNCH_CHECK();
  return NL_OK;
}
