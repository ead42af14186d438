// Weight gradients of the linear layers (training; SURVEY.md §8 row f2, reference: torch autograd's addmm / conv backward under
// compute_render_loss, conditional_nerf/model.py:641-685):
//
//     gW[m][n] += sum_r dY[r][m] * X[r][n]        gb[m] += sum_r dY[r][m]
//
// dY (rows, M) = the gradient at the layer's pre-activation output, X (rows, N) = the layer's input, both fp32 row-major in the
// backward call's workspace.  rows is 10^5 .. 10^6 and M x N at most 256 x 416: the reduction runs over the LONG dimension, so the
// kernel is a split-K GEMM with both operands transposed on the way into LDS:
//   * a workgroup owns one 128 x 128 tile of gW and a contiguous range of rows; it walks the range 32 rows at a time;
//   * loader: thread (cg, kg) reads rows 8 kg .. 8 kg + 7 of four neighbouring columns (coalesced float4 rows), splits every value into
//     bf16 hi / lo ONCE (the fragment is then shared by the two waves that use it) and stores one 8-value k-fragment per column
//     (ds_write_b128; the column order inside LDS is permuted so that writes are contiguous and fragment reads conflict-free);
//   * four waves (2 x 2), each 64 x 64 of the tile = 4 accumulators; per 32 rows 2 k-steps x 4 tiles x 3 MFMAs (split-bf16 hi*hi +
//     hi*lo + lo*hi: a gradient's scale is arbitrary, so fp16's range is not an option; 2^-16 relative is far inside the
//     gradient tolerances);
//   * global loads of the next 32 rows are in flight while the current ones are multiplied (register double buffer);
//   * the partial tiles of the row ranges go to a scratch buffer and a second kernel adds them IN A FIXED ORDER into gW (+=, so the
//     caller's chunk loop accumulates and the result is bit-reproducible: no atomics).
// Convolutions over the samples of a ray (k = 3) are three such products with X shifted by -1 / 0 / +1 rows inside each ray
// (`shift`, `period`: rows whose neighbour falls outside the ray read zero) and an output stride (gW[m][n * cs + co]).
#include "common.h"

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int WG_S = 64 * 16 + 32;      // bytes between the four column phases (c = col & 3): the +32 staggers the banks of a fragment read
constexpr int WG_KG = 4 * WG_S;         // bytes per k-group (8 rows)
constexpr int WG_PLANE = 4 * WG_KG;     // hi plane, then lo plane
constexpr int WG_LDS = 2 * WG_PLANE;

// up to three products of one shape in a launch (the taps of a convolution, the phases of a transposed one): blockIdx.z = sub * nsplit + split
struct WgArgs {
  const float* dY[3]; int ldy; int M;
  const float* X[3]; int ldx; int N;
  long long rows, rows_per_split;
  int shift[3], period;
  int nsplit, bias_sub;   // bias_sub: the sub-problem whose dY also gives the bias gradient (or -1)
  float* part;    // [sub][nsplit][M][N]
  float* bpart;   // [nsplit][M] or null
};

__device__ __forceinline__ void wg_split(const float (&v)[8], uint4& hi, uint4& lo) {
  wg_bf16x8 h, l;
#pragma unroll
  for (int t = 0; t < 8; ++t) { const __bf16 b = (__bf16)v[t]; h[t] = b; l[t] = (__bf16)(v[t] - (float)b); }
  hi = __builtin_bit_cast(uint4, h); lo = __builtin_bit_cast(uint4, l);
}

__global__ __launch_bounds__(256) void wgrad_kernel(const WgArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[WG_LDS];
  __shared__ float bsum[4][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = lane, kg = wave;                 // loader role: columns 4 cg .. 4 cg + 3 of the 256 (128 of dY | 128 of X), rows 8 kg .. + 7
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  const bool isA = cg < 32;
  const int col0 = isA ? m0 + 4 * cg : n0 + 4 * (cg - 32);
  const int sub = blockIdx.z / a.nsplit, zsp = blockIdx.z - sub * a.nsplit;
  const float* src = isA ? a.dY[sub] : a.X[sub];
  const int ld = isA ? a.ldy : a.ldx;
  const bool colok = col0 < (isA ? a.M : a.N);    // (the float4 may run up to 3 columns past M / N — inside the row, ld is a multiple of 4 — and is dropped at the end)
  const int shift = isA ? 0 : a.shift[sub];
  const long long r_begin = (long long)zsp * a.rows_per_split;
  const long long r_end = min(a.rows, r_begin + a.rows_per_split);
  const bool has_bias = a.bpart && sub == a.bias_sub;
  const bool want_bias = has_bias && blockIdx.y == 0 && isA;

  float4 ld_[8];
  auto fetch = [&](long long r0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const long long r = r0 + 8 * kg + t;
      bool ok = colok && r < r_end;
      long long rr = r;
      if (shift != 0) {   // (rows < 2^31: checked by the launcher)
        const int s = (int)((unsigned)r % (unsigned)a.period) + shift;
        ok = ok && s >= 0 && s < a.period;
        rr = r + shift;
      }
      ld_[t] = ok ? *(const float4*)(src + (size_t)rr * ld + col0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  auto stage = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = c == 0 ? ld_[t].x : c == 1 ? ld_[t].y : c == 2 ? ld_[t].z : ld_[t].w;
      if (want_bias) {
#pragma unroll
        for (int t = 0; t < 8; ++t) bs[c] += v[t];
      }
      uint4 hi, lo;
      wg_split(v, hi, lo);
      char* p = lds + kg * WG_KG + c * WG_S + cg * 16;
      *(uint4*)p = hi;
      *(uint4*)(p + WG_PLANE) = lo;
    }
  };
  // fragment of LDS column `col` (0..255), k-group kgi
  auto frag = [&](int col, int kgi, int plane) {
    return __builtin_bit_cast(wg_bf16x8, *(const uint4*)(lds + plane * WG_PLANE + kgi * WG_KG + (col & 3) * WG_S + (col >> 2) * 16));
  };
  const int wm = wave >> 1, wn = wave & 1, hh = lane >> 5, i = lane & 31;
  wg_f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  if (r_begin < r_end) fetch(r_begin);
  for (long long r0 = r_begin; r0 < r_end; r0 += 32) {
    __syncthreads();           // the previous step's fragment reads are done
    stage();
    __syncthreads();
    if (r0 + 32 < r_end) fetch(r0 + 32);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      wg_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        ah[x] = frag(wm * 64 + x * 32 + i, 2 * s + hh, 0); al[x] = frag(wm * 64 + x * 32 + i, 2 * s + hh, 1);
        bh[x] = frag(128 + wn * 64 + x * 32 + i, 2 * s + hh, 0); bl[x] = frag(128 + wn * 64 + x * 32 + i, 2 * s + hh, 1);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[x], bh[y], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[x], bl[y], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[x], bh[y], acc[x][y], 0, 0, 0);
        }
    }
  }
  float* part = a.part + (size_t)blockIdx.z * a.M * a.N;   // (= [sub][split])
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int n = n0 + wn * 64 + y * 32 + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + x * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
        if (m < a.M && n < a.N) part[(size_t)m * a.N + n] = acc[x][y][r];
      }
    }
  if (has_bias && blockIdx.y == 0) {
    if (isA) {
#pragma unroll
      for (int c = 0; c < 4; ++c) bsum[kg][4 * cg + c] = bs[c];
    }
    __syncthreads();
    if (tid < 128 && m0 + tid < a.M) a.bpart[(size_t)zsp * a.M + m0 + tid] = (bsum[0][tid] + bsum[1][tid]) + (bsum[2][tid] + bsum[3][tid]);
  }
}

// The same product for the SMALL layers (M, N <= 32: the NeuRay decoders, the blend MLP's tail, ray_diff_fc): one 32 x 32 tile per workgroup, so the
// four waves split the ROWS instead — a step is 128 rows, wave w multiplies rows 32 w .. 32 w + 31 of it — and add their accumulators through LDS at
// the end.  16x fewer MFMAs than padding the tile to 128 x 128.
constexpr int WS_S = 16 * 16 + 32, WS_KG = 4 * WS_S, WS_PLANE = 16 * WS_KG, WS_LDS = 2 * WS_PLANE;
__global__ __launch_bounds__(256) void wgrad_small_kernel(const WgArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[WS_LDS];
  __shared__ float bsum[16][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = tid & 15, kg = tid >> 4;          // loader role: columns 4 cg .. + 3 of the 64 (32 of dY | 32 of X), rows 8 kg .. + 7 of the 128
  const bool isA = cg < 8;
  const int col0 = isA ? 4 * cg : 4 * (cg - 8);
  const int sub = blockIdx.z / a.nsplit, zsp = blockIdx.z - sub * a.nsplit;
  const float* src = isA ? a.dY[sub] : a.X[sub];
  const int ld = isA ? a.ldy : a.ldx;
  const bool colok = col0 < (isA ? a.M : a.N);
  const int shift = isA ? 0 : a.shift[sub];
  const long long r_begin = (long long)zsp * a.rows_per_split;
  const long long r_end = min(a.rows, r_begin + a.rows_per_split);
  const bool has_bias = a.bpart && sub == a.bias_sub;
  const bool want_bias = has_bias && isA;
  float4 ld_[8];
  auto fetch = [&](long long r0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const long long r = r0 + 8 * kg + t;
      bool ok = colok && r < r_end;
      long long rr = r;
      if (shift != 0) {
        const int s = (int)((unsigned)r % (unsigned)a.period) + shift;
        ok = ok && s >= 0 && s < a.period;
        rr = r + shift;
      }
      ld_[t] = ok ? *(const float4*)(src + (size_t)rr * ld + col0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  auto stage = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = c == 0 ? ld_[t].x : c == 1 ? ld_[t].y : c == 2 ? ld_[t].z : ld_[t].w;
      if (want_bias) {
#pragma unroll
        for (int t = 0; t < 8; ++t) bs[c] += v[t];
      }
      uint4 hi, lo;
      wg_split(v, hi, lo);
      char* p = lds + kg * WS_KG + c * WS_S + cg * 16;
      *(uint4*)p = hi;
      *(uint4*)(p + WS_PLANE) = lo;
    }
  };
  auto frag = [&](int col, int kgi, int plane) {
    return __builtin_bit_cast(wg_bf16x8, *(const uint4*)(lds + plane * WS_PLANE + kgi * WS_KG + (col & 3) * WS_S + (col >> 2) * 16));
  };
  const int hh = lane >> 5, i = lane & 31;
  wg_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (r_begin < r_end) fetch(r_begin);
  for (long long r0 = r_begin; r0 < r_end; r0 += 128) {
    __syncthreads();
    stage();
    __syncthreads();
    if (r0 + 128 < r_end) fetch(r0 + 128);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kgi = 4 * wave + 2 * s + hh;
      const wg_bf16x8 ah = frag(i, kgi, 0), al = frag(i, kgi, 1), bh = frag(32 + i, kgi, 0), bl = frag(32 + i, kgi, 1);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);   // [wave][r 16][lane 64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  if (want_bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[kg][4 * cg + c] = bs[c];
  }
  __syncthreads();
  float* part = a.part + (size_t)blockIdx.z * a.M * a.N;
#pragma unroll
  for (int q = 0; q < 4; ++q) {   // 1024 outputs, 256 threads
    const int e = q * 256 + tid, r = e >> 6, l = e & 63;
    const float v = (red[(0 * 16 + r) * 64 + l] + red[(1 * 16 + r) * 64 + l]) + (red[(2 * 16 + r) * 64 + l] + red[(3 * 16 + r) * 64 + l]);
    const int m = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3), n = l & 31;
    if (m < a.M && n < a.N) part[(size_t)m * a.N + n] = v;
  }
  if (has_bias && tid < 32 && tid < a.M) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += bsum[g][tid];
    a.bpart[(size_t)zsp * a.M + tid] = v;
  }
}

// gW[m * ldc + n * cs + co[sub]] += sum_z part[sub][z][m][n]  (fixed order; sub = blockIdx.y);  gb[m] += sum_z bpart[z][m] (with sub bias_sub)
struct WgCo { int co[3]; };
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part_all, const float* __restrict__ bpart, int nsplit, int M, int N,
                                                           float* __restrict__ gW, int ldc, int cs, const WgCo co, int bias_sub, float* __restrict__ gb) {
  // 32 elements per block, 8 threads each over interleaved row ranges, combined in a fixed order
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, zz = threadIdx.x >> 5;
  const int sub = blockIdx.y;
  const float* part = part_all + (size_t)sub * nsplit * M * N;
  const bool wb = gb && bpart && sub == bias_sub;
  const int tot = M * N + (wb ? M : 0);
  const int e = blockIdx.x * 32 + el;
  float s = 0.f;
  if (e < tot) {
    const bool isb = e >= M * N;
    const float* src = isb ? bpart + (e - M * N) : part + e;
    const size_t stride = isb ? (size_t)M : (size_t)M * N;
    for (int z = zz; z < nsplit; z += 8) s += src[(size_t)z * stride];
  }
  red[zz][el] = s;
  __syncthreads();
  if (zz == 0 && e < tot) {
    const float v = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
    if (e < M * N) { const int m = e / N, n = e - m * N; gW[(size_t)m * ldc + (size_t)n * cs + co.co[sub]] += v; }
    else gb[e - M * N] += v;
  }
}

}  // namespace

// scratch floats needed for one call
size_t nl_wgrad_scratch_floats(int64_t rows, int M, int N) {
  (void)rows;
  return (size_t)256 * ((size_t)M * N + M);   // up to 256 row ranges
}

// gW (M, ldc...) += dY^T X over `rows` rows; gb (M) += column sums of dY (gb may be null).  ldy, ldx multiples of 4, pointers 16-byte aligned.
// X rows are read at r + shift when period > 0 and (r % period) + shift stays inside [0, period), else as zero.
// nsub (1..3) products of one shape in ONE launch pair: dY[i], X[i], shift[i] -> gW[m * ldc + n * cs + co[i]]; gb from dY[bias_sub] (-1: none)
int nl_launch_wgrad_multi(int nsub, const float* const* dY, int ldy, int M, const float* const* X, int ldx, int N, int64_t rows, const int* shift, int period,
                          float* gW, int ldc, int cs, const int* co, float* gb, int bias_sub, float* scratch, size_t scratch_floats, hipStream_t st) {
  if (rows <= 0 || M <= 0 || N <= 0 || nsub <= 0) return NL_OK;
  if (nsub > 3) return NL_ERR_BAD_ARG;
  if ((ldy & 3) || (ldx & 3)) return NL_ERR_UNSUPPORTED;
  for (int i = 0; i < nsub; ++i) {
    if (((uintptr_t)dY[i] & 15) || ((uintptr_t)X[i] & 15)) return NL_ERR_UNSUPPORTED;
    if (shift[i] != 0 && period <= 0) return NL_ERR_BAD_ARG;
  }
  if (rows >= (1ll << 31)) return NL_ERR_BAD_ARG;
  if (!gb) bias_sub = -1;
  const bool small = M <= 32 && N <= 32;
  const int nbm = small ? 1 : (int)nl_cdiv(M, 128), nbn = small ? 1 : (int)nl_cdiv(N, 128);
  int nsplit = (small ? 256 : 1024 / (nbm * nbn)) / nsub;   // the same number of workgroups (and of partial tiles) whatever nsub is
  const int64_t maxsplit = nl_cdiv(rows, small ? 1024 : 256);
  if (nsplit > maxsplit) nsplit = (int)maxsplit;
  if (nsplit > 256 / nsub) nsplit = 256 / nsub;
  if (nsplit < 1) nsplit = 1;
  if ((size_t)nsub * nsplit * ((size_t)M * N) + (size_t)nsplit * M > scratch_floats) return NL_ERR_WORKSPACE;
  WgArgs a;
  WgCo cc;
  for (int i = 0; i < 3; ++i) { const int j = i < nsub ? i : 0; a.dY[i] = dY[j]; a.X[i] = X[j]; a.shift[i] = shift[j]; cc.co[i] = co[j]; }
  a.ldy = ldy; a.M = M; a.ldx = ldx; a.N = N; a.rows = rows;
  a.rows_per_split = nl_align_up((size_t)nl_cdiv(rows, nsplit), 128);
  nsplit = (int)nl_cdiv(rows, a.rows_per_split);
  a.period = period > 0 ? period : 1;
  a.nsplit = nsplit; a.bias_sub = bias_sub;
  a.part = scratch; a.bpart = gb ? scratch + (size_t)nsub * nsplit * M * N : nullptr;
  if (small) hipLaunchKernelGGL(wgrad_small_kernel, dim3(1, 1, nsplit * nsub), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wgrad_kernel, dim3(nbm, nbn, nsplit * nsub), dim3(256), 0, st, a);
  NL_LAUNCH_CHECK();
  const int tot = M * N + (gb ? M : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)nl_cdiv(tot, 32), nsub), dim3(256), 0, st, a.part, a.bpart, nsplit, M, N, gW, ldc, cs, cc, bias_sub, gb);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
int nl_launch_wgrad(const float* dY, int ldy, int M, const float* X, int ldx, int N, int64_t rows, int shift, int period, float* gW, int ldc, int cs, int co,
                    float* gb, float* scratch, size_t scratch_floats, hipStream_t st) {
  return nl_launch_wgrad_multi(1, &dY, ldy, M, &X, ldx, N, rows, &shift, period, gW, ldc, cs, &co, gb, 0, scratch, scratch_floats, st);
}
