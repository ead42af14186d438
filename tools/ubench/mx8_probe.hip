// Probe of gfx950's block-scaled FP8 matrix instruction for the "fp16 hi.hi + two MX-FP8 cross terms" arithmetic:
//   1. operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) A / B: which k does byte t of lane l hold?  (two hypotheses checked against a CPU product)
//   2. what the per-lane E8M0 scale operand multiplies, and what v_cvt_scalef32_pk_fp8_f32's scale does (multiply or divide)
//   3. issue rate: 12 x v_mfma_f32_32x32x16_bf16 (three-term split-bf16, today) against 4 x v_mfma_f32_32x32x16_f16 + 2 x the scaled fp8 instruction per K = 64 slab
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mx8_probe.hip -o tools/ubench/mx8_probe.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short v2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

static float e4m3_decode(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m / 8.f, -6);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -v : v;
}

__global__ void cvt_kernel(const float* x, unsigned char* out, int n, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n + 1) return;
  v2s r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[2 * i], x[2 * i + 1], scale, false);
  out[2 * i] = (unsigned char)(r[0] & 0xff);
  out[2 * i + 1] = (unsigned char)((r[0] >> 8) & 0xff);
}

__global__ void mfma_once(const v8i* a, const v8i* b, f32x16* d, const int* sa, const int* sb) {
  const int l = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  d[l] = acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r);
  bf16x8 a, b; f16x8 ah, bh; v8i a8, b8;
  for (int t = 0; t < 8; ++t) { a[t] = (__bf16)(0.001f * threadIdx.x + t); b[t] = (__bf16)(0.002f * threadIdx.x - t); ah[t] = (_Float16)(0.001f * threadIdx.x + t); bh[t] = (_Float16)(0.002f * threadIdx.x - t); a8[t] = 0x38383838 + threadIdx.x; b8[t] = 0x30303030 + t; }
  const int sc = 127;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {   // 8 K = 64 slabs per iteration
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[0], 0, 0, 0, sc, 0, sc);
        acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc[0], 0, 0, 0, sc, 0, sc);
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[0], 0, 0, 0, sc, 0, sc);
      } else {   // MODE 3: same as 1 with the two fp8 instructions on a second accumulator (is the dependent chain what limits them?)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[1], 0, 0, 0, sc, 0, sc);
        acc[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc[1], 0, 0, 0, sc, 0, sc);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void rate(float* out, const char* what) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<MODE><<<256, 256>>>(out, 10);
  hipEventRecord(e0);
  rate_kernel<MODE><<<256, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double slabs = (double)iters * 8;
  printf("%-70s %.1f ns per K=64 slab per wave; algorithmic %.0f TFLOP/s (32x32x64 MACs per slab, 1024 waves)\n", what, ms * 1e6 / slabs,
         1024.0 * slabs * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12);
}

int main() {
  // ---- 1. conversion semantics
  {
    const int n = 16;
    float hx[n] = {0.5f, 1.0f, 1.5f, 3.0f, -2.0f, 0.1f, 448.f, 500.f, 0.001f, 7.3f, -0.3f, 100.f, 0.0625f, 0.02f, 12.f, -448.f};
    float* dx; unsigned char* dout; hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, n);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (float sc : {1.0f, 4.0f, 0.25f}) {
      cvt_kernel<<<1, 64>>>(dx, dout, n, sc);
      unsigned char ho[n]; hipMemcpy(ho, dout, n, hipMemcpyDeviceToHost);
      printf("cvt_scalef32_pk_fp8_f32 scale=%g:", sc);
      for (int i = 0; i < n; ++i) printf(" %g->%g", hx[i], e4m3_decode(ho[i]));
      printf("\n");
    }
  }
  // ---- 2. operand layout + scale semantics
  {
    std::vector<float> A(32 * 64), B(64 * 32);
    srand(3);
    auto rv = [] { const float tab[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, -1.f, -0.5f, 3.f}; return tab[rand() & 7]; };
    for (auto& v : A) v = rv();
    for (auto& v : B) v = rv();
    auto enc = [](float v) -> unsigned char {   // exact for the table's values
      for (int b = 0; b < 256; ++b) if (e4m3_decode((unsigned char)b) == v && !(b == 0x80)) return (unsigned char)b;
      return 0; };
    v8i *da, *db; f32x16* dd; int *dsa, *dsb;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dd, 64 * 64); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    for (int hyp = 0; hyp < 2; ++hyp) {
      unsigned char pa[64][32], pb[64][32];
      for (int l = 0; l < 64; ++l)
        for (int t = 0; t < 32; ++t) {
          const int hh = l >> 5, j = l & 31;
          const int k = hyp == 0 ? 32 * hh + t : (t < 16 ? 16 * hh + t : 32 + 16 * hh + (t - 16));
          pa[l][t] = enc(A[j * 64 + k]);       // A[i = j][k]
          pb[l][t] = enc(B[k * 32 + j]);       // B[k][j]
        }
      for (int variant = 0; variant < 3; ++variant) {   // 0: all scales 1; 1: A's scale x2 for lanes >= 32; 2: B's scale x4 for lanes < 32
        int sa[64], sb[64];
        for (int l = 0; l < 64; ++l) { sa[l] = 127 + (variant == 1 && l >= 32 ? 1 : 0); sb[l] = 127 + (variant == 2 && l < 32 ? 2 : 0); }
        hipMemcpy(da, pa, sizeof(pa), hipMemcpyHostToDevice); hipMemcpy(db, pb, sizeof(pb), hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa, sizeof(sa), hipMemcpyHostToDevice); hipMemcpy(dsb, sb, sizeof(sb), hipMemcpyHostToDevice);
        mfma_once<<<1, 64>>>(da, db, dd, dsa, dsb);
        float hd[64][16]; hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
        double err = 0, mx = 0;
        for (int l = 0; l < 64; ++l)
          for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            double ref = 0;
            for (int k = 0; k < 64; ++k) {
              const double fa = (variant == 1 && k >= 32) ? 2.0 : 1.0, fb = (variant == 2 && k < 32) ? 4.0 : 1.0;   // expectation: a lane's scale applies to ITS 32 k values
              ref += fa * fb * (double)A[row * 64 + k] * (double)B[k * 32 + col];
            }
            err = fmax(err, fabs(ref - hd[l][r])); mx = fmax(mx, fabs(ref));
          }
        printf("layout hypothesis %d (%s), scale variant %d: max |D - ref| = %g (max |ref| %g)\n", hyp, hyp == 0 ? "lane holds k = 32 hh + t" : "k = 16 hh + t | 32 + 16 hh + t - 16", variant, err, mx);
      }
    }
  }
  // ---- 3. issue rates
  float* out; hipMalloc(&out, 256 * 256 * 4);
  rate<0>(out, "12 x mfma_f32_32x32x16_bf16 (three-term split-bf16, today)");
  rate<1>(out, "4 x mfma_f32_32x32x16_f16 + 2 x mfma_scale_f32_32x32x64 fp8 (one chain)");
  rate<3>(out, "4 x mfma_f32_32x32x16_f16 + 2 x mfma_scale fp8 (second accumulator)");
  rate<2>(out, "6 x mfma_scale_f32_32x32x64 fp8 only");
  return 0;
}
