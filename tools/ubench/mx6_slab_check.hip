// Check of the "fp16 hi.hi + two MX-FP6 cross terms" arithmetic on ONE K = 64 slab, outside point_fused2_kernel: the same conversions, block scales, position maps and
// matrix instructions as the kernel's MX-FP6 path (activations: lane = (row j, K half hh), 32 values = two row tiles x 16 accumulator registers; weights packed on the host
// with the maps of pack_point_mx6_kernel), against the exact product.  Prints the error of hi.hi alone, + each cross term, + both.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static int kmap(int s, int t, int hh) { return 16 * s + 8 * hh + t; }   // k of (k-step s, element t, half hh)

static unsigned e2m3(float a) {
  if (!(a < 7.5f)) return 31u;
  if (a < 1.f) return (unsigned)rintf(a * 8.f);
  const int e = a < 2.f ? 0 : a < 4.f ? 1 : 2;
  unsigned m = (unsigned)rintf(ldexpf(a, 3 - e));
  unsigned c = ((unsigned)(e + 1) << 3) + (m - 8u);
  return c > 31u ? 31u : c;
}

// act: [32 rows j][64 k]; wf16: [lane][k-step 4][8] f16 bits; w6: [image 2][lane][6]; wsc: [lane][2]; out: [mode 4][lane][16]
__global__ void slab_kernel(const float* act, const unsigned short* wf16, const unsigned* w6, const int* wsc, float* out) {
  const int l = threadIdx.x, j = l & 31, hh = l >> 5;
  unsigned hp[2][8]; float lo[2][16]; float amax = 0.f;
  for (int par = 0; par < 2; ++par)
    for (int p = 0; p < 8; ++p) {
      const int r0 = 2 * p, r1 = 2 * p + 1;
      float v0 = act[j * 64 + (16 * (2 * par + (r0 >> 3)) + 8 * hh + (r0 & 7))], v1 = act[j * 64 + (16 * (2 * par + (r1 >> 3)) + 8 * hh + (r1 & 7))];
      unsigned hi;
      asm volatile("v_max3_f32 %3, |%1|, |%2|, %3\n\tv_cvt_pk_f16_f32 %0, %1, %2" : "=&v"(hi), "+v"(v0), "+v"(v1), "+v"(amax));
      hp[par][p] = hi;
      asm volatile("v_fma_mix_f32 %0, %4, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %4, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                   : "=&v"(lo[par][r0]), "=&v"(lo[par][r1]) : "v"(v0), "v"(v1), "v"(hi));
    }
  int eb = __builtin_amdgcn_frexp_expf(amax) + 124;
  eb = eb < 12 ? 12 : (eb > 254 ? 254 : eb);
  const float scf = __builtin_bit_cast(float, eb << 23);
  const u32x16 H = {hp[0][0], hp[0][1], hp[0][2], hp[0][3], hp[0][4], hp[0][5], hp[0][6], hp[0][7], hp[1][0], hp[1][1], hp[1][2], hp[1][3], hp[1][4], hp[1][5], hp[1][6], hp[1][7]};
  const u32x6 h6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, H), scf);
  f32x16 l0, l1;
  for (int i = 0; i < 16; ++i) { l0[i] = lo[0][i]; l1[i] = lo[1][i]; }
  const u32x6 l6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(l0, l1, scf * 0.00048828125f);
  for (int mode = 0; mode < 4; ++mode) {   // 0: hi.hi, 1: + w_hi6 x a_lo6, 2: + w_lo6 x a_hi6, 3: both
    f32x16 acc = {};
    for (int s = 0; s < 4; ++s) {
      f16x8 a;
      for (int t = 0; t < 8; ++t) a[t] = __builtin_bit_cast(_Float16, wf16[(l * 4 + s) * 8 + t]);
      const u32x4 bq = s == 0 ? u32x4{H[0], H[1], H[2], H[3]} : s == 1 ? u32x4{H[4], H[5], H[6], H[7]} : s == 2 ? u32x4{H[8], H[9], H[10], H[11]} : u32x4{H[12], H[13], H[14], H[15]};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, bq), acc, 0, 0, 0);
    }
    if (mode & 1) {
      const i32x8 wa = {(int)w6[(0 * 64 + l) * 6], (int)w6[(0 * 64 + l) * 6 + 1], (int)w6[(0 * 64 + l) * 6 + 2], (int)w6[(0 * 64 + l) * 6 + 3], (int)w6[(0 * 64 + l) * 6 + 4], (int)w6[(0 * 64 + l) * 6 + 5], 0, 0};
      const i32x8 xb = {(int)l6[0], (int)l6[1], (int)l6[2], (int)l6[3], (int)l6[4], (int)l6[5], 0, 0};
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, xb, acc, 2, 2, 0, wsc[2 * l], 0, eb - 11);
    }
    if (mode & 2) {
      const i32x8 wa = {(int)w6[(1 * 64 + l) * 6], (int)w6[(1 * 64 + l) * 6 + 1], (int)w6[(1 * 64 + l) * 6 + 2], (int)w6[(1 * 64 + l) * 6 + 3], (int)w6[(1 * 64 + l) * 6 + 4], (int)w6[(1 * 64 + l) * 6 + 5], 0, 0};
      const i32x8 xb = {(int)h6[0], (int)h6[1], (int)h6[2], (int)h6[3], (int)h6[4], (int)h6[5], 0, 0};
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, xb, acc, 2, 2, 0, wsc[2 * l + 1], 0, eb);
    }
    for (int i = 0; i < 16; ++i) out[(mode * 64 + l) * 16 + i] = acc[i];
  }
}

int main() {
  static float act[32 * 64], wt[32 * 64];
  srand(11);
  for (int i = 0; i < 32 * 64; ++i) { act[i] = ((rand() % 20001) - 10000) * 1e-4f * ((i % 7) == 0 ? 0.01f : 1.f); wt[i] = ((rand() % 20001) - 10000) * 6e-6f; }
  static unsigned short wf16[64 * 4 * 8]; static unsigned w6[2 * 64 * 6]; static int wsc[128];
  for (int l = 0; l < 64; ++l) {
    const int n = l & 31, hh = l >> 5;
    for (int s = 0; s < 4; ++s)
      for (int t = 0; t < 8; ++t) wf16[(l * 4 + s) * 8 + t] = __builtin_bit_cast(unsigned short, (_Float16)wt[n * 64 + kmap(s, t, hh)]);
    for (int im = 0; im < 2; ++im) {
      float v[32], mx = 0.f;
      for (int P = 0; P < 32; ++P) {
        const int sI = im == 0 ? 2 * (P & 1) + (P >> 4) : (P >> 3), t = im == 0 ? (P >> 1) & 7 : (P & 7);
        const float w = wt[n * 64 + kmap(sI, t, hh)], h = (float)(_Float16)w;
        v[P] = im == 0 ? h : w - h;
        mx = fmaxf(mx, fabsf(v[P]));
      }
      int E = -60;
      if (mx > 0.f) { int ex; (void)frexpf(mx, &ex); E = ex - 1; }
      int sb = E - 2 + 127; sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
      const float inv = ldexpf(1.f, 127 - sb);
      unsigned d[6] = {0, 0, 0, 0, 0, 0};
      for (int P = 0; P < 32; ++P) {
        const unsigned c = e2m3(fabsf(v[P]) * inv) | (v[P] < 0.f ? 32u : 0u);
        const int b = 6 * P;
        d[b >> 5] |= c << (b & 31);
        if ((b & 31) > 26) d[(b >> 5) + 1] |= c >> (32 - (b & 31));
      }
      for (int i = 0; i < 6; ++i) w6[(im * 64 + l) * 6 + i] = d[i];
      wsc[2 * l + im] = sb;
    }
  }
  float *da, *dout; unsigned short* dw; unsigned* d6; int* ds;
  (void)hipMalloc(&da, sizeof(act)); (void)hipMalloc(&dw, sizeof(wf16)); (void)hipMalloc(&d6, sizeof(w6)); (void)hipMalloc(&ds, sizeof(wsc)); (void)hipMalloc(&dout, 4 * 64 * 16 * 4);
  (void)hipMemcpy(da, act, sizeof(act), hipMemcpyHostToDevice); (void)hipMemcpy(dw, wf16, sizeof(wf16), hipMemcpyHostToDevice);
  (void)hipMemcpy(d6, w6, sizeof(w6), hipMemcpyHostToDevice); (void)hipMemcpy(ds, wsc, sizeof(wsc), hipMemcpyHostToDevice);
  slab_kernel<<<1, 64>>>(da, dw, d6, ds, dout);
  static float ho[4 * 64 * 16]; (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  const char* names[4] = {"hi.hi alone", "+ w_hi6 x a_lo6", "+ w_lo6 x a_hi6", "+ both"};
  for (int mode = 0; mode < 4; ++mode) {
    double worst = 0, big = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;   // D[n][j]: row n = weight row, column j = activation row
        double e = 0;
        for (int k = 0; k < 64; ++k) e += (double)wt[n * 64 + k] * (double)act[j * 64 + k];
        worst = fmax(worst, fabs(e - ho[(mode * 64 + l) * 16 + r])); big = fmax(big, fabs(e));
      }
    printf("%-18s worst |error| / max |D| = %.3g\n", names[mode], worst / big);
  }
  return 0;
}
