// Micro-benchmark + semantics probe for "fp16 hi.hi + two MX-FP6 (e2m3) cross terms" (round 5): the f16mx arithmetic of point_fused2_kernel with the cross terms
// moved from fp8 (16 passes per K = 64) to fp6 (8 passes): 1 + 2 x 0.25 = 1.5 MFMA-equivalents per product instead of 2.0, 28 instead of 32 KB of weight images per
// row tile, and ONE conversion instruction per 32 values and image (v_cvt_scalef32_pk32_fp6_f16 / v_cvt_scalef32_2xpk16_fp6_f32) instead of 16.
//   A. conversion semantics: element order of the two packing conversions, rounding, saturation, the scale operand
//   B. the matrix instruction with fp6 operands against a host product (lane / K layout, per-lane E8M0 scales through op_sel bytes)
//   C. the instruction mix of ONE output row tile (K = 256): 16 v_mfma_f32_32x32x16_f16 + 8 fp6 v_mfma_scale (32 cycles each) + 16 ds_read_b128 (f16 fragments) +
//      8 x (ds_read_b128 + ds_read_b64) (fp6 images) + 7 LDS-DMA pieces + the previous tile's epilogue (per pair: 2 mul, 2 max, max3, cvt_pk_f16 | 2 fma_mix; per two
//      tiles: the block scale (5 VALU) and the two packing conversions) — against tools/ubench/mx8_rowtile.hip (fp8 cross terms: 1 556 ticks with everything on)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef unsigned v16u __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

static float e2m3_decode(unsigned c) {
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? (float)m / 8.f : ldexpf(1.f + (float)m / 8.f, e - 1);
  return s ? -v : v;
}
static unsigned field6(const unsigned* w, int p) {   // 6-bit field p of a 192-bit little-endian value
  const int b = 6 * p;
  unsigned long long lo = w[b >> 5];
  if ((b >> 5) + 1 < 6) lo |= (unsigned long long)w[(b >> 5) + 1] << 32;
  return (unsigned)(lo >> (b & 31)) & 63u;
}

// ---- A: conversions.  x: 64 lanes x 32 floats; out: [lane][12] = 2xpk16_f32 result | pk32_f16 result
__global__ void cvt_kernel(const float* x, unsigned* o, float scale) {
  const int l = threadIdx.x;
  f32x16 a, b;
  h32 h;
  for (int i = 0; i < 16; ++i) { a[i] = x[l * 32 + i]; b[i] = x[l * 32 + 16 + i]; }
  for (int i = 0; i < 32; ++i) h[i] = (_Float16)x[l * 32 + i];
  const v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
  const v6u r2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, scale);
  for (int i = 0; i < 6; ++i) { o[l * 12 + i] = r[i]; o[l * 12 + 6 + i] = r2[i]; }
}

// ---- B: one matrix instruction, fp6 x fp6.  a, b: [lane][6] dwords; sc: [lane][2] E8M0 bytes (A side, B side), read from byte 1 of the scale VGPR (op_sel = 1)
template <int SEL>
__global__ void opsel_kernel(const unsigned* a, const unsigned* b, float* o) {   // scale dwords 0x7c7d7e7f: which byte does op_sel = SEL read?  (result scales by 2^(byteA - 127 + byteB - 127))
  const int l = threadIdx.x;
  const v8i A = {(int)a[l * 6], (int)a[l * 6 + 1], (int)a[l * 6 + 2], (int)a[l * 6 + 3], (int)a[l * 6 + 4], (int)a[l * 6 + 5], 0, 0};
  const v8i B = {(int)b[l * 6], (int)b[l * 6 + 1], (int)b[l * 6 + 2], (int)b[l * 6 + 3], (int)b[l * 6 + 4], (int)b[l * 6 + 5], 0, 0};
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 2, 2, SEL, 0x7c7d7e7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 16; ++i) o[l * 16 + i] = c[i];
}
__global__ void mfma_kernel(const unsigned* a, const unsigned* b, const int* sc, float* o) {
  const int l = threadIdx.x;
  const v8i A = {(int)a[l * 6], (int)a[l * 6 + 1], (int)a[l * 6 + 2], (int)a[l * 6 + 3], (int)a[l * 6 + 4], (int)a[l * 6 + 5], 0, 0};
  const v8i B = {(int)b[l * 6], (int)b[l * 6 + 1], (int)b[l * 6 + 2], (int)b[l * 6 + 3], (int)b[l * 6 + 4], (int)b[l * 6 + 5], 0, 0};
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 2, 2, 1, sc[2 * l] << 8 | 0x55, 1, sc[2 * l + 1] << 8 | 0x33);
  for (int i = 0; i < 16; ++i) o[l * 16 + i] = c[i];
}

// ---- C: timing
template <int MASK>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, const char* wsrc, int iters) {
  __shared__ float lds[4 * 8192];   // 4 x 32 KB ring
  for (int i = threadIdx.x; i < 4 * 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = (float)(threadIdx.x + r);
  f16x8 b;
  for (int t = 0; t < 8; ++t) b[t] = (_Float16)(0.002f * threadIdx.x - t);
  v6i b6;
  for (int t = 0; t < 6; ++t) b6[t] = 0x38303438 + threadIdx.x * 3 + t;
  float x[16], lo[16]; unsigned pk[8];
  for (int i = 0; i < 16; ++i) { x[i] = 0.001f * threadIdx.x + i; lo[i] = 0.f; }
  for (int i = 0; i < 8; ++i) pk[i] = 0;
  f32x16 cvA, cvB; v16u cvH; v6u c6a, c6b;
  for (int i = 0; i < 16; ++i) { cvA[i] = 0.01f * i + threadIdx.x; cvB[i] = 0.02f * i; cvH[i] = 0x3c003800u + i; }
  for (int i = 0; i < 6; ++i) { c6a[i] = 0; c6b[i] = 0; }
  float amax = 0.f, scf = 1.f; int e8 = 127, sc4 = 0;
  f32x4 afr[4]; i32x4 w6a[2]; i32x2 w6b[2];   // (int vectors: __builtin_bit_cast applied to an ext-vector ELEMENT reads element 0 whatever the index)
  for (int i = 0; i < 4; ++i) afr[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  for (int i = 0; i < 2; ++i) { w6a[i] = i32x4{1, 2, 3, 4}; w6b[i] = i32x2{1, 2}; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds_base = (unsigned)(size_t)lds;
  const unsigned lds_addr = lds_base + lane * 16, lds_addr8 = lds_base + lane * 8;
  const char* gp = wsrc + wave * 8192 + lane * 16;
  const int sA = 127, sB = 116;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned slot = (it & 3) * 32768u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // slabs of K = 64: 4 KB of f16 fragments + 3 KB of fp6 images per wave-read
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (MASK & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(afr[s]) : "v"(lds_addr + slot + 7168u * q + 1024u * s));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(afr[s]), "v"(b));
        if (MASK & 1) {
          if (s == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(w6a[0]) : "v"(lds_addr + slot + 7168u * q + 4096u));
          if (s == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(w6b[0]) : "v"(lds_addr8 + slot + 7168u * q + 5120u));
          if (s == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(w6a[1]) : "v"(lds_addr + slot + 7168u * q + 5632u));
          if (s == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(w6b[1]) : "v"(lds_addr8 + slot + 7168u * q + 6656u));
        }
        if (MASK & 4) {   // hi step of pair p = 2 q + (s >> 1) on even s (6 VALU), lo step on odd s (2 VALU)
          const int p = 2 * q + (s >> 1);
          if ((s & 1) == 0) {
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[2 * p]) : "v"(x[(2 * p + 2) & 15]), "v"(0.01f));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[2 * p + 1]) : "v"(x[(2 * p + 3) & 15]), "v"(0.01f));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[2 * p]) : "v"(x[(2 * p + 4) & 15]));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[2 * p + 1]) : "v"(x[(2 * p + 5) & 15]));
            asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(x[2 * p]), "v"(x[2 * p + 1]));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[p]) : "v"(x[2 * p]), "v"(x[2 * p + 1]));
          } else {
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[2 * p]) : "v"(pk[p]), "v"(x[2 * p]));
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[2 * p + 1]) : "v"(pk[p]), "v"(x[2 * p + 1]));
          }
        }
      }
      {
        const v6i a6 = {w6a[0][0], w6a[0][1], w6a[0][2], w6a[0][3], w6b[0][0], w6b[0][1]};
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(acc) : "v"(a6), "v"(b6), "v"(sA), "v"(sB));
      }
      if ((MASK & 2) && q < 3) {
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_base + ((slot + 65536u) & 131071u) + wave * 8192u + (2 * q) * 1024u)));
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp + (2 * q) * 1024) : "memory");
      }
      if ((MASK & 4) && (q & 1) == 1) {   // every second slab stands for "every second tile": half of the block-scale arithmetic + one packing conversion
        if (q == 1) {
          asm volatile("v_frexp_exp_i32_f32 %0, %1" : "=v"(e8) : "v"(amax));
          asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(e8) : "v"(-110), "v"(120));
          asm volatile("v_lshlrev_b32 %0, 23, %1" : "=v"(scf) : "v"(e8));
          asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=v"(c6a) : "v"(cvH), "v"(scf));
        } else {
          asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(sc4) : "v"(e8));
          asm volatile("v_mov_b32 %0, 0" : "=v"(amax));
          asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=v"(c6b) : "v"(cvA), "v"(cvB), "v"(scf));
        }
      }
      {
        const v6i a6 = {w6a[1][0], w6a[1][1], w6a[1][2], w6a[1][3], w6b[1][0], w6b[1][1]};
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(acc) : "v"(a6), "v"(b6), "v"(sB), "v"(sA));
      }
      if (MASK & 2) {
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_base + ((slot + 65536u) & 131071u) + wave * 8192u + (2 * q + 1) * 1024u)));
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp + (2 * q + 1) * 1024) : "memory");
      }
    }
    if (MASK & 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if (MASK & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_readcyclecounter();
  float res = amax + scf + (float)sc4;
  for (int r = 0; r < 16; ++r) res += acc[r] + lo[r];
  for (int i = 0; i < 16; ++i) res += x[i];
  for (int i = 0; i < 8; ++i) res += __uint_as_float(pk[i]);
  for (int i = 0; i < 6; ++i) res += __uint_as_float(c6a[i]) + __uint_as_float(c6b[i]);
  for (int i = 0; i < 4; ++i) res += afr[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MASK>
void run(float* out, long long* cyc, const char* w) {
  const int iters = 1000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MASK><<<256, 256>>>(out, cyc, w, iters / 10);
  (void)hipEventRecord(e0);
  k<MASK><<<256, 256>>>(out, cyc, w, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("ds_read=%d dma=%d valu=%d : %7.1f counter ticks per row tile (fp8 cross terms, everything on: 1 556; its matrix instructions alone: 1 024)  %7.3f ms  = %.2f PFLOP/s algorithmic\n",
         MASK & 1, (MASK >> 1) & 1, (MASK >> 2) & 1, (double)c / iters, ms, 256.0 * 4 * iters * 32 * 32 * 256 * 2 / (ms * 1e-3) / 1e15);
}

int main() {
  // ---- A: element order, rounding, saturation, scale
  {
    float hx[64 * 32];
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 32; ++i) {
        float v = e2m3_decode((unsigned)i);                       // lane 0: the 32 non-negative codes in order -> the position map
        if (l == 1) v = -v;                                       // lane 1: signs
        if (l == 2) v = e2m3_decode((unsigned)i) + (i & 1 ? 0.0625f : 0.03f);   // lane 2: ties (odd i: exactly between two codes below 2) and near-ties
        if (l == 3) v = 7.5f + 0.25f * i;                         // lane 3: saturation
        if (l >= 4) v = (float)((l * 131 + i * 37) % 97 - 48) * 0.11f;
        hx[l * 32 + i] = v;
      }
    float* dx; unsigned* dout; (void)hipMalloc(&dx, sizeof(hx)); (void)hipMalloc(&dout, 64 * 12 * 4);
    (void)hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
      const float scale = pass == 0 ? 1.0f : 2.0f;
      cvt_kernel<<<1, 64>>>(dx, dout, scale);
      unsigned ho[64 * 12]; (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
      printf("scale operand %g:\n", scale);
      for (int which = 0; which < 2; ++which) {
        const char* name = which == 0 ? "v_cvt_scalef32_2xpk16_fp6_f32" : "v_cvt_scalef32_pk32_fp6_f16 ";
        printf("  %s lane 0, code at position p:", name);
        for (int p = 0; p < 32; ++p) printf(" %u", field6(ho + which * 6, p));
        printf("\n  %s lane 1 (negated):          ", name);
        for (int p = 0; p < 32; ++p) printf(" %u", field6(ho + 12 + which * 6, p));
        printf("\n  %s lane 2 (ties)  in -> out: ", name);
        for (int p = 0; p < 8; ++p) printf(" %g->%g", hx[2 * 32 + p], e2m3_decode(field6(ho + 24 + which * 6, p)));
        printf("\n  %s lane 3 (large) in -> out: ", name);
        for (int p = 0; p < 6; ++p) printf(" %g->%g", hx[3 * 32 + p], e2m3_decode(field6(ho + 36 + which * 6, p)));
        // identity-order check on the random lanes: position p holds round(x[p] / scale)?
        int ident = 0, inter = 0, tot = 0;
        for (int l = 4; l < 64; ++l)
          for (int p = 0; p < 32; ++p) {
            const float got = e2m3_decode(field6(ho + l * 12 + which * 6, p)) * scale;
            const float xi = hx[l * 32 + p], xj = hx[l * 32 + ((p & 1) * 16 + (p >> 1))];
            ident += fabsf(got - fminf(fmaxf(xi, -7.5f * scale), 7.5f * scale)) <= 0.26f * scale;
            inter += fabsf(got - fminf(fmaxf(xj, -7.5f * scale), 7.5f * scale)) <= 0.26f * scale;
            ++tot;
          }
        printf("\n  %s random lanes: %d of %d positions consistent with the identity order, %d with the interleaved order (src0[i], src1[i] -> 2 i, 2 i + 1)\n", name, ident, tot, inter);
      }
    }
  }
  // ---- B: the matrix instruction with fp6 operands
  {
    unsigned ha[64 * 6] = {0}, hb[64 * 6] = {0}; int hs[128];
    static unsigned ca[32][64], cb[64][32];
    srand(7);
    for (int r = 0; r < 32; ++r) for (int kx = 0; kx < 64; ++kx) { ca[r][kx] = rand() & 63; cb[kx][r] = rand() & 63; }
    auto put = [](unsigned* w, int p, unsigned c) {
      const int bpos = 6 * p;
      w[bpos >> 5] |= c << (bpos & 31);
      if ((bpos & 31) > 26) w[(bpos >> 5) + 1] |= c >> (32 - (bpos & 31));
    };
    for (int l = 0; l < 64; ++l) {
      for (int t = 0; t < 32; ++t) { put(ha + l * 6, t, ca[l & 31][32 * (l >> 5) + t]); put(hb + l * 6, t, cb[32 * (l >> 5) + t][l & 31]); }
      hs[2 * l] = 120 + (l * 5) % 13; hs[2 * l + 1] = 125 + (l * 3) % 7;
    }
    unsigned *da, *db; int* ds; float* dc;
    (void)hipMalloc(&da, sizeof(ha)); (void)hipMalloc(&db, sizeof(hb)); (void)hipMalloc(&ds, sizeof(hs)); (void)hipMalloc(&dc, 64 * 16 * 4);
    (void)hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice); (void)hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(da, db, ds, dc);
    float hc[64 * 16]; (void)hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    double worst = 0, big = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        double e = 0;
        for (int kx = 0; kx < 64; ++kx)
          e += (double)e2m3_decode(ca[row][kx]) * ldexp(1.0, hs[2 * (row + 32 * (kx >> 5))] - 127) * (double)e2m3_decode(cb[kx][col]) * ldexp(1.0, hs[2 * (col + 32 * (kx >> 5)) + 1] - 127);
        worst = fmax(worst, fabs(e - hc[l * 16 + r])); big = fmax(big, fabs(e));
      }
    {
      float ref[64 * 16];
      float* dr; (void)hipMalloc(&dr, sizeof(ref));
      double sums[4];
      opsel_kernel<0><<<1, 64>>>(da, db, dr); (void)hipMemcpy(ref, dr, sizeof(ref), hipMemcpyDeviceToHost); sums[0] = 0; for (int i = 0; i < 1024; ++i) sums[0] += fabs(ref[i]);
      opsel_kernel<1><<<1, 64>>>(da, db, dr); (void)hipMemcpy(ref, dr, sizeof(ref), hipMemcpyDeviceToHost); sums[1] = 0; for (int i = 0; i < 1024; ++i) sums[1] += fabs(ref[i]);
      opsel_kernel<2><<<1, 64>>>(da, db, dr); (void)hipMemcpy(ref, dr, sizeof(ref), hipMemcpyDeviceToHost); sums[2] = 0; for (int i = 0; i < 1024; ++i) sums[2] += fabs(ref[i]);
      opsel_kernel<3><<<1, 64>>>(da, db, dr); (void)hipMemcpy(ref, dr, sizeof(ref), hipMemcpyDeviceToHost); sums[3] = 0; for (int i = 0; i < 1024; ++i) sums[3] += fabs(ref[i]);
      printf("scale dword 0x7c7d7e7f (bytes 0..3 = 2^0, 2^-1, 2^-2, 2^-3): op_sel 0 / 1 / 2 / 3 scale the result by %.4g / %.4g / %.4g / %.4g of op_sel 0's\n", 1.0, sums[1] / sums[0],
             sums[2] / sums[0], sums[3] / sums[0]);
    }
    printf("fp6 x fp6 matrix instruction (lane = row / column l & 31, K = 32 (l >> 5) + position, per-lane E8M0 scales from byte 1): worst |error| %.3g of max |C| %.3g\n", worst, big);
  }
  float* out; long long* cyc; char* w;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20);
  run<0>(out, cyc, w); run<1>(out, cyc, w); run<2>(out, cyc, w); run<4>(out, cyc, w); run<3>(out, cyc, w); run<5>(out, cyc, w); run<7>(out, cyc, w);
  return 0;
}
