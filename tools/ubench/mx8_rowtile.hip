// Micro-benchmark + semantics probe for the "fp16 hi.hi + two MX-FP8 cross terms" arithmetic of the output-stationary point kernel.
//   A. conversion semantics: does MODE.FP16_OVFL make v_cvt_scalef32_pk_fp8_{f16,f32} saturate instead of producing NaN?  which half does op_sel write?
//   B. the split itself on random data: hi = f16(x), lo = x - hi via v_fma_mix_f32, fp8 images, reconstruction error of x.w products
//   C. the instruction mix of ONE output row tile (K = 256) in that arithmetic — 16 v_mfma_f32_32x32x16_f16 + 8 v_mfma_scale_f32_32x32x64_f8f6f4 (fp8) +
//      32 ds_read_b128 (the same 32 KB of A fragments as bf16x3) + 8 LDS-DMA pieces + the previous tile's epilogue (per pair: 2 mul, 2 max, cvt_pk_f16 |
//      2 fma_mix, 2 cvt_scalef32_pk_fp8 = 9 VALU x 8 pairs) — against tools/ubench/mfma_rowtile.hip's 48 bf16 MFMAs with the same side work (37.7 cycles per MFMA)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static float e4m3_decode(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m / 8.f, -6);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -v : v;
}

__global__ void sem_kernel(const float* x, unsigned* o, int ovfl) {
  const int l = threadIdx.x;
  if (ovfl) __builtin_amdgcn_s_setreg(1473, 1);   // hwreg(HW_REG_MODE, 23, 1): FP16_OVFL
  const float v0 = x[2 * l], v1 = x[2 * l + 1];
  unsigned hi, lo8 = 0xaaaaaaaau, hi8 = 0xaaaaaaaau, lo8b = 0xaaaaaaaau;
  float l0, l1;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(v0));
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(v1));
  asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(hi8) : "v"(hi), "v"(1.0f));
  asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(lo8) : "v"(l0), "v"(l1), "v"(0.00048828125f));
  asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(lo8b) : "v"(l0), "v"(l1), "v"(0.00048828125f));
  o[4 * l] = hi; o[4 * l + 1] = hi8; o[4 * l + 2] = lo8; o[4 * l + 3] = lo8b;
  __builtin_amdgcn_s_setreg(1473, 0);
}

template <int MASK>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, const char* wsrc, int iters) {
  __shared__ float lds[4 * 8192];   // 4 x 32 KB ring
  for (int i = threadIdx.x; i < 4 * 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = (float)(threadIdx.x + r);
  f16x8 b;
  for (int t = 0; t < 8; ++t) b[t] = (_Float16)(0.002f * threadIdx.x - t);
  v8i b8;
  for (int t = 0; t < 8; ++t) b8[t] = 0x38303438 + threadIdx.x * 3 + t;
  float x[16]; unsigned pk[8], q8[8];
  for (int i = 0; i < 16; ++i) x[i] = 0.001f * threadIdx.x + i;
  for (int i = 0; i < 8; ++i) { pk[i] = 0; q8[i] = 0; }
  f32x4 afr[8];
  for (int i = 0; i < 8; ++i) afr[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds_base = (unsigned)(size_t)lds;
  const unsigned lds_addr = lds_base + lane * 16;
  const char* gp = wsrc + wave * 8192 + lane * 16;
  const int sA = 127, sB = 116;
  const float one = 1.0f, sc11 = 0.00048828125f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned slot = (it & 3) * 32768u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // slabs of K = 64
      // A fragments of this slab: 4 f16 k-steps + 2 x 32 B of fp8 (w_hi8, w_lo8) = 8 ds_read_b128
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (MASK & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(afr[s]) : "v"(lds_addr + slot + 8192u * q + 1024u * s));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(afr[s]), "v"(b));
        if (MASK & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(afr[4 + s]) : "v"(lds_addr + slot + 8192u * q + 4096u + 1024u * s));
        if (MASK & 4) {   // hi step of pair p = 2 q + (s >> 1) on even s (5 VALU), lo step on odd s (4 VALU)
          const int p = 2 * q + (s >> 1);
          if ((s & 1) == 0) {
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[2 * p]) : "v"(x[(2 * p + 2) & 15]), "v"(0.01f));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[2 * p + 1]) : "v"(x[(2 * p + 3) & 15]), "v"(0.01f));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[2 * p]) : "v"(x[(2 * p + 4) & 15]));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[2 * p + 1]) : "v"(x[(2 * p + 5) & 15]));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[p]) : "v"(x[2 * p]), "v"(x[2 * p + 1]));
          } else {
            float l0, l1;
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(pk[p]), "v"(x[2 * p]));
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(pk[p]), "v"(x[2 * p + 1]));
            asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(q8[p]) : "v"(pk[p]), "v"(one));
            asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(q8[(p + 1) & 7]) : "v"(l0), "v"(l1), "v"(sc11));
          }
        }
      }
      {
        v8i a8 = {__builtin_bit_cast(int, afr[4][0]), __builtin_bit_cast(int, afr[4][1]), __builtin_bit_cast(int, afr[4][2]), __builtin_bit_cast(int, afr[4][3]),
                  __builtin_bit_cast(int, afr[5][0]), __builtin_bit_cast(int, afr[5][1]), __builtin_bit_cast(int, afr[5][2]), __builtin_bit_cast(int, afr[5][3])};
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a8), "v"(b8), "v"(sA), "v"(sB));
      }
      if (MASK & 2) {
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_base + ((slot + 65536u) & 131071u) + wave * 8192u + (2 * q) * 1024u)));
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp + (2 * q) * 1024) : "memory");
      }
      {
        v8i a8 = {__builtin_bit_cast(int, afr[6][0]), __builtin_bit_cast(int, afr[6][1]), __builtin_bit_cast(int, afr[6][2]), __builtin_bit_cast(int, afr[6][3]),
                  __builtin_bit_cast(int, afr[7][0]), __builtin_bit_cast(int, afr[7][1]), __builtin_bit_cast(int, afr[7][2]), __builtin_bit_cast(int, afr[7][3])};
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a8), "v"(b8), "v"(sB), "v"(sA));
      }
      if (MASK & 2) {
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_base + ((slot + 65536u) & 131071u) + wave * 8192u + (2 * q + 1) * 1024u)));
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp + (2 * q + 1) * 1024) : "memory");
      }
    }
    if (MASK & 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (MASK & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int r = 0; r < 16; ++r) res += acc[r];
  for (int i = 0; i < 16; ++i) res += x[i];
  for (int i = 0; i < 8; ++i) res += __uint_as_float(pk[i]) + __uint_as_float(q8[i]);
  for (int i = 0; i < 8; ++i) res += afr[i][0];
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
  printf("ds_read=%d dma=%d valu=%d : %7.1f counter ticks per row tile (bf16x3 today: 48 x 37.7 = 1810)  %7.3f ms  = %.2f PFLOP/s algorithmic (32x32x256 MACs per tile)\n", MASK & 1,
         (MASK >> 1) & 1, (MASK >> 2) & 1, (double)c / iters, ms, 256.0 * 4 * iters * 32 * 32 * 256 * 2 / (ms * 1e-3) / 1e15);
}

int main() {
  // ---- A / B: semantics
  {
    const int n = 128;
    float hx[n];
    const float special[12] = {500.f, 1000.f, 70000.f, -600.f, 448.f, 449.f, 1e-3f, 3.14159f, -0.007f, 100.25f, 0.3333f, -17.77f};
    for (int i = 0; i < n; ++i) hx[i] = i < 12 ? special[i] : (float)((i * 37 % 101) - 50) * 0.137f + 0.001f * i;
    float* dx; unsigned* dout; (void)hipMalloc(&dx, sizeof(hx)); (void)hipMalloc(&dout, 64 * 16);
    (void)hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
      sem_kernel<<<1, 64>>>(dx, dout, ovfl);
      unsigned ho[256]; (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
      printf("FP16_OVFL=%d:\n", ovfl);
      double worst = 0;
      for (int l = 0; l < 64; ++l) {
        const unsigned hi = ho[4 * l], hi8 = ho[4 * l + 1], lo8 = ho[4 * l + 2], lo8b = ho[4 * l + 3];
        for (int e = 0; e < 2; ++e) {
          const float xv = hx[2 * l + e];
          const unsigned short hb = (unsigned short)(hi >> (16 * e));
          const float hf = (float)__builtin_bit_cast(_Float16, hb);
          const float h8 = e4m3_decode((unsigned char)(hi8 >> (8 * e))), l8 = e4m3_decode((unsigned char)(lo8 >> (8 * e))) * 0.00048828125f;
          const float l8b = e4m3_decode((unsigned char)(lo8b >> (16 + 8 * e))) * 0.00048828125f;
          if (l < 6) printf("  x=%-10g f16 hi=%-10g fp8(hi)=%-8g fp8(lo)*2^-11=%-12g (op_sel hi half: %-12g; untouched halves %04x %04x)  x-hi=%g\n", xv, hf, h8, l8, l8b, hi8 >> 16, lo8b & 0xffff,
                           xv - hf);
          else if (fabsf(xv) < 400.f) worst = fmax(worst, fabs((double)(xv - hf) - l8) / fmax(fabs((double)xv), 1e-3));
        }
      }
      printf("  worst |(x - hi) - fp8(lo)| / |x| over the ordinary values: %.3g (2^-15 = %.3g)\n", worst, ldexp(1.0, -15));
    }
  }
  float* out; long long* cyc; char* w;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20);
  run<0>(out, cyc, w); run<1>(out, cyc, w); run<2>(out, cyc, w); run<4>(out, cyc, w); run<3>(out, cyc, w); run<5>(out, cyc, w); run<7>(out, cyc, w);
  return 0;
}
