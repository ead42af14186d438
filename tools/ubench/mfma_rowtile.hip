// Micro-benchmark: the instruction mix of ONE output row tile of the output-stationary point kernel (bf16x3):
//   48 dependent-chain MFMAs + 32 ds_read_b128 (A fragments) + 8 LDS-DMA pieces (next chunk) + the previous tile's epilogue
//   (16 v_mul, 16 v_max, 8 cvt_pk hi, 16 unpack, 16 v_sub, 8 cvt_pk lo = 80 VALU) hand-interleaved per k-step group of 3 MFMAs.
// Knock-out flags (bit mask): 1 = ds_reads, 2 = DMA, 4 = epilogue VALU.  Prints shader cycles per MFMA for every subset.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MASK, int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, const char* wsrc, int iters) {
  __shared__ float lds[4 * 8192];   // 4 x 32 KB ring
  for (int i = threadIdx.x; i < 4 * 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r);
  bf16x8 b;
  for (int t = 0; t < 8; ++t) b[t] = (__bf16)(0.002f * threadIdx.x - t);
  float x[16]; unsigned pk[8];
  for (int i = 0; i < 16; ++i) x[i] = 0.001f * threadIdx.x + i;
  for (int i = 0; i < 8; ++i) pk[i] = 0;
  f32x4 afr[4];
  for (int i = 0; i < 4; ++i) afr[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds_base = (unsigned)(size_t)lds;
  const unsigned lds_addr = lds_base + lane * 16;
  const char* gp = wsrc + wave * 8192 + lane * 16;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned slot = (it & 3) * 32768u;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      // A fragments of this k-step (hi, lo): read two groups ahead in the real kernel; here just issued in place
      if (MASK & 1) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(afr[(2 * ks) & 3]) : "v"(lds_addr + slot + 2048u * (ks & 7)));
        asm volatile("ds_read_b128 %0, %1" : "=v"(afr[(2 * ks + 1) & 3]) : "v"(lds_addr + slot + 2048u * (ks & 7) + 1024u));
      }
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(afr[0]), "v"(b));
      if (MASK & 4) {   // 5 VALU: 2 mul, 2 max, 1 cvt  (even ks) | 2 unpack, 2 sub, 1 cvt (odd ks)
        const int p = ks >> 1;
        if ((ks & 1) == 0) {
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[2 * p]) : "v"(x[(2 * p + 2) & 15]), "v"(0.01f));
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[2 * p + 1]) : "v"(x[(2 * p + 3) & 15]), "v"(0.01f));
        } else {
          asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(x[2 * p]) : "v"(pk[p]));
          asm volatile("v_and_b32 %0, %1, %2" : "=v"(x[2 * p + 1]) : "v"(0xffff0000u), "v"(pk[p]));
        }
      }
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[NACC > 1 ? 1 : 0]) : "v"(afr[1]), "v"(b));
      if (MASK & 4) {
        const int p = ks >> 1;
        if ((ks & 1) == 0) {
          asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[2 * p]) : "v"(x[(2 * p + 4) & 15]));
          asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[2 * p + 1]) : "v"(x[(2 * p + 5) & 15]));
        } else {
          asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[2 * p]) : "v"(x[(2 * p + 6) & 15]));
          asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[2 * p + 1]) : "v"(x[(2 * p + 7) & 15]));
        }
      }
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(afr[2]), "v"(b));
      if (MASK & 4) {
        const int p = ks >> 1;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[p]) : "v"(x[2 * p]), "v"(x[2 * p + 1]));
      }
      if ((MASK & 2) && (ks & 1)) {   // one DMA piece per two k-steps: 8 per row tile, into the slot after next
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_base + ((slot + 65536u) & 131071u) + wave * 8192u + (ks >> 1) * 1024u)));
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp + (ks >> 1) * 1024) : "memory");
      }
    }
    if (MASK & 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (MASK & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
  for (int i = 0; i < 16; ++i) res += x[i];
  for (int i = 0; i < 8; ++i) res += __uint_as_float(pk[i]);
  for (int i = 0; i < 4; ++i) res += afr[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MASK, int NACC>
void run(float* out, long long* cyc, const char* w) {
  const int iters = 1000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MASK, NACC><<<256, 256>>>(out, cyc, w, iters / 10);
  (void)hipEventRecord(e0);
  k<MASK, NACC><<<256, 256>>>(out, cyc, w, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = (double)iters * 48;
  printf("ds_read=%d dma=%d valu=%d acc=%d : %6.2f cycles/MFMA  %7.3f ms  (%.2f PFLOP/s chip)\n", MASK & 1, (MASK >> 1) & 1, (MASK >> 2) & 1, NACC,
         c / nm, ms, 256.0 * 4 * nm * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e15);
}

int main() {
  float* out; long long* cyc; char* w;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20);
  run<0, 1>(out, cyc, w); run<1, 1>(out, cyc, w); run<2, 1>(out, cyc, w); run<4, 1>(out, cyc, w);
  run<3, 1>(out, cyc, w); run<5, 1>(out, cyc, w); run<6, 1>(out, cyc, w); run<7, 1>(out, cyc, w);
  run<0, 2>(out, cyc, w); run<7, 2>(out, cyc, w);
  return 0;
}
