// Micro-benchmark: how many single-issue instructions of the SAME wave hide in the 32-cycle issue gap of
// v_mfma_f32_32x32x16_bf16 when one wave owns a SIMD (the regime of point_fused_kernel)?
// One 256-thread workgroup per CU (64 KB of LDS keeps a second one out), every wave runs ITERS x 16 MFMAs with F fillers
// hand-placed (volatile inline asm keeps program order) after each MFMA.
//   filler kinds: 0 v_fma_f32 (independent chains)   1 v_cvt_pk_bf16_f32   2 ds_read_b128   3 mix: per MFMA {ds_read_b128, cvt_pk, fma...}
//   acc modes:    0 every MFMA on the same accumulator (dependent chain)   1 two accumulators alternating   2 four accumulators
// Prints shader cycles per MFMA (s_memtime) and the wall-clock rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void filler(int i, float (&x)[8], unsigned (&pk)[4], f32x4 (&ld)[4], unsigned lds_addr) {
  if (KIND == 0 || (KIND == 3 && (i % 3) == 2)) {
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(1.0001f), "v"(0.5f));
  } else if (KIND == 1 || (KIND == 3 && (i % 3) == 1)) {
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i & 3]) : "v"(x[i & 7]), "v"(x[(i + 1) & 7]));
  } else {
    asm volatile("ds_read_b128 %0, %1" : "=v"(ld[i & 3]) : "v"(lds_addr + 1024u * (i & 3)));
  }
}

template <int F, int KIND, int ACCM>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r);
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (__bf16)(0.001f * threadIdx.x + t); b[t] = (__bf16)(0.002f * threadIdx.x - t); }
  float x[8]; unsigned pk[4] = {0, 0, 0, 0}; f32x4 ld[4];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * threadIdx.x + i;
  for (int i = 0; i < 4; ++i) ld[i] = f32x4{0, 0, 0, 0};
  const unsigned lds_addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      constexpr int NA = ACCM == 0 ? 1 : (ACCM == 1 ? 2 : 4);
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % NA]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < F; ++f) filler<KIND>(m * F + f, x, pk, ld, lds_addr);
    }
    if (KIND >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
  for (int i = 0; i < 8; ++i) res += x[i];
  for (int i = 0; i < 4; ++i) res += __uint_as_float(pk[i]) + ld[i][0] + ld[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int F, int KIND, int ACCM>
void run(float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<F, KIND, ACCM><<<256, 256>>>(out, cyc, iters / 10);
  (void)hipEventRecord(e0);
  k<F, KIND, ACCM><<<256, 256>>>(out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = (double)iters * 16;
  printf("fill=%2d kind=%d acc=%d : %7.2f counter-ticks/MFMA  %7.3f ms  -> %6.1f ns/MFMA  (%.2f PFLOP/s chip)\n", F, KIND, ACCM, c / nm, ms,
         ms * 1e6 / nm, 256.0 * 4 * nm * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e15);
}

template <int KIND, int ACCM>
void sweep(float* out, long long* cyc) {
  run<0, KIND, ACCM>(out, cyc); run<2, KIND, ACCM>(out, cyc); run<4, KIND, ACCM>(out, cyc); run<5, KIND, ACCM>(out, cyc);
  run<6, KIND, ACCM>(out, cyc); run<8, KIND, ACCM>(out, cyc); run<12, KIND, ACCM>(out, cyc);
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  sweep<0, 0>(out, cyc); sweep<0, 1>(out, cyc); sweep<0, 2>(out, cyc);
  sweep<1, 0>(out, cyc); sweep<1, 1>(out, cyc);
  sweep<2, 0>(out, cyc); sweep<2, 1>(out, cyc);
  sweep<3, 0>(out, cyc); sweep<3, 1>(out, cyc);
  return 0;
}
