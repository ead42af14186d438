// Micro-benchmark: issue interval of v_mfma_f32_32x32x16_bf16 as a function of how many independent accumulators a single
// wave rotates over (1 = fully dependent chain).  One wave per SIMD (256 threads per block, 1 block per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, long long* cyc) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r);
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (__bf16)(0.001f * threadIdx.x + t); b[t] = (__bf16)(0.002f * threadIdx.x - t); }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 48 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int WAVES>
void run(float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, WAVES><<<256, 64 * WAVES>>>(out, 10, cyc);
  hipEventRecord(e0);
  k<NACC, WAVES><<<256, 64 * WAVES>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 48;
  const double tf = 256.0 * WAVES * n * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  printf("NACC=%d waves/CU=%d: %.1f ns per MFMA per wave (%.0f counter ticks/MFMA), %.0f TFLOP/s\n", NACC, WAVES, ms * 1e6 / n, (double)c / n, tf);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<1, 4>(out, cyc); run<2, 4>(out, cyc); run<3, 4>(out, cyc); run<4, 4>(out, cyc); run<8, 4>(out, cyc);
  run<1, 8>(out, cyc); run<2, 8>(out, cyc); run<4, 8>(out, cyc); run<8, 8>(out, cyc);
  return 0;
}
