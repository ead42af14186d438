// Micro-benchmark: do MFMA and ordinary VALU instructions of two DIFFERENT waves on the same SIMD overlap?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) issue back-to-back v_mfma_f32_32x32x16_bf16,
// waves 4-7 (their SIMD partners) issue dependent-free v_fma_f32.  Run each half alone and both together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, int mfma_iters, int valu_iters, int mode) {   // mode bit0: mfma waves work, bit1: valu waves work
  const int wave = threadIdx.x >> 6;
  float res = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r);
      bf16x8 a, b;
      for (int t = 0; t < 8; ++t) { a[t] = (__bf16)(0.001f * threadIdx.x + t); b[t] = (__bf16)(0.002f * threadIdx.x - t); }
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 12; ++rep)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    }
  } else if (mode & 2) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * threadIdx.x + i;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 12; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], m, c);
    }
    for (int i = 0; i < 16; ++i) res += x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

float run(float* out, int mi, int vi, int mode) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<<<256, 512>>>(out, mi / 10, vi / 10, mode);
  (void)hipEventRecord(e0);
  k<<<256, 512>>>(out, mi, vi, mode);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int mi = 4000;               // 4000 x 48 MFMAs x 32 cycles = 6.1 M cycles
  const int vi = 8000;               // 8000 x 192 v_fma x 4 cycles  = 6.1 M cycles
  const float tm = run(out, mi, vi, 1), tv = run(out, mi, vi, 2), tb = run(out, mi, vi, 3);
  printf("MFMA waves alone %.2f ms | VALU waves alone %.2f ms | both (same SIMDs) %.2f ms  -> %s\n", tm, tv, tb,
         tb < 0.75f * (tm + tv) ? "they overlap" : "they serialise");
  return 0;
}
