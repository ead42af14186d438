// Micro-benchmark: how fast can every CU stream the SAME L2-resident buffer into LDS?  (the weight stream of the fused
// kernels: 864 KB per 128-row tile).  Variants: LDS-DMA (global_load_lds_dwordx4) vs register-staged (global_load_dwordx4
// + ds_write_b128), 4 or 8 waves per workgroup, one workgroup per CU, ring of 4 x 32 KB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int WAVES, bool DMA>
__global__ __launch_bounds__(64 * WAVES) void fill(const uint4* __restrict__ src, int chunks, int reps, float* out) {
  __shared__ uint4 ring[4][2048];   // 4 x 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PW = 32 / WAVES;   // 1-KB pieces per wave per chunk
  uint4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (int c = 0; c < chunks; ++c) {
      const uint4* s = src + (size_t)c * 2048;
      if (DMA) {
#pragma unroll
        for (int j = 0; j < PW; ++j) { const int i = wave + WAVES * j; glds16(s + i * 64 + lane, &ring[c & 3][i * 64]); }
        if ((c & 3) == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc.x += ring[c & 3][tid].x; }
      } else {
        uint4 v[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) v[j] = s[(wave + WAVES * j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < PW; ++j) ring[c & 3][(wave + WAVES * j) * 64 + lane] = v[j];
        if ((c & 3) == 3) { __syncthreads(); acc.x += ring[c & 3][tid].x; }
      }
    }
  }
  if (acc.x == 0x12345678u) out[blockIdx.x] = 1.f;
}

template <int WAVES, bool DMA>
void run(const uint4* src, float* out, const char* name) {
  const int chunks = 27, reps = 200;
  fill<WAVES, DMA><<<256, 64 * WAVES>>>(src, chunks, 2, out);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  fill<WAVES, DMA><<<256, 64 * WAVES>>>(src, chunks, reps, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * chunks * reps * 32768.0;
  printf("%-28s %7.2f ms  %6.1f GB/s per CU  %5.2f TB/s chip\n", name, ms, bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  uint4* src; float* out;
  (void)hipMalloc(&src, 27 * 32768); (void)hipMemset(src, 1, 27 * 32768); (void)hipMalloc(&out, 4096);
  run<4, true>(src, out, "LDS-DMA, 4 waves");
  run<8, true>(src, out, "LDS-DMA, 8 waves");
  run<4, false>(src, out, "register-staged, 4 waves");
  run<8, false>(src, out, "register-staged, 8 waves");
  return 0;
}
