// Micro-benchmark: what does a wave pay to ISSUE row-strided 16-byte vector memory instructions (lane = row, the access pattern of a
// transposed-MFMA kernel's B-operand loads and accumulator stores) compared with contiguous ones?  One workgroup of 4 waves per CU
// (one wave per SIMD), every wave issues `n` back-to-back instructions, optionally with `gap` MFMAs between two of them; prints shader
// cycles per instruction as seen by wave 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE /*0 strided store, 1 contiguous store, 2 strided load, 3 contiguous load*/, int GAP>
__global__ __launch_bounds__(256, 1) void k(float* buf, long long* cyc, float* sink, int iters, unsigned bytes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  // each wave works in its own 32-KB window per iteration: 32 rows x 1 KB; strided: lane (j, hh) -> row j, 16 B at column block
  const unsigned base = (blockIdx.x * 4 + wave) * 32768u;
  const unsigned off_s = base + (lane & 31) * 1024u + (lane >> 5) * 16u;
  const unsigned off_c = base + lane * 16u;
  f32x4 v{1.f, 2.f, 3.f, (float)lane};
  f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.f; b[i] = (__bf16)(float)lane; }
  f32x4 ld[32];
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const unsigned o = (MODE & 1) ? off_c + i * 1024u : off_s + i * 32u;
      if (MODE < 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, o, 0, 0);
      else ld[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0));
#pragma unroll
      for (int g = 0; g < GAP; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 32; ++i) v += ld[i];
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < 16; ++i) s += acc[i];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int GAP>
void run(float* buf, long long* cyc, float* sink, unsigned bytes) {
  const int iters = 64;
  k<MODE, GAP><<<256, 256>>>(buf, cyc, sink, 4, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<MODE, GAP><<<256, 256>>>(buf, cyc, sink, iters, bytes);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const char* names[4] = {"store strided   ", "store contiguous", "load  strided   ", "load  contiguous"};
  printf("%s gap=%2d MFMA : %7.1f cycles per instruction (of which MFMA %4d)   %.2f TB/s chip\n", names[MODE], GAP, (double)c / (iters * 32), GAP * 32,
         256.0 * 4 * iters * 32 * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  const unsigned bytes = 256u * 4 * 32768;   // 32 MB: every wave rewrites / rereads its own 32 KB (L2-resident after the first pass)
  float *buf, *sink; long long* cyc;
  (void)hipMalloc(&buf, bytes); (void)hipMemset(buf, 0, bytes); (void)hipMalloc(&sink, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  run<0, 0>(buf, cyc, sink, bytes); run<1, 0>(buf, cyc, sink, bytes); run<2, 0>(buf, cyc, sink, bytes); run<3, 0>(buf, cyc, sink, bytes);
  run<0, 2>(buf, cyc, sink, bytes); run<1, 2>(buf, cyc, sink, bytes); run<2, 2>(buf, cyc, sink, bytes); run<3, 2>(buf, cyc, sink, bytes);
  run<0, 6>(buf, cyc, sink, bytes); run<1, 6>(buf, cyc, sink, bytes); run<2, 6>(buf, cyc, sink, bytes); run<3, 6>(buf, cyc, sink, bytes);
  run<0, 12>(buf, cyc, sink, bytes); run<1, 12>(buf, cyc, sink, bytes); run<2, 12>(buf, cyc, sink, bytes); run<3, 12>(buf, cyc, sink, bytes);
  return 0;
}
