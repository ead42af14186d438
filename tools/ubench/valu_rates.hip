// Micro-benchmark: issue cost (cycles per wave64 instruction, one wave per SIMD) of the vector instructions used in the
// bf16 hi/lo split epilogues.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = 0.001f * threadIdx.x + i;
  unsigned u[16];
  for (int i = 0; i < 16; ++i) u[i] = threadIdx.x * 977u + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        if (OP == 0) { x[i] = fmaf(x[i], 1.0001f, 0.5f); x[i + 1] = fmaf(x[i + 1], 1.0001f, 0.5f); }
        if (OP == 1) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[i + 1])); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i + 1]) : "v"(x[i + 1]), "v"(x[i])); }
        if (OP == 2) { asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[i + 1]), "s"(0x07060302)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i + 1]) : "v"(u[i + 1]), "v"(u[i]), "s"(0x07060302)); }
        if (OP == 3) { asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i]) : "v"(u[i])); asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i + 1]) : "v"(u[i + 1])); }
        if (OP == 4) { f2 a = {x[i], x[i + 1]}; asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(a) : "v"(a)); x[i] = a[0]; x[i + 1] = a[1]; asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(a) : "v"(a)); x[i] = a[0]; }
        if (OP == 5) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(x[i + 1])); asm volatile("v_max_f32 %0, %1, %2" : "=v"(x[i + 1]) : "v"(x[i + 1]), "v"(x[i])); }
        if (OP == 6) { asm volatile("v_exp_f32 %0, %1" : "=v"(x[i]) : "v"(x[i])); asm volatile("v_exp_f32 %0, %1" : "=v"(x[i + 1]) : "v"(x[i + 1])); }
        if (OP == 8) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[i + 1])); asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i + 1]) : "v"(x[i + 1]), "v"(x[i])); }
        if (OP == 9) { asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(x[i]) : "v"(u[i]), "v"(x[i])); asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(x[i + 1]) : "v"(u[i]), "v"(x[i + 1])); }
        if (OP == 10) { asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, 1.0" : "+v"(u[i]) : "v"(u[i + 1])); asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, 1.0 op_sel:[0,0,1]" : "+v"(u[i + 1]) : "v"(u[i])); }
        if (OP == 11) { asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(u[i]) : "v"(x[i]), "v"(x[i + 1]), "v"(x[0])); asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(u[i + 1]) : "v"(x[i + 1]), "v"(x[i]), "v"(x[0])); }
        if (OP == 12) { asm volatile("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(x[i]) : "v"(x[i]), "v"(x[i + 1]), "v"(x[(i + 2) & 15])); asm volatile("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(x[i + 1]) : "v"(x[i + 1]), "v"(x[i]), "v"(x[(i + 3) & 15])); }
        if (OP == 13) { f2 a = {x[i], x[i + 1]}; asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(a) : "v"(a)); x[i] = a[0]; x[i + 1] = a[1]; asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(a) : "v"(a)); x[i] = a[0]; }
        if (OP == 7) { double d = x[i]; asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(d) : "v"(d)); asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(d) : "v"(d)); x[i] = (float)d; }
      }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(float* out, const char* name) {
  const int iters = 20000;
  k<OP><<<256, 256>>>(out, 100);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<OP><<<256, 256>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s %.2f ns per instruction per wave\n", name, ms * 1e6 / ((double)iters * 8 * 16));
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  run<0>(out, "v_fma_f32"); run<1>(out, "v_cvt_pk_bf16_f32"); run<2>(out, "v_perm_b32"); run<3>(out, "v_and_b32");
  run<8>(out, "v_cvt_pk_f16_f32"); run<9>(out, "v_fma_mix_f32"); run<10>(out, "v_cvt_scalef32_pk_fp8_f16"); run<11>(out, "v_cvt_scalef32_pk_fp8_f32"); run<12>(out, "v_max3_f32 |.|"); run<13>(out, "v_pk_fma_f32");
  run<4>(out, "v_pk_mul_f32"); run<5>(out, "v_max_f32"); run<6>(out, "v_exp_f32"); run<7>(out, "v_fma_f64 (+cvt)");
  return 0;
}
