// Micro-benchmark for DESIGN.md 9 (round 5), "two waves per SIMD for the dominant kernel, variant (ii)": the instruction mix of ONE output row tile of the f16mx
// neural-point kernel (K = 256: 16 v_mfma_f32_32x32x16_f16 + 8 v_mfma_scale_f32_32x32x64_f8f6f4, 32 ds_read_b128 of A fragments, 8 LDS-DMA pieces per wave of four,
// the previous tile's epilogue: 8 pairs x 9 VALU) issued
//   MODE 1: by ONE wave per SIMD, as today (= tools/ubench/mx8_rowtile.hip with everything on: 1 557 ticks per row tile against 1 025 for the MFMAs alone), or
//   MODE 2: by TWO waves per SIMD that split K — each wave multiplies its 128 k of the tile (8 f16 + 4 fp8 matrix instructions, 16 A-fragment reads, 4 DMA pieces),
//           the wave that does not own the tile's output channels hands its partial accumulator over through LDS (4 ds_write_b128; the owner: 4 ds_read_b128 + 16 adds
//           + the epilogue), roles alternating tile by tile; one workgroup barrier per row tile as today.
// What it answers: can a second wave's issue slots hide the side work that one wave issues at ~5 cycles per instruction?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NW>   // waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD, K split)
__global__ __launch_bounds__(64 * NW, NW / 4) void k(float* out, long long* cyc, const char* wsrc, int iters) {
  __shared__ float lds[4 * 8192 + 8 * 1024];   // 4 x 32 KB weight ring + 8 x 4 KB hand-over tiles
  for (int i = threadIdx.x; i < 4 * 8192 + 8 * 1024; i += 64 * NW) lds[i] = (float)i;
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
  const unsigned xch = lds_base + 4 * 32768u + (unsigned)(wave & 3) * 4096u + lane * 16;   // the SIMD pair's hand-over tile
  const char* gp = wsrc + wave * 4096 + lane * 16;
  const int sA = 127, sB = 116;
  const float one = 1.0f, sc11 = 0.00048828125f;
  constexpr int NSLAB = NW == 4 ? 4 : 2;   // K = 64 slabs per wave and row tile
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned slot = (it & 3) * 32768u + (NW == 8 ? (unsigned)(wave >> 2) * 16384u : 0u);   // the wave's K half of the chunk
    const bool owner = NW == 4 || (((it + (wave >> 2)) & 1) == 0);
    if (NW == 8 && owner) {   // the partner's partial tile of the PREVIOUS row tile: 4 reads, 16 adds (then the epilogue below)
      f32x4 p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(p[i]) : "v"(xch + 1024u * i));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] += p[i >> 2][i & 3];
    }
#pragma unroll
    for (int q = 0; q < NSLAB; ++q) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(afr[s]) : "v"(lds_addr + slot + 8192u * q + 1024u * s));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(afr[s]), "v"(b));
        asm volatile("ds_read_b128 %0, %1" : "=v"(afr[4 + s]) : "v"(lds_addr + slot + 8192u * q + 4096u + 1024u * s));
        if (owner) {   // epilogue of the previous tile: 8 pairs; a K-split wave runs all 8 inside its 2 slabs (two steps per k-step)
          constexpr int SPS = NW == 4 ? 1 : 2;
#pragma unroll
          for (int u = 0; u < SPS; ++u) {
            const int st = (4 * q + s) * SPS + u;   // 0..15: pair st >> 1, half st & 1
            const int p = st >> 1;
            if ((st & 1) == 0) {
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
      }
      {
        v8i a8 = {__builtin_bit_cast(int, afr[4][0]), __builtin_bit_cast(int, afr[4][1]), __builtin_bit_cast(int, afr[4][2]), __builtin_bit_cast(int, afr[4][3]),
                  __builtin_bit_cast(int, afr[5][0]), __builtin_bit_cast(int, afr[5][1]), __builtin_bit_cast(int, afr[5][2]), __builtin_bit_cast(int, afr[5][3])};
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a8), "v"(b8), "v"(sA), "v"(sB));
      }
      {
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_base + (((it + 2) & 3) * 32768u) + wave * (NW == 4 ? 8192u : 4096u) + (2 * q) * 1024u)));
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp + (2 * q) * 1024) : "memory");
      }
      {
        v8i a8 = {__builtin_bit_cast(int, afr[6][0]), __builtin_bit_cast(int, afr[6][1]), __builtin_bit_cast(int, afr[6][2]), __builtin_bit_cast(int, afr[6][3]),
                  __builtin_bit_cast(int, afr[7][0]), __builtin_bit_cast(int, afr[7][1]), __builtin_bit_cast(int, afr[7][2]), __builtin_bit_cast(int, afr[7][3])};
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a8), "v"(b8), "v"(sB), "v"(sA));
      }
      {
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_base + (((it + 2) & 3) * 32768u) + wave * (NW == 4 ? 8192u : 4096u) + (2 * q + 1) * 1024u)));
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp + (2 * q + 1) * 1024) : "memory");
      }
    }
    if (NW == 8 && !owner) {   // hand this tile's partial accumulator to the owner
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 v = {acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
        asm volatile("ds_write_b128 %0, %1" ::"v"(xch + 1024u * i), "v"(v) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NW == 4 ? 8 : 4) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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

template <int NW>
void run(float* out, long long* cyc, const char* w) {
  const int iters = 1000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NW><<<256, 64 * NW>>>(out, cyc, w, iters / 10);
  (void)hipEventRecord(e0);
  k<NW><<<256, 64 * NW>>>(out, cyc, w, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // a row tile = 32 rows x 32 outputs x K = 256 per SIMD in both modes (mode 2: two waves x K = 128)
  printf("%d wave(s) per SIMD: %7.1f counter ticks per row tile  %7.3f ms  = %.2f PFLOP/s algorithmic (32x32x256 MACs per SIMD and tile)\n", NW / 4, (double)c / iters, ms,
         256.0 * 4 * iters * 32 * 32 * 256 * 2 / (ms * 1e-3) / 1e15);
}

int main() {
  float* out; long long* cyc; char* w;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20);
  run<4>(out, cyc, w); run<8>(out, cyc, w); run<4>(out, cyc, w); run<8>(out, cyc, w);
  return 0;
}
