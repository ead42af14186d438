// Ray U-Net glue (conditional_nerf/ray_unet.py:5-69).
// The k=3 convolutions / stride-2 transposed convolutions along the ray run on the MFMA segment-GEMM
// (3 shifted row-taps; even/odd output phases for the transposed ones).  This file holds what sits
// between them: LayerNorm over the whole (C, L) slab of one ray with per-(c, l) affine
// (nn.LayerNorm([C, L]), eps 1e-5), ELU, and the fused MaxPool1d(2).
// Activations are sample-major: rows = (ray, position), columns = channels.
#include "common.h"

namespace {

// one block per ray: in (L, Cc) -> out (L, Cc) = ELU(LN(in)) ; pooled (L/2, Cc) = max over position pairs (optional)
__global__ __launch_bounds__(256) void ln_slab_elu_kernel(const float* __restrict__ in, int L, int Cc,
                                                          const float* __restrict__ gamma /*(Cc,L)*/, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ out, float* __restrict__ pooled) {
  __shared__ float red[8];
  const int r = blockIdx.x;
  const int n = L * Cc;
  const float* x = in + (size_t)r * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int i = tid; i < n; i += 256) s += x[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)n;
  __syncthreads();
  float v = 0.f;
  for (int i = tid; i < n; i += 256) { float d = x[i] - mean; v += d * d; }
  v = wave_sum(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const float rstd = 1.f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)n + eps);
  if (pooled) {
    // thread handles (position pair, channel)
    const int half = (L / 2) * Cc;
    for (int i = tid; i < half; i += 256) {
      const int p = i / Cc, c = i - p * Cc;
      const int l0 = 2 * p, l1 = 2 * p + 1;
      float a = nl_elu((x[l0 * Cc + c] - mean) * rstd * gamma[c * L + l0] + beta[c * L + l0]);
      float b = nl_elu((x[l1 * Cc + c] - mean) * rstd * gamma[c * L + l1] + beta[c * L + l1]);
      if (out) { out[(size_t)r * n + l0 * Cc + c] = a; out[(size_t)r * n + l1 * Cc + c] = b; }
      pooled[(size_t)r * half + i] = fmaxf(a, b);
    }
  } else {
    for (int i = tid; i < n; i += 256) {
      const int l = i / Cc, c = i - l * Cc;
      out[(size_t)r * n + i] = nl_elu((x[i] - mean) * rstd * gamma[c * L + l] + beta[c * L + l]);
    }
  }
}

}  // namespace

int nl_launch_ln_slab_elu(const float* in, int64_t R, int L, int Cc, const float* gamma, const float* beta, float eps,
                          float* out, float* pooled, hipStream_t st) {
  if (R <= 0) return NL_OK;
  hipLaunchKernelGGL(ln_slab_elu_kernel, dim3((unsigned)R), dim3(256), 0, st, in, L, Cc, gamma, beta, eps, out, pooled);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
