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
                                                          const float* __restrict__ gamma /*(L,Cc): position-major like the slab*/, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ out, float* __restrict__ pooled) {
  __shared__ float red[8];
  const int r = blockIdx.x;
  const int n = L * Cc;
  const float* x = in + (size_t)r * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // one pass over the slab: shifted sums (shift = first element) give mean and variance without the cancellation of
  // E[x^2] - mean^2 and without a second read
  const float x0 = x[0];
  float s = 0.f, q = 0.f;
  if ((n & 3) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int i = tid; i < n / 4; i += 256) {
      const float4 v = x4[i];
      const float a = v.x - x0, b = v.y - x0, c = v.z - x0, d = v.w - x0;
      s += (a + b) + (c + d);
      q += (a * a + b * b) + (c * c + d * d);
    }
  } else {
    for (int i = tid; i < n; i += 256) { const float d = x[i] - x0; s += d; q += d * d; }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if (lane == 0) { red[wave] = s; red[4 + wave] = q; }
  __syncthreads();
  const float ms = (red[0] + red[1] + red[2] + red[3]) / (float)n;
  const float mean = x0 + ms;
  const float var = fmaxf((red[4] + red[5] + red[6] + red[7]) / (float)n - ms * ms, 0.f);
  const float rstd = 1.f / sqrtf(var + eps);
  if (pooled) {
    // thread handles (position pair, channel)
    const int half = (L / 2) * Cc;
    for (int i = tid; i < half; i += 256) {
      const int p = i / Cc, c = i - p * Cc;
      const int l0 = 2 * p, l1 = 2 * p + 1;
      float a = nl_elu((x[l0 * Cc + c] - mean) * rstd * gamma[l0 * Cc + c] + beta[l0 * Cc + c]);
      float b = nl_elu((x[l1 * Cc + c] - mean) * rstd * gamma[l1 * Cc + c] + beta[l1 * Cc + c]);
      if (out) { out[(size_t)r * n + l0 * Cc + c] = a; out[(size_t)r * n + l1 * Cc + c] = b; }
      pooled[(size_t)r * half + i] = fmaxf(a, b);
    }
  } else {
    for (int i = tid; i < n; i += 256) out[(size_t)r * n + i] = nl_elu((x[i] - mean) * rstd * gamma[i] + beta[i]);
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
