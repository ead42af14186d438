// Ray sampling (rows a2-a3) and the post-U-Net heads + compositing (rows a14-a18).
#include "mvdec.h"

namespace {

// z = near*(1-t) + far*t with torch.linspace's symmetric formula (model.py:451-458); xyz = o + d*z (model.py:498)
__global__ void sample_points_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int R, int S,
                                     float near_, float far_, const float* __restrict__ z_in, float* __restrict__ z_out,
                                     float* __restrict__ xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * S) return;
  const int r = i / S, s = i - r * S;
  float z;
  if (z_in) z = z_in[i];
  else {
    const float step = S > 1 ? (1.f - 0.f) / (float)(S - 1) : 0.f;
    const float t = (s < S / 2) ? __fmul_rn(step, (float)s) : __fsub_rn(1.f, __fmul_rn(step, (float)(S - 1 - s)));
    z = __fadd_rn(__fmul_rn(near_, __fsub_rn(1.f, t)), __fmul_rn(far_, t));
  }
  if (z_out) z_out[i] = z;
#pragma unroll
  for (int d = 0; d < 3; ++d) xyz[3 * (size_t)i + d] = __fadd_rn(rays_o[3 * r + d], __fmul_rn(rays_d[3 * r + d], z));
}

// sigma = softplus(w . geo + b)  (model.py:525, sigma_mlp :83) — one wave per sample
__global__ __launch_bounds__(256) void sigma_kernel(const float* __restrict__ geo, int N, int W, const float* __restrict__ w,
                                                    const float* __restrict__ b, float* __restrict__ sigma) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s = 0.f;
  for (int c = lane; c < W; c += 64) s = fmaf(geo[(size_t)n * W + c], w[c], s);
  s = wave_sum(s);
  if (lane == 0) sigma[n] = nl_softplus(s + b[0]);
}

// colour blending tail (model.py:535-538): layer 1 = LeakyReLU(hA[n] + h1[n,v]) (split by linearity) -> 16 -> 1,
// masked_fill(vis == 0, -1e9), softmax over views, rgb = sum_v w_v * rgb_in.   One lane per sample.
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ hA, const float* __restrict__ h1, const float* __restrict__ rgbv /*(N*V,4) = [r,g,b,vis]*/,
                                                    int N, int V,
                                                    const float* __restrict__ w2 /*[16][32]*/, const float* __restrict__ b2,
                                                    const float* __restrict__ w4 /*[16]*/, const float* __restrict__ b4,
                                                    float* __restrict__ rgb_s, const int* __restrict__ n_alive, int S) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (n_alive) {   // early termination: the sample lies behind the point where the ray's transmittance fell below eps
    const int r = n / S;
    if (n - r * S >= n_alive[r]) { rgb_s[3 * (size_t)n] = 0.f; rgb_s[3 * (size_t)n + 1] = 0.f; rgb_s[3 * (size_t)n + 2] = 0.f; return; }
  }
  float lg[NL_MAX_VIEWS];
  float mx = -3.4e38f;
  float xa[32];   // per-sample part of layer 1 (feature_agg columns of rgb_blending_mlp.0)
#pragma unroll
  for (int i4 = 0; i4 < 8; ++i4) {
    float4 t = *(const float4*)(hA + (size_t)n * 32 + 4 * i4);
    xa[4 * i4] = t.x; xa[4 * i4 + 1] = t.y; xa[4 * i4 + 2] = t.z; xa[4 * i4 + 3] = t.w;
  }
  for (int v = 0; v < V; ++v) {
    const float* h = h1 + ((size_t)n * V + v) * 32;
    float x[32];
#pragma unroll
    for (int i4 = 0; i4 < 8; ++i4) {
      float4 t = *(const float4*)(h + 4 * i4);
      x[4 * i4] = nl_lrelu(xa[4 * i4] + t.x); x[4 * i4 + 1] = nl_lrelu(xa[4 * i4 + 1] + t.y);
      x[4 * i4 + 2] = nl_lrelu(xa[4 * i4 + 2] + t.z); x[4 * i4 + 3] = nl_lrelu(xa[4 * i4 + 3] + t.w);
    }
    float o = b4[0];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float a = b2[j];
#pragma unroll
      for (int i = 0; i < 32; ++i) a = fmaf(w2[j * 32 + i], x[i], a);
      o = fmaf(w4[j], nl_lrelu(a), o);
    }
    const float vis = rgbv[((size_t)n * V + v) * 4 + 3];
    o = (vis == 0.f) ? -1e9f : o;
    lg[v] = o;
    mx = fmaxf(mx, o);
  }
  float den = 0.f;
  for (int v = 0; v < V; ++v) { lg[v] = expf(lg[v] - mx); den += lg[v]; }
  float r = 0.f, g = 0.f, b = 0.f;
  for (int v = 0; v < V; ++v) {
    const float wv = lg[v] / den;
    const float* c = rgbv + ((size_t)n * V + v) * 4;
    r += c[0] * wv; g += c[1] * wv; b += c[2] * wv;
  }
  rgb_s[3 * (size_t)n] = r; rgb_s[3 * (size_t)n + 1] = g; rgb_s[3 * (size_t)n + 2] = b;
}

// The same tail with the per-(sample, view) part of layer 1 RECOMPUTED instead of read back (round 4: mv_front_kernel no longer writes the N x V x 32 rows,
// 0.7 GB out and in per config-2 batch): IBRNet projection of the sample into view v (ibrnet.py:169-192), the bilinear (zeros, align_corners = True) tap of
// the per-frame projected feature map pfeat = W[:, feat] . featmap (32 channels: a linear layer commutes with the tap), the view-angle features
// (ibrnet.py:144-167) and the [rgb | vis | angle] columns + bias (blw).  One lane per sample; views the sample is invisible in (vis == 0: logit masked to
// -1e9, model.py:536) are skipped by the lane.  rgbv (N*V, 4) = tapped colours + visibility from mv_front_kernel.
__global__ __launch_bounds__(256) void blend_taps_kernel(const NlViews vw, const float* __restrict__ viewsdev, const float* __restrict__ pfeat /*(V,h,w,32)*/,
                                                         const float* __restrict__ blw /*[32][8], bias[32]*/, const float* __restrict__ xyz,
                                                         const float* __restrict__ hA, const float* __restrict__ rgbv, int N,
                                                         const float* __restrict__ w2 /*[16][32]*/, const float* __restrict__ b2,
                                                         const float* __restrict__ w4 /*[16]*/, const float* __restrict__ b4,
                                                         float* __restrict__ rgb_s, const int* __restrict__ n_alive, int S) {
  using namespace nlmv;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (n_alive) {
    const int r = n / S;
    if (n - r * S >= n_alive[r]) { rgb_s[3 * (size_t)n] = 0.f; rgb_s[3 * (size_t)n + 1] = 0.f; rgb_s[3 * (size_t)n + 2] = 0.f; return; }
  }
  const int V = vw.V;
  const float X = xyz[3 * (size_t)n], Y = xyz[3 * (size_t)n + 1], Z = xyz[3 * (size_t)n + 2];
  float qc0 = vw.qcam[0], qc1 = vw.qcam[1], qc2 = vw.qcam[2];
  if (vw.qrows) { const float* qr = vw.qrows + 3 * (size_t)(n / vw.qS); qc0 = qr[0]; qc1 = qr[1]; qc2 = qr[2]; }
  float tq[3] = {qc0 - X, qc1 - Y, qc2 - Z};
  const float rq = 1.f / (sqrtf(tq[0] * tq[0] + tq[1] * tq[1] + tq[2] * tq[2]) + 1e-6f);
  tq[0] *= rq; tq[1] *= rq; tq[2] *= rq;
  float xa[32];   // per-sample part of layer 1 (feature_agg columns of rgb_blending_mlp.0)
#pragma unroll
  for (int i4 = 0; i4 < 8; ++i4) {
    const float4 t = *(const float4*)(hA + (size_t)n * 32 + 4 * i4);
    xa[4 * i4] = t.x; xa[4 * i4 + 1] = t.y; xa[4 * i4 + 2] = t.z; xa[4 * i4 + 3] = t.w;
  }
  const size_t fmap = (size_t)vw.h * vw.w;
  float lg[NL_MAX_VIEWS];
  float mx = -3.4e38f;
  for (int v = 0; v < V; ++v) {
    const float4 cv = *(const float4*)(rgbv + ((size_t)n * V + v) * 4);
    float o = -1e9f;
    if (cv.w != 0.f) {
      const float4 p0 = *(const float4*)(viewsdev + 12 * v), p1 = *(const float4*)(viewsdev + 12 * v + 4), p2 = *(const float4*)(viewsdev + 12 * v + 8);
      const float cx = fmaf(p0.z, Z, fmaf(p0.y, Y, p0.x * X)) + p0.w;
      const float cy = fmaf(p1.z, Z, fmaf(p1.y, Y, p1.x * X)) + p1.w;
      const float cz = fmaf(p2.z, Z, fmaf(p2.y, Y, p2.x * X)) + p2.w;
      const float zc = fmaxf(cz, 1e-8f);
      float px = cx / zc, py = cy / zc;
      px = fminf(fmaxf(px, -1e6f), 1e6f);
      py = fminf(fmaxf(py, -1e6f), 1e6f);
      const float xn = 2.f * px / (float)(vw.Wimg - 1) - 1.f;
      const float yn = 2.f * py / (float)(vw.H - 1) - 1.f;
      const Taps tf = make_taps<true, false>(xn, yn, vw.w, vw.h);
      int of[4];
      unpack_taps(pack_taps(tf, vw.w, vw.h), vw.w, of);
      const float w0 = (tf.mn && tf.mw) ? tf.nw : 0.f, w1 = (tf.mn && tf.me) ? tf.ne : 0.f, w2t = (tf.ms && tf.mw) ? tf.sw : 0.f, w3 = (tf.ms && tf.me) ? tf.se : 0.f;
      float tt[3] = {viewsdev[192 + 3 * v] - X, viewsdev[192 + 3 * v + 1] - Y, viewsdev[192 + 3 * v + 2] - Z};
      const float rt = 1.f / (sqrtf(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]) + 1e-6f);
      tt[0] *= rt; tt[1] *= rt; tt[2] *= rt;
      const float df[3] = {tq[0] - tt[0], tq[1] - tt[1], tq[2] - tt[2]};
      const float rd = 1.f / fmaxf(sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]), 1e-6f);
      const float in8[8] = {cv.x, cv.y, cv.z, cv.w, df[0] * rd, df[1] * rd, df[2] * rd, tq[0] * tt[0] + tq[1] * tt[1] + tq[2] * tt[2]};
      const float* pb = pfeat + (size_t)v * fmap * 32;
      float x[32];
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 a = *(const float4*)(pb + (size_t)of[0] * 32 + 4 * c4), b = *(const float4*)(pb + (size_t)of[1] * 32 + 4 * c4);
        const float4 c = *(const float4*)(pb + (size_t)of[2] * 32 + 4 * c4), d = *(const float4*)(pb + (size_t)of[3] * 32 + 4 * c4);
        x[4 * c4] = fmaf(d.x, w3, fmaf(c.x, w2t, fmaf(b.x, w1, a.x * w0)));
        x[4 * c4 + 1] = fmaf(d.y, w3, fmaf(c.y, w2t, fmaf(b.y, w1, a.y * w0)));
        x[4 * c4 + 2] = fmaf(d.z, w3, fmaf(c.z, w2t, fmaf(b.z, w1, a.z * w0)));
        x[4 * c4 + 3] = fmaf(d.w, w3, fmaf(c.w, w2t, fmaf(b.w, w1, a.w * w0)));
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float a = x[i] + blw[256 + i];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) a = fmaf(blw[i * 8 + jj], in8[jj], a);
        x[i] = nl_lrelu(xa[i] + a);
      }
      o = b4[0];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a = b2[j];
#pragma unroll
        for (int i = 0; i < 32; ++i) a = fmaf(w2[j * 32 + i], x[i], a);
        o = fmaf(w4[j], nl_lrelu(a), o);
      }
    }
    lg[v] = o;
    mx = fmaxf(mx, o);
  }
  float den = 0.f;
  for (int v = 0; v < V; ++v) { lg[v] = expf(lg[v] - mx); den += lg[v]; }
  float r = 0.f, g = 0.f, b = 0.f;
  for (int v = 0; v < V; ++v) {
    const float wv = lg[v] / den;
    const float* c = rgbv + ((size_t)n * V + v) * 4;
    r += c[0] * wv; g += c[1] * wv; b += c[2] * wv;
  }
  rgb_s[3 * (size_t)n] = r; rgb_s[3 * (size_t)n + 1] = g; rgb_s[3 * (size_t)n + 2] = b;
}

// Front-to-back compositing (model.py:541-560,597) + valid-ray mask (:572-575).  One wave per ray.
// Each lane owns CH = ceil(S/64) consecutive samples; transmittance = exclusive product scan
// (lane-local serial product, then a wave-level multiplicative scan of the lane totals).
template <int CH>
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ z_vals, const float* __restrict__ sigma,
                                                        const float* __restrict__ rgb_s, const float* __restrict__ ft /*(R*S,C)*/,
                                                        const int* __restrict__ valid_s, int R, int S, int C, int white_bkgd,
                                                        float* __restrict__ o_rgb, float* __restrict__ o_depth,
                                                        float* __restrict__ o_w, unsigned char* __restrict__ o_mask,
                                                        float* __restrict__ o_unc, float* __restrict__ o_feat, float* __restrict__ o_wsum,
                                                        const int* __restrict__ n_alive) {
  __shared__ float wsh[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wv;
  if (r >= R) return;
  const int na = n_alive ? n_alive[r] : S;   // early termination: colours / features of samples >= na are not evaluated (their weight is < eps)
  const float* z = z_vals + (size_t)r * S;
  float zs[CH], al[CH];
  float prod = 1.f;
  int nvalid = 0;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) {
      zs[j] = z[s];
      const float delta = (s + 1 < S) ? z[s + 1] - zs[j] : 1e2f;
      al[j] = 1.f - expf(-delta * sigma[(size_t)r * S + s]);
      prod *= (1.f - al[j]);
      nvalid += valid_s ? valid_s[(size_t)r * S + s] : 0;
    } else { zs[j] = 0.f; al[j] = 0.f; }
  }
  // inclusive multiplicative scan of lane products, shifted to exclusive
  float inc = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(inc, o, 64);
    if (lane >= o) inc *= t;
  }
  float T = __shfl_up(inc, 1, 64);
  if (lane == 0) T = 1.f;
  float w[CH];
  float wsum = 0.f, dsum = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) {
      w[j] = al[j] * T;
      T *= (1.f - al[j]);
      wsum += w[j];
      dsum += w[j] * zs[j];
      if (s < na) {
        const float* c = rgb_s + 3 * ((size_t)r * S + s);
        cr += w[j] * c[0]; cg += w[j] * c[1]; cb += w[j] * c[2];
      }
      if (o_w) o_w[(size_t)r * S + s] = w[j];
      wsh[wv][s] = w[j];
    } else w[j] = 0.f;
  }
  wsum = wave_sum(wsum); dsum = wave_sum(dsum);
  cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb);
  int nv = nvalid;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o, 64);
  float us = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) { float d = zs[j] - dsum; us += w[j] * (d * d); }
  }
  us = wave_sum(us);
  if (lane == 0) {
    if (white_bkgd) { cr += 1.f - wsum; cg += 1.f - wsum; cb += 1.f - wsum; }
    if (o_rgb) { o_rgb[3 * (size_t)r] = cr; o_rgb[3 * (size_t)r + 1] = cg; o_rgb[3 * (size_t)r + 2] = cb; }
    if (o_depth) o_depth[r] = dsum;
    if (o_unc) o_unc[r] = us;
    if (o_mask) o_mask[r] = nv > 8 ? 1 : 0;
    if (o_wsum) o_wsum[r] = wsum;
  }
  if (o_feat && ft) {
    __builtin_amdgcn_wave_barrier();
    // lanes across channels, serial over samples; weights broadcast from LDS (written by this wave only).  Four channels per lane and four samples in flight where
    // the rows allow it: at 512 rays (a rank's shard of config 2 on 8 GPUs) there are two waves per CU and the loop is pure load latency (68 -> 25 us)
    if ((C & 3) == 0 && (((size_t)ft | (size_t)o_feat) & 15) == 0) {
      for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + 4 * lane;
        if (c < C) {
          const float* fp = ft + (size_t)r * S * C + c;
          float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
          int s = 0;
          for (; s + 4 <= na; s += 4) {
            const float4 f0 = *(const float4*)(fp + (size_t)s * C), f1 = *(const float4*)(fp + (size_t)(s + 1) * C), f2 = *(const float4*)(fp + (size_t)(s + 2) * C),
                         f3 = *(const float4*)(fp + (size_t)(s + 3) * C);
            const float w0 = wsh[wv][s], w1 = wsh[wv][s + 1], w2 = wsh[wv][s + 2], w3 = wsh[wv][s + 3];
            a0.x = fmaf(w0, f0.x, a0.x); a0.y = fmaf(w0, f0.y, a0.y); a0.z = fmaf(w0, f0.z, a0.z); a0.w = fmaf(w0, f0.w, a0.w);
            a1.x = fmaf(w1, f1.x, a1.x); a1.y = fmaf(w1, f1.y, a1.y); a1.z = fmaf(w1, f1.z, a1.z); a1.w = fmaf(w1, f1.w, a1.w);
            a2.x = fmaf(w2, f2.x, a2.x); a2.y = fmaf(w2, f2.y, a2.y); a2.z = fmaf(w2, f2.z, a2.z); a2.w = fmaf(w2, f2.w, a2.w);
            a3.x = fmaf(w3, f3.x, a3.x); a3.y = fmaf(w3, f3.y, a3.y); a3.z = fmaf(w3, f3.z, a3.z); a3.w = fmaf(w3, f3.w, a3.w);
          }
          for (; s < na; ++s) {
            const float4 f0 = *(const float4*)(fp + (size_t)s * C);
            const float w0 = wsh[wv][s];
            a0.x = fmaf(w0, f0.x, a0.x); a0.y = fmaf(w0, f0.y, a0.y); a0.z = fmaf(w0, f0.z, a0.z); a0.w = fmaf(w0, f0.w, a0.w);
          }
          *(float4*)(o_feat + (size_t)r * C + c) = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                                                               (a0.w + a1.w) + (a2.w + a3.w));
        }
      }
    } else
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int c = c0 + lane;
      float acc = 0.f;
      if (c < C)
        for (int s = 0; s < na; ++s) acc = fmaf(wsh[wv][s], ft[((size_t)r * S + s) * C + c], acc);
      if (c < C) o_feat[(size_t)r * C + c] = acc;
    }
  }
}

// Early termination (BASELINE config 5 / SURVEY §8f-4).  The density needs the whole ray (the U-Net runs along it), so only what
// comes AFTER it can be skipped: the colour blend tail, feat_mlp.0's rows and the per-sample reads of the compositing.  One wave
// per ray: n_alive[r] = number of leading samples whose transmittance T_s (exclusive cumprod of 1 - alpha, model.py:549-552) is
// >= eps.  T is non-increasing, so the dropped samples are a suffix and their total weight is < eps.
template <int CH>
__global__ __launch_bounds__(256) void term_kernel(const float* __restrict__ z_vals, const float* __restrict__ sigma, int R, int S, float eps,
                                                   int* __restrict__ n_alive) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wv;
  if (r >= R) return;
  const float* z = z_vals + (size_t)r * S;
  float al[CH];
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) {
      const float delta = (s + 1 < S) ? z[s + 1] - z[s] : 1e2f;
      al[j] = 1.f - expf(-delta * sigma[(size_t)r * S + s]);
      prod *= (1.f - al[j]);
    } else al[j] = 0.f;
  }
  float inc = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(inc, o, 64);
    if (lane >= o) inc *= t;
  }
  float T = __shfl_up(inc, 1, 64);
  if (lane == 0) T = 1.f;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int s = lane * CH + j;
    if (s < S) { cnt += T >= eps ? 1 : 0; T *= (1.f - al[j]); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) n_alive[r] = cnt;
}

// compact list of the 32-row tiles of (R*S) that hold at least one live sample (order arbitrary: every tile is computed independently)
__global__ void tile_list_kernel(const int* __restrict__ n_alive, int R, int S, int ntiles, int* __restrict__ tile_list, int* __restrict__ count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const int r0 = (32 * t) / S, s0 = 32 * t - r0 * S;
  const bool alive = s0 < n_alive[r0] || s0 + 32 > S;   // a tile that runs into the next ray holds that ray's first samples (always live)
  if (alive) tile_list[atomicAdd(count, 1)] = t;
}

}  // namespace

int nl_launch_termination(const float* z_vals, const float* sigma, int64_t R, int S, float eps, int* n_alive, int* tile_list, int* tile_count,
                          hipStream_t st) {
  if (R <= 0) return NL_OK;
  if (S > 256) return NL_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nl_cdiv(R, 4));
#define NL_TERM(CH) hipLaunchKernelGGL(term_kernel<CH>, grid, dim3(256), 0, st, z_vals, sigma, (int)R, S, eps, n_alive)
  if (S <= 64) NL_TERM(1); else if (S <= 128) NL_TERM(2); else if (S <= 192) NL_TERM(3); else NL_TERM(4);
#undef NL_TERM
  NL_CHECK_HIP(hipMemsetAsync(tile_count, 0, sizeof(int), st));
  const int ntiles = (int)nl_cdiv(R * S, 32);
  hipLaunchKernelGGL(tile_list_kernel, dim3((unsigned)nl_cdiv(ntiles, 256)), dim3(256), 0, st, n_alive, (int)R, S, ntiles, tile_list, tile_count);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_sample_points(const float* rays_o, const float* rays_d, int64_t R, int S, float near_, float far_,
                            const float* z_in, float* z_out, float* xyz, hipStream_t st) {
  if (R <= 0) return NL_OK;
  hipLaunchKernelGGL(sample_points_kernel, dim3((unsigned)nl_cdiv(R * S, 256)), dim3(256), 0, st, rays_o, rays_d, (int)R, S,
                     near_, far_, z_in, z_out, xyz);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_sigma(const float* geo, int64_t N, int W, const float* w, const float* b, float* sigma, hipStream_t st) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(sigma_kernel, dim3((unsigned)nl_cdiv(N, 4)), dim3(256), 0, st, geo, (int)N, W, w, b, sigma);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_launch_blend(const float* hA, const float* h1, const float* rgbv, int64_t N, int V, const float* w2,
                    const float* b2, const float* w4, const float* b4, float* rgb_s, hipStream_t st, const int* n_alive, int S) {
  if (N <= 0) return NL_OK;
  hipLaunchKernelGGL(blend_kernel, dim3((unsigned)nl_cdiv(N, 256)), dim3(256), 0, st, hA, h1, rgbv, (int)N, V, w2, b2, w4, b4, rgb_s, n_alive, S);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

// Round 5: the same function with lanes = (sample row r, channel octet g) instead of lane = sample.  What bound the lane-per-sample kernel was not its ~900 vector
// instructions per (sample, view) but its taps: 32 sixteen-byte loads per lane and view, every lane on its own 128-byte texel row — the CU's address unit takes ~64
// cycles per such instruction (tools/ubench/vmem_issue.hip; `r5_pmc_sq*.csv`: 2.7e6 load instructions per launch, waves waiting 63 % of their cycles).  Here a wave walks
// 16 consecutive samples; the four lanes of a row own 8 of the projected map's 32 channels each: a tap is TWO loads per lane, and one instruction touches 16 texel rows
// instead of 64.  Per view: phase A (lane = (row, view of a group of four): projection, tap cell / weights, view-angle features -> a 48-byte LDS slot, wave-private),
// phase B (lane = (row, octet): 8 loads, the 8 x 8 [rgb | vis | angle] columns of layer 1 in registers, LeakyReLU), then rgb_blending_mlp.2 (32 -> 16) as three
// v_mfma_f32_16x16x32_bf16 (split-bf16 like every parity-mode product; its A fragments are built from the fp32 weights at kernel start) — the lane's octet IS its slice of
// the B operand — LeakyReLU, the 16 -> 1 layer as 4 FMAs + two cross-lane adds, and a streaming softmax over the views (no per-view array).
namespace {
typedef __bf16 bt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float bt_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bt_u32x4 __attribute__((ext_vector_type(4)));
constexpr int BT_SLOT = 9;    // dwords per (row, view): cell, w0 .. w3, the four view-angle features (an odd stride: dword accesses, no bank conflicts; 24 KB at 10 views)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void blend_taps_mfma_kernel(const NlViews vw, const float* __restrict__ viewsdev, const float* __restrict__ pfeat /*(V,h,w,32)*/,
                                                              const float* __restrict__ blw /*[32][8], bias[32]*/, const float* __restrict__ xyz,
                                                              const float* __restrict__ hA, const float* __restrict__ rgbv, int N,
                                                              const float* __restrict__ w2 /*[16][32]*/, const float* __restrict__ b2,
                                                              const float* __restrict__ w4 /*[16]*/, const float* __restrict__ b4,
                                                              float* __restrict__ rgb_s, const int* __restrict__ n_alive, int S) {
  using namespace nlmv;
  // Dynamic LDS: [288 floats: layer 1's small columns [32][8] + bias [32]] [4 waves][16 rows][V views][BT_SLOT].  The kernel is latency-bound (a dependent chain per view),
  // so it is built for waves in flight: the small columns are read from LDS (a broadcast per octet group) instead of held in 72 registers (156 -> 83), the slots are sized by
  // the frame's view count (24 KB at 10 views): 6 waves per SIMD where the first version had 3
  extern __shared__ __attribute__((aligned(16))) float bt_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 15, g = lane >> 4;
  const int V = vw.V;
  for (int i = threadIdx.x; i < 288; i += 256) bt_lds[i] = blw[i];
  __syncthreads();
  const float* wls = bt_lds + 64 * g;        // rows 8 g .. 8 g + 7 of the [32][8] block
  const float* bls = bt_lds + 256 + 8 * g;
  // ---- resident: layer 2 as A fragments (row n = r, k = 8 g + t), the output layer's slice
  bt_u32x4 ah, al;
  {
    unsigned hw[8], lw[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float w = w2[r * 32 + 8 * g + t];
      unsigned u = __float_as_uint(w);
      u += 0x7fffu + ((u >> 16) & 1u);
      hw[t] = u >> 16;
      unsigned v = __float_as_uint(w - __uint_as_float(hw[t] << 16));
      v += 0x7fffu + ((v >> 16) & 1u);
      lw[t] = v >> 16;
    }
    ah = bt_u32x4{hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16)};
    al = bt_u32x4{lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16), lw[4] | (lw[5] << 16), lw[6] | (lw[7] << 16)};
  }
  float b2q[4], w4q[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { b2q[t] = b2[4 * g + t]; w4q[t] = w4[4 * g + t]; }
  const float b4v = b4[0];
  const size_t fmap = (size_t)vw.h * vw.w;
  float* const myslots = bt_lds + 288 + (size_t)wave * 16 * V * BT_SLOT;   // [row][view][BT_SLOT]

  const int n = (blockIdx.x * 4 + wave) * 16 + r;
  const bool live = n < N;
  const int nn = live ? n : N - 1;
  bool alive = live;
  if (n_alive) { const int ry = nn / S; alive = live && (nn - ry * S) < n_alive[ry]; }
  const float X = xyz[3 * (size_t)nn], Y = xyz[3 * (size_t)nn + 1], Z = xyz[3 * (size_t)nn + 2];
  float qc0 = vw.qcam[0], qc1 = vw.qcam[1], qc2 = vw.qcam[2];
  if (vw.qrows) { const float* qr = vw.qrows + 3 * (size_t)(nn / vw.qS); qc0 = qr[0]; qc1 = qr[1]; qc2 = qr[2]; }
  float tq[3] = {qc0 - X, qc1 - Y, qc2 - Z};
  const float rq = 1.f / (sqrtf(tq[0] * tq[0] + tq[1] * tq[1] + tq[2] * tq[2]) + 1e-6f);
  tq[0] *= rq; tq[1] *= rq; tq[2] *= rq;
  float xa[8];   // per-sample part of layer 1 (feature_agg columns of rgb_blending_mlp.0), the lane's octet
  {
    const float4 t0 = *(const float4*)(hA + (size_t)nn * 32 + 8 * g), t1 = *(const float4*)(hA + (size_t)nn * 32 + 8 * g + 4);
    xa[0] = t0.x; xa[1] = t0.y; xa[2] = t0.z; xa[3] = t0.w; xa[4] = t1.x; xa[5] = t1.y; xa[6] = t1.z; xa[7] = t1.w;
  }
  // ---- phase A: lane = (row r, view 4 i + g)
  for (int v0 = 0; v0 < V; v0 += 4) {
    const int v = v0 + g;
    if (v < V) {
      const float4 p0 = *(const float4*)(viewsdev + 12 * v), p1 = *(const float4*)(viewsdev + 12 * v + 4), p2 = *(const float4*)(viewsdev + 12 * v + 8);
      const float cx = fmaf(p0.z, Z, fmaf(p0.y, Y, p0.x * X)) + p0.w;
      const float cy = fmaf(p1.z, Z, fmaf(p1.y, Y, p1.x * X)) + p1.w;
      const float cz = fmaf(p2.z, Z, fmaf(p2.y, Y, p2.x * X)) + p2.w;
      const float zc = fmaxf(cz, 1e-8f);
      float px = cx / zc, py = cy / zc;
      px = fminf(fmaxf(px, -1e6f), 1e6f);
      py = fminf(fmaxf(py, -1e6f), 1e6f);
      const float xn = 2.f * px / (float)(vw.Wimg - 1) - 1.f;
      const float yn = 2.f * py / (float)(vw.H - 1) - 1.f;
      const Taps tf = make_taps<true, false>(xn, yn, vw.w, vw.h);
      const float w0 = (tf.mn && tf.mw) ? tf.nw : 0.f, w1 = (tf.mn && tf.me) ? tf.ne : 0.f, w2t = (tf.ms && tf.mw) ? tf.sw : 0.f, w3 = (tf.ms && tf.me) ? tf.se : 0.f;
      float tt[3] = {viewsdev[192 + 3 * v] - X, viewsdev[192 + 3 * v + 1] - Y, viewsdev[192 + 3 * v + 2] - Z};
      const float rt = 1.f / (sqrtf(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]) + 1e-6f);
      tt[0] *= rt; tt[1] *= rt; tt[2] *= rt;
      const float df[3] = {tq[0] - tt[0], tq[1] - tt[1], tq[2] - tt[2]};
      const float rd = 1.f / fmaxf(sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]), 1e-6f);
      float* sl = myslots + ((size_t)r * V + v) * BT_SLOT;
      sl[0] = __uint_as_float(pack_taps(tf, vw.w, vw.h)); sl[1] = w0; sl[2] = w1; sl[3] = w2t; sl[4] = w3;
      sl[5] = df[0] * rd; sl[6] = df[1] * rd; sl[7] = df[2] * rd;
      sl[8] = tq[0] * tt[0] + tq[1] * tt[1] + tq[2] * tt[2];
    }
  }
  __builtin_amdgcn_wave_barrier();   // the slots are wave-private

  // ---- phase B: lane = (row r, channels 8 g .. 8 g + 7), streaming softmax over the views.  (Measured, same box: the first version of this kernel — small columns in registers, static slots: 3 waves per SIMD — 289 us,
  // the lane-per-sample kernel 284-291 us at config 2.  Both read the same 2.7 GB of texel rows per launch — 5.2 M (sample, view) pairs x 4 taps x 128 bytes — and that, not the
  // arithmetic, is the bound at full size: with the next view's taps prefetched into a second register set the kernel took 346 us.  Sharing a cell's rows between the
  // consecutive samples that fall into it, as mv_front_kernel does, is what would cut it.)
  float m = -3.4e38f, den = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
  for (int v = 0; v < V; ++v) {
    const float4 cv = *(const float4*)(rgbv + ((size_t)nn * V + v) * 4);
    const bool vis = alive && cv.w != 0.f;
    float o = -1e9f;
    if (__ballot(vis) != 0ull) {   // (wave-uniform: views none of the 16 samples is visible in are skipped)
      const float* sl = myslots + ((size_t)r * V + v) * BT_SLOT;
      const float4 s0 = make_float4(sl[0], sl[1], sl[2], sl[3]), s1 = make_float4(sl[4], sl[5], sl[6], sl[7]);
      const float a7 = sl[8];
      int of[4];
      unpack_taps(__float_as_uint(s0.x), vw.w, of);
      const float* pb = pfeat + (size_t)v * fmap * 32 + 8 * g;
      float x[8];
      {
        const float4 a0 = *(const float4*)(pb + (size_t)of[0] * 32), a1 = *(const float4*)(pb + (size_t)of[0] * 32 + 4);
        const float4 c0 = *(const float4*)(pb + (size_t)of[1] * 32), c1 = *(const float4*)(pb + (size_t)of[1] * 32 + 4);
        const float4 d0 = *(const float4*)(pb + (size_t)of[2] * 32), d1 = *(const float4*)(pb + (size_t)of[2] * 32 + 4);
        const float4 e0 = *(const float4*)(pb + (size_t)of[3] * 32), e1 = *(const float4*)(pb + (size_t)of[3] * 32 + 4);
        const float w0 = s0.y, w1 = s0.z, w2t = s0.w, w3 = s1.x;
        x[0] = fmaf(e0.x, w3, fmaf(d0.x, w2t, fmaf(c0.x, w1, a0.x * w0))); x[1] = fmaf(e0.y, w3, fmaf(d0.y, w2t, fmaf(c0.y, w1, a0.y * w0)));
        x[2] = fmaf(e0.z, w3, fmaf(d0.z, w2t, fmaf(c0.z, w1, a0.z * w0))); x[3] = fmaf(e0.w, w3, fmaf(d0.w, w2t, fmaf(c0.w, w1, a0.w * w0)));
        x[4] = fmaf(e1.x, w3, fmaf(d1.x, w2t, fmaf(c1.x, w1, a1.x * w0))); x[5] = fmaf(e1.y, w3, fmaf(d1.y, w2t, fmaf(c1.y, w1, a1.y * w0)));
        x[6] = fmaf(e1.z, w3, fmaf(d1.z, w2t, fmaf(c1.z, w1, a1.z * w0))); x[7] = fmaf(e1.w, w3, fmaf(d1.w, w2t, fmaf(c1.w, w1, a1.w * w0)));
      }
      const float in8[8] = {cv.x, cv.y, cv.z, cv.w, s1.y, s1.z, s1.w, a7};
      unsigned hh[8], hl[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = x[j] + bls[j];
        const float4 wa = *(const float4*)(wls + 8 * j), wb = *(const float4*)(wls + 8 * j + 4);
        a = fmaf(wa.x, in8[0], a); a = fmaf(wa.y, in8[1], a); a = fmaf(wa.z, in8[2], a); a = fmaf(wa.w, in8[3], a);
        a = fmaf(wb.x, in8[4], a); a = fmaf(wb.y, in8[5], a); a = fmaf(wb.z, in8[6], a); a = fmaf(wb.w, in8[7], a);
        const float hv = nl_lrelu(xa[j] + a);
        unsigned u = __float_as_uint(hv);
        u += 0x7fffu + ((u >> 16) & 1u);
        hh[j] = u >> 16;
        unsigned w = __float_as_uint(hv - __uint_as_float(hh[j] << 16));
        w += 0x7fffu + ((w >> 16) & 1u);
        hl[j] = w >> 16;
      }
      const bt_u32x4 bh = {hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16), hh[4] | (hh[5] << 16), hh[6] | (hh[7] << 16)};
      const bt_u32x4 blo = {hl[0] | (hl[1] << 16), hl[2] | (hl[3] << 16), hl[4] | (hl[5] << 16), hl[6] | (hl[7] << 16)};
      bt_f32x4 acc = {b2q[0], b2q[1], b2q[2], b2q[3]};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, al), __builtin_bit_cast(bt_bf16x8, bh), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, ah), __builtin_bit_cast(bt_bf16x8, blo), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, ah), __builtin_bit_cast(bt_bf16x8, bh), acc, 0, 0, 0);
      float op = w4q[0] * nl_lrelu(acc[0]);
      op = fmaf(w4q[1], nl_lrelu(acc[1]), op); op = fmaf(w4q[2], nl_lrelu(acc[2]), op); op = fmaf(w4q[3], nl_lrelu(acc[3]), op);
      op += __shfl_xor(op, 16, 64);
      op += __shfl_xor(op, 32, 64);
      o = vis ? op + b4v : -1e9f;
    }
    const float mn = fmaxf(m, o);
    const float sc = expf(m - mn), e = expf(o - mn);
    den = fmaf(den, sc, e);
    cr = fmaf(cr, sc, cv.x * e); cg = fmaf(cg, sc, cv.y * e); cb = fmaf(cb, sc, cv.z * e);
    m = mn;
  }
  if (live && g == 0) {
    const float inv = alive ? 1.f / den : 0.f;
    rgb_s[3 * (size_t)n] = cr * inv; rgb_s[3 * (size_t)n + 1] = cg * inv; rgb_s[3 * (size_t)n + 2] = cb * inv;
  }
}
}  // namespace

int nl_launch_blend_taps(const NlViews& vw, const float* viewsdev, const float* pfeat, const float* blw, const float* xyz, const float* hA, const float* rgbv, int64_t N,
                         const float* w2, const float* b2, const float* w4, const float* b4, float* rgb_s, hipStream_t st, const int* n_alive, int S) {
  if (N <= 0) return NL_OK;
#ifdef NL_BLEND_TAPS_V1   // (A/B builds: the lane-per-sample kernel of round 4)
  hipLaunchKernelGGL(blend_taps_kernel, dim3((unsigned)nl_cdiv(N, 256)), dim3(256), 0, st, vw, viewsdev, pfeat, blw, xyz, hA, rgbv, (int)N, w2, b2, w4, b4, rgb_s, n_alive, S);
#else
  const size_t lds = (288 + (size_t)4 * 16 * vw.V * BT_SLOT) * sizeof(float);   // <= 38 KB at 16 views
  hipLaunchKernelGGL(blend_taps_mfma_kernel, dim3((unsigned)nl_cdiv(N, 64)), dim3(256), lds, st, vw, viewsdev, pfeat, blw, xyz, hA, rgbv, (int)N, w2, b2, w4, b4, rgb_s, n_alive, S);
#endif
  NL_LAUNCH_CHECK();
  return NL_OK;
}

// `ft` (R*S, C) is composited into feat_dst (R, C) (any channel count; the caller passes feat_mlp's hidden layer, see abi.hip)
int nl_launch_composite(const float* z_vals, const float* sigma, const float* rgb_s, const float* ft, const int* valid_s,
                        int64_t R, int S, int C, int white_bkgd, const nl_render_out* out, int64_t ray0, float* feat_dst, float* wsum_dst,
                        hipStream_t st, const int* n_alive, float* w_scratch) {
  if (R <= 0) return NL_OK;
  if (S > 256) return NL_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nl_cdiv(R, 4));
  float* o_rgb = out->rgb ? out->rgb + 3 * ray0 : nullptr;
  float* o_depth = out->depth ? out->depth + ray0 : nullptr;
  float* o_w = out->weights ? out->weights + ray0 * S : w_scratch;   // (w_scratch: the samples' weights for feat_comp_mx_kernel when the caller does not take them)
  unsigned char* o_mask = out->mask ? out->mask + ray0 : nullptr;
  float* o_unc = out->depth_uncertainty ? out->depth_uncertainty + ray0 : nullptr;
#define NL_COMP(CH) hipLaunchKernelGGL(composite_kernel<CH>, grid, dim3(256), 0, st, z_vals, sigma, rgb_s, ft, valid_s, (int)R, S, C, \
                                       white_bkgd, o_rgb, o_depth, o_w, o_mask, o_unc, feat_dst, wsum_dst, n_alive)
  if (S <= 64) NL_COMP(1);
  else if (S <= 128) NL_COMP(2);
  else if (S <= 192) NL_COMP(3);
  else NL_COMP(4);
#undef NL_COMP
  NL_LAUNCH_CHECK();
  return NL_OK;
}
