// Per-frame setup on the GPU: what the reference runs once per query frame before any ray is rendered.
//
//   nl_cross_view_features   DepthFusionNet's hand-made input channels (conditional_nerf/depth_fusion.py:150-227, 269-278):
//                            normalised inverse depth + the cross-view colour / inverse-depth consistency statistics, written
//                            straight into the (V,12,H,W) tensor the per-frame CNN consumes.  One thread per (view, pixel): lift the
//                            pixel with its own depth, project into all V views, bilinear taps (border, align_corners), masked
//                            mean / variance over the views.  The reference materialises (V, V*H*W, c) tensors for this; here
//                            nothing but the 12 output channels (and one depth plane) touches HBM.
//   nl_backproject_support   ConditionalNeRF.backproject_support_frame (conditional_nerf/model.py:203-265): every valid depth pixel
//                            of every support view at one stride becomes a neural point.  The reference does a nonzero() + ~25
//                            tensor ops per view (each nonzero is a device sync); here: count per image row, one scan, one fill —
//                            order-preserving (view-major, then row-major), so the tables are index-compatible with the reference's.
//
// Both are HBM/latency-trivial (a few MB, tens of microseconds); what they remove is ~700 framework launches and 2V syncs.
#include "common.h"
#include "mvdec.h"

namespace {

constexpr int VM_STRIDE = 80;   // floats per view in the derived-matrix table
constexpr int VM_KINV = 0, VM_W2C = 9, VM_KRT = 21, VM_RC = 33, VM_TC = 42, VM_S2R = 45, VM_KS = 57, VM_C2W = 61;
constexpr int SETUP_MAX_V = 16;

// Gauss-Jordan with partial pivoting in double, rounded once to fp32 (torch.inverse is an fp32 LU: agreement ~1e-7 relative)
template <int N>
__device__ void invert(const float* a, float* out) {
  double m[N][2 * N];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) { m[r][c] = a[r * N + c]; m[r][N + c] = r == c ? 1.0 : 0.0; }
  for (int c = 0; c < N; ++c) {
    int p = c;
    for (int r = c + 1; r < N; ++r) if (fabs(m[r][c]) > fabs(m[p][c])) p = r;
    if (p != c) for (int k = 0; k < 2 * N; ++k) { double t = m[c][k]; m[c][k] = m[p][k]; m[p][k] = t; }
    const double d = 1.0 / m[c][c];
    for (int k = 0; k < 2 * N; ++k) m[c][k] *= d;
    for (int r = 0; r < N; ++r) {
      if (r == c) continue;
      const double f = m[r][c];
      for (int k = 0; k < 2 * N; ++k) m[r][k] -= f * m[c][k];
    }
  }
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) out[r * N + c] = (float)m[r][N + c];
}

// One thread per view: every small matrix the two kernels need, in the reference's composition order.
__global__ void view_matrices_kernel(const float* __restrict__ Ks, const float* __restrict__ c2w, int V, float stride,
                                     float* __restrict__ vm) {
  const int v = threadIdx.x;
  if (v >= V) return;
  float* o = vm + (size_t)v * VM_STRIDE;
  float K[9], P[16], Pi[16], P0i[16];
  for (int i = 0; i < 9; ++i) K[i] = Ks[9 * v + i];
  for (int i = 0; i < 6; ++i) K[i] = K[i] / stride;                       // K[:2] /= stride (model.py:228-229)
  for (int i = 0; i < 16; ++i) P[i] = c2w[16 * v + i];
  invert<3>(K, o + VM_KINV);
  invert<4>(P, Pi);                                                        // poses.inverse() (depth_fusion.py:273)
  invert<4>(c2w, P0i);                                                     // w2c of the reference view (model.py:222)
  for (int i = 0; i < 12; ++i) o[VM_W2C + i] = Pi[i];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {                                          // K @ Rt (depth_fusion.py:91)
      float a = K[3 * r] * Pi[c];
      a += K[3 * r + 1] * Pi[4 + c];
      a += K[3 * r + 2] * Pi[8 + c];
      o[VM_KRT + 4 * r + c] = a;
    }
  for (int r = 0; r < 3; ++r)                                              // R = Rt[:3,:3]^T, t = -R @ Rt[:,3] (depth_fusion.py:160-161)
    for (int c = 0; c < 3; ++c) o[VM_RC + 3 * r + c] = Pi[4 * c + r];
  for (int r = 0; r < 3; ++r) {
    float a = -o[VM_RC + 3 * r] * Pi[3];
    a += -o[VM_RC + 3 * r + 1] * Pi[7];
    a += -o[VM_RC + 3 * r + 2] * Pi[11];
    o[VM_TC + r] = a;
  }
  for (int r = 0; r < 3; ++r)                                              // src2ref = w2c_ref @ c2w (model.py:241)
    for (int c = 0; c < 4; ++c) {
      float a = P0i[4 * r] * P[c];
      a += P0i[4 * r + 1] * P[4 + c];
      a += P0i[4 * r + 2] * P[8 + c];
      a += P0i[4 * r + 3] * P[12 + c];
      o[VM_S2R + 4 * r + c] = a;
    }
  o[VM_KS] = K[0]; o[VM_KS + 1] = K[4]; o[VM_KS + 2] = K[2]; o[VM_KS + 3] = K[5];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) o[VM_C2W + 4 * r + c] = P[4 * r + c];
}

// extract_depth_for_init (depth_fusion.py:209-227) and the metric depth get_diff_feats re-derives from it (:172-175)
__global__ void depth_planes_kernel(const float* __restrict__ imgs, const float* __restrict__ depths, int V, int HW, float near_, float far_,
                                    float* __restrict__ cnn_in, float* __restrict__ depth_rec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)V * HW) return;
  const int v = (int)(i / HW), p = (int)(i - (int64_t)v * HW);
  const float ni = -1.f / near_, fi = -1.f / far_;
  float d = fmaxf(depths[i], 1e-5f);
  d = -1.f / d;
  d = (d - ni) / (fi - ni);
  d = fminf(fmaxf(d, 0.f), 1.f);
  float* o = cnn_in + (size_t)v * 12 * HW + p;
  o[0] = imgs[((size_t)v * 3) * HW + p];
  o[(size_t)HW] = imgs[((size_t)v * 3 + 1) * HW + p];
  o[(size_t)2 * HW] = imgs[((size_t)v * 3 + 2) * HW + p];
  o[(size_t)3 * HW] = d;
  depth_rec[i] = -1.f / (d * (fi - ni) + ni);
}

// get_diff_feats (depth_fusion.py:163-207)
__global__ __launch_bounds__(256) void cross_view_kernel(const float* __restrict__ imgs, const float* __restrict__ depth_rec,
                                                         const float* __restrict__ vm, int V, int H, int W, float near_, float far_,
                                                         float* __restrict__ cnn_in) {
  const int HW = H * W;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)V * HW) return;
  const int u = (int)(i / HW), p = (int)(i - (int64_t)u * HW);
  const int py = p / W, px = p - py * W;
  const float* mu = vm + (size_t)u * VM_STRIDE;
  const float dep = depth_rec[i];
  // depth2pts3d: K^-1 (x d, y d, d), then the camera-to-world re-derived from the w2c
  const float a0 = (float)px * dep, a1 = (float)py * dep, a2 = dep;
  float pc[3], pw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) pc[r] = (mu[VM_KINV + 3 * r] * a0 + mu[VM_KINV + 3 * r + 1] * a1) + mu[VM_KINV + 3 * r + 2] * a2;
#pragma unroll
  for (int r = 0; r < 3; ++r) pw[r] = ((mu[VM_RC + 3 * r] * pc[0] + mu[VM_RC + 3 * r + 1] * pc[1]) + mu[VM_RC + 3 * r + 2] * pc[2]) + mu[VM_TC + r];
  const float own[3] = {imgs[((size_t)u * 3) * HW + p], imgs[((size_t)u * 3 + 1) * HW + p], imgs[((size_t)u * 3 + 2) * HW + p]};
  const float ni = -1.f / near_, fi = -1.f / far_;
  float x[SETUP_MAX_V][4], m[SETUP_MAX_V];
  float msum = 0.f, s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < SETUP_MAX_V; ++v) {
    if (v < V) {
      const float* mv = vm + (size_t)v * VM_STRIDE + VM_KRT;
      float cam[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) cam[r] = ((mv[4 * r] * pw[0] + mv[4 * r + 1] * pw[1]) + mv[4 * r + 2] * pw[2]) + mv[4 * r + 3];
      float z = cam[2];
      const bool bad = fabsf(z) < 1e-4f;
      if (bad) z = 1e-3f;
      const float qx = cam[0] / z, qy = cam[1] / z;
      const bool outside = (qx < -0.5f) | (qx >= (float)W - 0.5f) | (qy < -0.5f) | (qy >= (float)H - 0.5f);
      const float mk = (!bad && !outside) ? 1.f : 0.f;
      // interpolate_feats(border, align_corners=True): pixel -> [-1,1] -> ATen's un-normalisation, clip, 4 taps
      const float xn = qx / (float)(W - 1) * 2.f - 1.f, yn = qy / (float)(H - 1) * 2.f - 1.f;
      const nlmv::Taps t = nlmv::make_taps<true, true>(xn, yn, W, H);
      const int x0 = t.x0, y0 = t.y0, x1 = t.me ? x0 + 1 : x0, y1 = t.ms ? y0 + 1 : y0;
      const float wnw = t.nw, wne = t.me ? t.ne : 0.f, wsw = t.ms ? t.sw : 0.f, wse = (t.me && t.ms) ? t.se : 0.f;
      const int o00 = y0 * W + x0, o01 = y0 * W + x1, o10 = y1 * W + x0, o11 = y1 * W + x1;
      const float* dp = depth_rec + (size_t)v * HW;
      float di = ((dp[o00] * wnw + dp[o01] * wne) + dp[o10] * wsw) + dp[o11] * wse;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* ip = imgs + ((size_t)v * 3 + c) * HW;
        const float ci = ((ip[o00] * wnw + ip[o01] * wne) + ip[o10] * wsw) + ip[o11] * wse;
        x[v][c] = fabsf(ci - own[c]);
      }
      di = fmaxf(di, 1e-5f);
      const float pd = fmaxf(z, 1e-5f);
      float dd = fabsf(-1.f / di + 1.f / pd);
      dd = fminf(dd / (fi - ni), 1.5f);
      x[v][3] = dd;
      m[v] = mk;
      msum += mk;
#pragma unroll
      for (int c = 0; c < 4; ++c) s[c] += x[v][c] * mk;
    }
  }
  // masked_mean_var over the views (neuray_ops.py:38-43)
  const float den = fmaxf(msum, 1e-4f);
  float mean[4], var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) mean[c] = s[c] / den;
#pragma unroll
  for (int v = 0; v < SETUP_MAX_V; ++v)
    if (v < V) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float e = x[v][c] - mean[c]; var[c] += e * e * m[v]; }
    }
  float* o = cnn_in + (size_t)u * 12 * HW + p;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[(size_t)(4 + c) * HW] = mean[c];
    o[(size_t)(7 + c) * HW] = var[c] / den;
  }
  o[(size_t)10 * HW] = mean[3];
  o[(size_t)11 * HW] = var[3] / den;
}

// ------------------------------------------------------------------------------------------------ back-projection
// F.interpolate(mode="nearest") source index for an explicit output size (ATen nearest_neighbor_compute_source_index)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  const int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

__global__ void bp_count_kernel(const float* __restrict__ depths, int H, int W, int Hs, int Ws, float sy, float sx, int* __restrict__ rowcnt) {
  const int row = blockIdx.x, v = row / Hs, y = row - v * Hs;
  const float* dp = depths + ((size_t)v * H + nearest_src(y, sy, H)) * W;
  int c = 0;
  for (int x0 = 0; x0 < Ws; x0 += 64) {
    const int xx = x0 + (int)threadIdx.x;
    const bool ok = xx < Ws && dp[nearest_src(xx < Ws ? xx : 0, sx, W)] > 0.f;
    c += __popcll(__ballot(ok));
  }
  if (threadIdx.x == 0) rowcnt[row] = c;
}

// exclusive scan of the row counts, in place; rowcnt[rows] = total
__global__ void bp_scan_kernel(int* __restrict__ rowcnt, int rows) {
  __shared__ int part[256];
  const int t = threadIdx.x, per = (rows + 255) / 256, b = t * per, e = min(b + per, rows);
  int s = 0;
  for (int i = b; i < e; ++i) s += rowcnt[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int a = 0;
    for (int i = 0; i < 256; ++i) { const int c = part[i]; part[i] = a; a += c; }
    rowcnt[rows] = a;
  }
  __syncthreads();
  int a = part[t];
  for (int i = b; i < e; ++i) { const int c = rowcnt[i]; rowcnt[i] = a; a += c; }
}

constexpr int BP_MAX_W = 4096;

__global__ __launch_bounds__(256) void bp_fill_kernel(const float* __restrict__ imgs, const float* __restrict__ feats, const float* __restrict__ depths,
                                                      const float* __restrict__ vm, const int* __restrict__ rowoff, int H, int W, int Hs, int Ws,
                                                      int fh, int fw, int C, float sy, float sx, int capacity, float* __restrict__ feature,
                                                      float* __restrict__ xyz, float* __restrict__ xyz_ref, float* __restrict__ direction) {
  __shared__ unsigned short cols[BP_MAX_W];
  __shared__ int s_cnt;
  const int row = blockIdx.x, v = row / Hs, y = row - v * Hs;
  const int ysrc = nearest_src(y, sy, H);
  const float* dp = depths + ((size_t)v * H + ysrc) * W;
  if (threadIdx.x < 64) {   // wave 0: ordered compaction of the row
    int base = 0;
    for (int x0 = 0; x0 < Ws; x0 += 64) {
      const int xx = x0 + (int)threadIdx.x;
      const bool ok = xx < Ws && dp[nearest_src(xx < Ws ? xx : 0, sx, W)] > 0.f;
      const unsigned long long b = __ballot(ok);
      if (ok) cols[base + __popcll(b & ((1ull << threadIdx.x) - 1ull))] = (unsigned short)xx;
      base += __popcll(b);
    }
    if (threadIdx.x == 0) s_cnt = base;
  }
  __syncthreads();
  const int cnt = s_cnt, off = rowoff[row];
  if (off + cnt > capacity) return;   // caller sized the tables too small: nl_backproject_support reports it from the total
  const float* mv = vm + (size_t)v * VM_STRIDE;
  // geometry: one thread per point (model.py:231-252, utils.py:56-70)
  for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
    const int x = cols[j];
    const float z = dp[nearest_src(x, sx, W)];
    const float fu = (float)x, fv = (float)y;
    float cam[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) cam[r] = ((mv[VM_KINV + 3 * r] * fu + mv[VM_KINV + 3 * r + 1] * fv) + mv[VM_KINV + 3 * r + 2]) * z;
    const size_t q = (size_t)(off + j);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float* c = mv + VM_C2W + 4 * r;
      xyz[3 * q + r] = ((c[0] * cam[0] + c[1] * cam[1]) + c[2] * cam[2]) + c[3];
      const float* s = mv + VM_S2R + 4 * r;
      xyz_ref[3 * q + r] = ((s[0] * cam[0] + s[1] * cam[1]) + s[2] * cam[2]) + s[3];
    }
    const float d0 = (fu - mv[VM_KS + 2]) / mv[VM_KS], d1 = (fv - mv[VM_KS + 3]) / mv[VM_KS + 1];
    float rd[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) rd[r] = (d0 * mv[VM_C2W + 4 * r] + d1 * mv[VM_C2W + 4 * r + 1]) + mv[VM_C2W + 4 * r + 2];
    const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
    direction[4 * q] = rd[0] / nrm; direction[4 * q + 1] = rd[1] / nrm; direction[4 * q + 2] = rd[2] / nrm;
    direction[4 * q + 3] = z;
  }
  // descriptors: one wave per point, lanes over the 3 + C channels (coalesced rows of the channels-last feature map)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int F = 3 + C;
  const size_t HW = (size_t)H * W;
  for (int j = wave; j < cnt; j += nw) {
    const int x = cols[j];
    const int xsrc = nearest_src(x, sx, W);
    const float* fp = feats + (((size_t)v * fh + y) * fw + x) * C;
    float* o = feature + (size_t)(off + j) * F;
    for (int c = lane; c < F; c += 64)
      o[c] = c < 3 ? imgs[((size_t)v * 3 + c) * HW + (size_t)ysrc * W + xsrc] : fp[c - 3];
  }
}

// ------------------------------------------------------------------------------------------------ ray generation (row a1)
// get_rays (conditional_nerf/utils.py:56-70) for the whole pixel grid (uv == null: ray r = pixel (r % W, r / W)) or
// points_2d_to_rays (model.py:687-700) for a list of pixel positions (truncated towards zero like .long()).
__global__ void rays_kernel(const float* __restrict__ K, const float* __restrict__ c2w, const float* __restrict__ uv, int W, int64_t R,
                            float* __restrict__ rays_o, float* __restrict__ rays_d) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float u, v;
  if (uv) { u = (float)(long long)uv[2 * r]; v = (float)(long long)uv[2 * r + 1]; }
  else { u = (float)(r % W); v = (float)(r / W); }
  const float d0 = (u - K[2]) / K[0], d1 = (v - K[5]) / K[4];
  float rd[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) rd[i] = (d0 * c2w[4 * i] + d1 * c2w[4 * i + 1]) + c2w[4 * i + 2];
  const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) { rays_d[3 * r + i] = rd[i] / nrm; rays_o[3 * r + i] = c2w[4 * i + 3]; }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C-ABI
size_t nl_setup_workspace_bytes(int V, int H, int W, int stride) {
  if (V < 1 || H < 1 || W < 1 || stride < 1) return 0;
  const size_t Hs = (size_t)(H / stride);
  return 256 + sizeof(float) * ((size_t)SETUP_MAX_V * VM_STRIDE + (size_t)V * H * W) + sizeof(int) * ((size_t)V * Hs + 64);
}

int nl_cross_view_features(const float* imgs, const float* depths, const float* Ks, const float* c2w, int V, int H, int W, float near_,
                           float far_, float* cnn_in, void* workspace, size_t workspace_bytes, void* stream) {
  if (!imgs || !depths || !Ks || !c2w || !cnn_in || !workspace || V < 1 || H < 2 || W < 2) return NL_ERR_BAD_ARG;
  if (V > SETUP_MAX_V) return NL_ERR_UNSUPPORTED;
  if (workspace_bytes < nl_setup_workspace_bytes(V, H, W, 1)) return NL_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* vm = (float*)workspace;
  float* depth_rec = vm + SETUP_MAX_V * VM_STRIDE;
  const int64_t n = (int64_t)V * H * W;
  const int blocks = (int)((n + 255) / 256);
  view_matrices_kernel<<<1, 64, 0, st>>>(Ks, c2w, V, 1.f, vm);
  depth_planes_kernel<<<blocks, 256, 0, st>>>(imgs, depths, V, H * W, near_, far_, cnn_in, depth_rec);
  cross_view_kernel<<<blocks, 256, 0, st>>>(imgs, depth_rec, vm, V, H, W, near_, far_, cnn_in);
  return hipGetLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}

int nl_backproject_support(const float* imgs, const float* feats, const float* depths, const float* Ks, const float* c2w, int V, int H, int W,
                           int fh, int fw, int C, int stride, int64_t capacity, float* feature, float* xyz, float* xyz_ref, float* direction,
                           int64_t* m_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!imgs || !feats || !depths || !Ks || !c2w || !m_out || !workspace || V < 1 || H < 1 || W < 1 || C < 1 || stride < 1 || capacity < 0)
    return NL_ERR_BAD_ARG;
  const int Hs = H / stride, Ws = W / stride;   // int(H / stride) (model.py:225-226)
  if (Hs < 1 || Ws < 1 || fh < Hs || fw < Ws) return NL_ERR_BAD_ARG;
  if (capacity > 0 && (!feature || !xyz || !xyz_ref || !direction)) return NL_ERR_BAD_ARG;
  if (V > SETUP_MAX_V || Ws > BP_MAX_W || (int64_t)V * Hs * Ws > 0x7fffffffll) return NL_ERR_UNSUPPORTED;
  if (workspace_bytes < nl_setup_workspace_bytes(V, H, W, stride)) return NL_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* vm = (float*)workspace;
  int* rowoff = (int*)(vm + SETUP_MAX_V * VM_STRIDE);
  const int rows = V * Hs;
  const float sy = (float)H / (float)Hs, sx = (float)W / (float)Ws;
  view_matrices_kernel<<<1, 64, 0, st>>>(Ks, c2w, V, (float)stride, vm);
  bp_count_kernel<<<rows, 64, 0, st>>>(depths, H, W, Hs, Ws, sy, sx, rowoff);
  bp_scan_kernel<<<1, 256, 0, st>>>(rowoff, rows);
  int total = 0;
  if (hipMemcpyAsync(&total, rowoff + rows, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return NL_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return NL_ERR_HIP;   // the reference's nonzero() syncs V times per level; this is the one sync
  *m_out = total;
  if (total > capacity) return NL_ERR_WORKSPACE;                     // tables too small: *m_out says how many rows are needed
  if (total == 0) return NL_OK;
  bp_fill_kernel<<<rows, 256, 0, st>>>(imgs, feats, depths, vm, rowoff, H, W, Hs, Ws, fh, fw, C, sy, sx, (int)capacity, feature, xyz, xyz_ref,
                                       direction);
  return hipGetLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}

int nl_get_rays(const float* K, const float* c2w, const float* uv, int H, int W, int64_t R, float* rays_o, float* rays_d, void* stream) {
  if (R == 0) return NL_OK;
  if (!K || !c2w || !rays_o || !rays_d || R < 0 || H < 1 || W < 1 || (!uv && R != (int64_t)H * W)) return NL_ERR_BAD_ARG;
  rays_kernel<<<(unsigned)((R + 255) / 256), 256, 0, (hipStream_t)stream>>>(K, c2w, uv, W, R, rays_o, rays_d);
  return hipGetLastError() == hipSuccess ? NL_OK : NL_ERR_HIP;
}
