// Exact K-nearest support points on a uniform grid.
//
// Result definition (identical to the reference's CPU op, ops/knn/src/knn_cpu.cpp:13-64, after the
// ascending sort of ops/knn/knn_utils.py:60-74): the K lexicographically smallest (dist2, idx) tuples,
// ascending, where dist2 = ((dx*dx) + dy*dy) + dz*dz in fp32 WITHOUT fma contraction.  Because the
// definition is order independent, the grid may visit candidates in any order: each candidate is packed
// as key = (bits(dist2) << 32) | idx (dist2 >= 0 so its bits are monotone) and a sorted list of K keys is
// kept per query in registers.  Slots k >= M stay (dist2 = 0, idx = 0) like the reference's zero fill.
//
// Grid: per frame, points are counting-sorted by cell (x fastest), so a row of cells in x is one
// contiguous range of the sorted array.  Search expands Chebyshev shells around the query's (clamped) cell
// until the K-th best dist2 is strictly below a conservative lower bound of everything unvisited.
#include "common.h"

namespace {

__global__ void knn_bbox_kernel(const float* __restrict__ xyz, int M, NlGridParams* gp, int target_cells_per_axis_max) {
  __shared__ float smin[3][256], smax[3][256];
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float v = xyz[3 * (size_t)i + d];
      mn[d] = fminf(mn[d], v);
      mx[d] = fmaxf(mx[d], v);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { smin[d][threadIdx.x] = mn[d]; smax[d][threadIdx.x] = mx[d]; }
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        smin[d][threadIdx.x] = fminf(smin[d][threadIdx.x], smin[d][threadIdx.x + s]);
        smax[d][threadIdx.x] = fmaxf(smax[d][threadIdx.x], smax[d][threadIdx.x + s]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float ext = 0.f;
    for (int d = 0; d < 3; ++d) ext = fmaxf(ext, smax[d][0] - smin[d][0]);
    // ~4 cells per point along the longest axis budget: G^3 ~ 4M, capped
    int G = (int)ceilf(cbrtf(4.0f * (float)(M > 1 ? M : 1)));
    if (G > target_cells_per_axis_max) G = target_cells_per_axis_max;
    if (G < 1) G = 1;
    float cell = ext > 0.f ? ext / (float)G : 1.f;
    cell *= 1.0001f;
    gp->cell = cell;
    gp->inv_cell = 1.f / cell;
    int nc = 1;
    for (int d = 0; d < 3; ++d) {
      gp->origin[d] = smin[d][0];
      gp->bmax[d] = smax[d][0];
      int n = (int)floorf((smax[d][0] - smin[d][0]) / cell) + 1;
      if (n > target_cells_per_axis_max) n = target_cells_per_axis_max;
      if (n < 1) n = 1;
      gp->dims[d] = n;
      nc *= n;
    }
    gp->ncells = nc;
  }
}

__device__ __forceinline__ int cell_coord(float v, float origin, float inv_cell, int n) {
  int c = (int)floorf((v - origin) * inv_cell);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ void knn_count_kernel(const float* __restrict__ xyz, int M, const NlGridParams* __restrict__ gp,
                                 int* __restrict__ counts, int* __restrict__ cell_of) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  int cx = cell_coord(xyz[3 * (size_t)i + 0], gp->origin[0], gp->inv_cell, gp->dims[0]);
  int cy = cell_coord(xyz[3 * (size_t)i + 1], gp->origin[1], gp->inv_cell, gp->dims[1]);
  int cz = cell_coord(xyz[3 * (size_t)i + 2], gp->origin[2], gp->inv_cell, gp->dims[2]);
  int cid = (cz * gp->dims[1] + cy) * gp->dims[0] + cx;
  cell_of[i] = cid;
  atomicAdd(&counts[cid], 1);
}

// single-block exclusive scan over ncells (<= 64^3); starts[ncells] = M
__global__ void knn_scan_kernel(const int* __restrict__ counts, int* __restrict__ starts, int* __restrict__ cursor,
                                const NlGridParams* __restrict__ gp) {
  __shared__ int part[1024];
  const int n = gp->ncells;
  const int per = (n + 1023) / 1024;
  const int b = threadIdx.x * per;
  int s = 0;
  for (int i = b; i < b + per && i < n; ++i) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = b; i < b + per && i < n; ++i) {
    starts[i] = run;
    cursor[i] = run;
    run += counts[i];
  }
  if (threadIdx.x == 1023) starts[n] = part[1023];
}

__global__ void knn_scatter_kernel(const float* __restrict__ xyz, int M, const int* __restrict__ cell_of,
                                   int* __restrict__ cursor, float4* __restrict__ sorted) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  int pos = atomicAdd(&cursor[cell_of[i]], 1);
  sorted[pos] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float(i));
}

template <int K>
__device__ __forceinline__ void knn_insert(unsigned long long (&best)[K], unsigned long long key) {
  if (key < best[K - 1]) {
    best[K - 1] = key;
#pragma unroll
    for (int i = K - 1; i > 0; --i) {
      unsigned long long a = best[i - 1], b = best[i];
      bool sw = b < a;
      best[i - 1] = sw ? b : a;
      best[i] = sw ? a : b;
    }
  }
}

template <int K>
__global__ __launch_bounds__(256) void knn_search_kernel(const float* __restrict__ q, int N, const NlGridParams* __restrict__ gpp,
                                                         const int* __restrict__ starts, const float4* __restrict__ sorted,
                                                         int M, int Kout, int* __restrict__ idx_out, float* __restrict__ d2_out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const NlGridParams gp = *gpp;
  const float qx = q[3 * (size_t)n], qy = q[3 * (size_t)n + 1], qz = q[3 * (size_t)n + 2];
  unsigned long long best[K];
#pragma unroll
  for (int i = 0; i < K; ++i) best[i] = ~0ull;

  const int nx = gp.dims[0], ny = gp.dims[1], nz = gp.dims[2];
  const int cx = cell_coord(qx, gp.origin[0], gp.inv_cell, nx);
  const int cy = cell_coord(qy, gp.origin[1], gp.inv_cell, ny);
  const int cz = cell_coord(qz, gp.origin[2], gp.inv_cell, nz);
  // distance from q to the bounding box of all points, per axis (0 inside)
  const float ox = fmaxf(fmaxf(gp.origin[0] - qx, qx - gp.bmax[0]), 0.f);
  const float oy = fmaxf(fmaxf(gp.origin[1] - qy, qy - gp.bmax[1]), 0.f);
  const float oz = fmaxf(fmaxf(gp.origin[2] - qz, qz - gp.bmax[2]), 0.f);
  const float slack = 1e-3f * gp.cell;
  const int rmax = max(max(max(cx, nx - 1 - cx), max(cy, ny - 1 - cy)), max(cz, nz - 1 - cz));

  for (int r = 0; r <= rmax; ++r) {
    const int z0 = max(cz - r, 0), z1 = min(cz + r, nz - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, ny - 1);
    const int x0 = max(cx - r, 0), x1 = min(cx + r, nx - 1);
    for (int z = z0; z <= z1; ++z) {
      const bool zface = (z == cz - r) || (z == cz + r);
      for (int y = y0; y <= y1; ++y) {
        const bool full = zface || (y == cy - r) || (y == cy + r);
        const int rowbase = (z * ny + y) * nx;
        // full row [x0,x1] or just the two end cells (when they are really on the shell)
        int nseg = full ? 1 : 2;
        for (int sgi = 0; sgi < nseg; ++sgi) {
          int xa, xb;
          if (full) { xa = x0; xb = x1; }
          else if (sgi == 0) { if (cx - r < 0) continue; xa = xb = cx - r; }
          else { if (cx + r > nx - 1 || r == 0) continue; xa = xb = cx + r; }
          const int pb = starts[rowbase + xa], pe = starts[rowbase + xb + 1];
          for (int p = pb; p < pe; ++p) {
            const float4 c = sorted[p];
            const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
            float d = __fmul_rn(dx, dx);
            d = __fadd_rn(d, __fmul_rn(dy, dy));
            d = __fadd_rn(d, __fmul_rn(dz, dz));
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)__float_as_int(c.w);
            knn_insert<K>(best, key);
          }
        }
      }
    }
    // conservative lower bound on dist2 of any point in a cell outside the visited box
    if (best[K - 1] != ~0ull) {
      float lb2 = 3.4e38f;
      const float kth = __uint_as_float((unsigned int)(best[K - 1] >> 32));
      if (cx + r + 1 < nx) { float f = fmaxf(gp.origin[0] + (float)(cx + r + 1) * gp.cell - qx - slack, 0.f); lb2 = fminf(lb2, f * f + oy * oy + oz * oz); }
      if (cx - r > 0)      { float f = fmaxf(qx - (gp.origin[0] + (float)(cx - r) * gp.cell) - slack, 0.f);     lb2 = fminf(lb2, f * f + oy * oy + oz * oz); }
      if (cy + r + 1 < ny) { float f = fmaxf(gp.origin[1] + (float)(cy + r + 1) * gp.cell - qy - slack, 0.f); lb2 = fminf(lb2, f * f + ox * ox + oz * oz); }
      if (cy - r > 0)      { float f = fmaxf(qy - (gp.origin[1] + (float)(cy - r) * gp.cell) - slack, 0.f);     lb2 = fminf(lb2, f * f + ox * ox + oz * oz); }
      if (cz + r + 1 < nz) { float f = fmaxf(gp.origin[2] + (float)(cz + r + 1) * gp.cell - qz - slack, 0.f); lb2 = fminf(lb2, f * f + ox * ox + oy * oy); }
      if (cz - r > 0)      { float f = fmaxf(qz - (gp.origin[2] + (float)(cz - r) * gp.cell) - slack, 0.f);     lb2 = fminf(lb2, f * f + ox * ox + oy * oy); }
      if (kth < lb2 * 0.999999f) break;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool ok = best[k] != ~0ull;
    if (k < Kout) {
      idx_out[(size_t)n * Kout + k] = ok ? (int)(unsigned int)(best[k] & 0xffffffffull) : 0;
      d2_out[(size_t)n * Kout + k] = ok ? __uint_as_float((unsigned int)(best[k] >> 32)) : 0.f;
    }
  }
}

}  // namespace

// ---- host side (called from abi.hip) ------------------------------------------------------------
struct NlKnnGrid {
  NlGridParams* params;  // device
  int* starts;           // device [max_cells + 1]
  int* counts;           // device [max_cells]
  int* cursor;           // device [max_cells]
  int* cell_of;          // device [M]
  float4* sorted;        // device [M]
  int M;
};

constexpr int NL_GRID_MAX_AXIS = 64;
constexpr int NL_GRID_MAX_CELLS = NL_GRID_MAX_AXIS * NL_GRID_MAX_AXIS * NL_GRID_MAX_AXIS;

size_t nl_knn_grid_bytes(int64_t M) {
  size_t b = 0;
  b += nl_align_up(sizeof(NlGridParams), 256);
  b += nl_align_up(sizeof(int) * (NL_GRID_MAX_CELLS + 1), 256);
  b += 2 * nl_align_up(sizeof(int) * NL_GRID_MAX_CELLS, 256);
  b += nl_align_up(sizeof(int) * (size_t)M, 256);
  b += nl_align_up(sizeof(float4) * (size_t)M, 256);
  return b;
}

int nl_knn_grid_build(NlKnnGrid* g, void* mem, const float* xyz, int64_t M, hipStream_t st) {
  char* p = (char*)mem;
  g->params = (NlGridParams*)p; p += nl_align_up(sizeof(NlGridParams), 256);
  g->starts = (int*)p; p += nl_align_up(sizeof(int) * (NL_GRID_MAX_CELLS + 1), 256);
  g->counts = (int*)p; p += nl_align_up(sizeof(int) * NL_GRID_MAX_CELLS, 256);
  g->cursor = (int*)p; p += nl_align_up(sizeof(int) * NL_GRID_MAX_CELLS, 256);
  g->cell_of = (int*)p; p += nl_align_up(sizeof(int) * (size_t)M, 256);
  g->sorted = (float4*)p;
  g->M = (int)M;
  if (M <= 0) return NL_OK;
  NL_CHECK_HIP(hipMemsetAsync(g->counts, 0, sizeof(int) * NL_GRID_MAX_CELLS, st));
  hipLaunchKernelGGL(knn_bbox_kernel, dim3(1), dim3(256), 0, st, xyz, (int)M, g->params, NL_GRID_MAX_AXIS);
  int nb = (int)nl_cdiv(M, 256);
  hipLaunchKernelGGL(knn_count_kernel, dim3(nb), dim3(256), 0, st, xyz, (int)M, g->params, g->counts, g->cell_of);
  hipLaunchKernelGGL(knn_scan_kernel, dim3(1), dim3(1024), 0, st, g->counts, g->starts, g->cursor, g->params);
  hipLaunchKernelGGL(knn_scatter_kernel, dim3(nb), dim3(256), 0, st, xyz, (int)M, g->cell_of, g->cursor, g->sorted);
  NL_LAUNCH_CHECK();
  return NL_OK;
}

int nl_knn_search(const NlKnnGrid* g, const float* xyz, int64_t N, int K, int* idx, float* d2, hipStream_t st) {
  if (N <= 0) return NL_OK;
  if (g->M <= 0) {
    NL_CHECK_HIP(hipMemsetAsync(idx, 0, sizeof(int) * (size_t)N * K, st));
    NL_CHECK_HIP(hipMemsetAsync(d2, 0, sizeof(float) * (size_t)N * K, st));
    return NL_OK;
  }
  dim3 grid((unsigned)nl_cdiv(N, 256));
  if (K == 8)
    hipLaunchKernelGGL(knn_search_kernel<8>, grid, dim3(256), 0, st, xyz, (int)N, g->params, g->starts, g->sorted, g->M, K, idx, d2);
  else if (K == 1)
    hipLaunchKernelGGL(knn_search_kernel<1>, grid, dim3(256), 0, st, xyz, (int)N, g->params, g->starts, g->sorted, g->M, K, idx, d2);
  else if (K <= 8)
    hipLaunchKernelGGL(knn_search_kernel<8>, grid, dim3(256), 0, st, xyz, (int)N, g->params, g->starts, g->sorted, g->M, K, idx, d2);
  else
    return NL_ERR_UNSUPPORTED;
  NL_LAUNCH_CHECK();
  return NL_OK;
}
