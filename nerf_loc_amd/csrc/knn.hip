// Exact K-nearest support points: wave-per-query search over an implicit Morton octree.
//
// Result definition (identical to the reference's CPU op, ops/knn/src/knn_cpu.cpp:13-64, after the ascending
// sort of ops/knn/knn_utils.py:60-74): the K lexicographically smallest (dist2, idx) tuples, ascending, where
// dist2 = ((dx*dx) + dy*dy) + dz*dz in fp32 WITHOUT fma contraction.  The definition is order independent, so
// candidates may be visited in any order: each is packed as key = (bits(dist2) << 32) | idx (dist2 >= 0, so its
// bits are monotone) and the K smallest keys win.  Slots k >= M stay (dist2 = 0, idx = 0) like the reference.
//
// Structure (per frame): points are counting-sorted by the 18-bit Morton code of their cell in a fixed 64^3 grid
// over the cloud's bounding box, so an octree node of ANY level l (side 2^l fine cells, Morton prefix m) is the
// contiguous range [starts[m << 3l], starts[(m+1) << 3l]) of the sorted array — no explicit tree is stored.
// Search (one wave64 per query, wave-uniform control flow):
//   phase 1  greedy descent: at every level the 8 children are tested by 8 lanes and the nearest child that still
//            holds >= K points is entered; the node where this stops (<= 8(K-1) points unless a fine cell is
//            dense) is scanned, giving an upper bound U on the K-th neighbour's dist2.
//   phase 2  breadth-first descent from the root, 8 nodes x 8 children per step, keeping children with points and
//            mindist2(q, child box) <= U (boxes widened by a rounding slack); small / fine nodes become leaf ranges
//            that are flattened with a wave prefix sum and scanned 64 candidates at a time (coalesced float4
//            loads); the K best keys live replicated in registers and are updated by a wave-min selection loop;
//            U tightens as soon as K real candidates are known.
// Typical cost is ~1.5k wave instructions per query, independent of how far the query is from the cloud.
#include "common.h"

namespace {

constexpr int GRID_BITS = 6;
constexpr int GRID_N = 1 << GRID_BITS;              // 64 cells per axis
constexpr int GRID_CELLS = GRID_N * GRID_N * GRID_N;  // 262144

__device__ __forceinline__ unsigned part1by2(unsigned x) {
  x &= 0x3ffu;
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}
__device__ __forceinline__ unsigned morton3(unsigned x, unsigned y, unsigned z) {
  return part1by2(x) | (part1by2(y) << 1) | (part1by2(z) << 2);
}

__global__ void knn_bbox_kernel(const float* __restrict__ xyz, int M, NlGridParams* gp) {
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
    float cell = ext > 0.f ? ext / (float)GRID_N : 1.f;
    cell *= 1.0001f;
    gp->cell = cell;
    gp->inv_cell = 1.f / cell;
    for (int d = 0; d < 3; ++d) {
      gp->origin[d] = smin[d][0];
      gp->bmax[d] = smax[d][0];
      gp->dims[d] = GRID_N;
    }
    gp->ncells = GRID_CELLS;
  }
}

__device__ __forceinline__ int cell_coord(float v, float origin, float inv_cell) {
  int c = (int)floorf((v - origin) * inv_cell);
  return c < 0 ? 0 : (c >= GRID_N ? GRID_N - 1 : c);
}

__global__ void knn_count_kernel(const float* __restrict__ xyz, int M, const NlGridParams* __restrict__ gp,
                                 int* __restrict__ counts, int* __restrict__ cell_of) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  int cx = cell_coord(xyz[3 * (size_t)i + 0], gp->origin[0], gp->inv_cell);
  int cy = cell_coord(xyz[3 * (size_t)i + 1], gp->origin[1], gp->inv_cell);
  int cz = cell_coord(xyz[3 * (size_t)i + 2], gp->origin[2], gp->inv_cell);
  int cid = (int)morton3(cx, cy, cz);
  cell_of[i] = cid;
  atomicAdd(&counts[cid], 1);
}

// single-block exclusive scan over the 262144 cells (Morton order); starts[ncells] = M
__global__ void knn_scan_kernel(const int* __restrict__ counts, int* __restrict__ starts, int* __restrict__ cursor) {
  __shared__ int part[1024];
  const int n = GRID_CELLS;
  const int per = n / 1024;
  const int b = threadIdx.x * per;
  int s = 0;
  for (int i = b; i < b + per; ++i) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = b; i < b + per; ++i) {
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

typedef unsigned long long u64;

__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    u64 t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

template <int K>
__device__ __forceinline__ void insert_sorted(u64 (&best)[K], u64 key) {
  // caller guarantees key < best[K-1]
  best[K - 1] = key;
#pragma unroll
  for (int i = K - 1; i > 0; --i) {
    u64 a = best[i - 1], b = best[i];
    bool sw = b < a;
    best[i - 1] = sw ? b : a;
    best[i] = sw ? a : b;
  }
}

__device__ __forceinline__ unsigned compact1by2(unsigned x) {
  x &= 0x09249249u;
  x = (x ^ (x >> 2)) & 0x030C30C3u;
  x = (x ^ (x >> 4)) & 0x0300F00Fu;
  x = (x ^ (x >> 8)) & 0x030000FFu;
  x = (x ^ (x >> 16)) & 0x3ffu;
  return x;
}

struct QueryCtx {
  float qx, qy, qz, org0, org1, org2, cell, slack;
};

// squared distance from the query to the (slack-widened) box of node (m, level L)
__device__ __forceinline__ float node_mindist2(const QueryCtx& c, unsigned m, int L) {
  const float cl = c.cell * (float)(1 << L);
  const float x0 = c.org0 + (float)compact1by2(m) * cl, y0 = c.org1 + (float)compact1by2(m >> 1) * cl, z0 = c.org2 + (float)compact1by2(m >> 2) * cl;
  const float dx = fmaxf(fmaxf(x0 - c.qx, c.qx - (x0 + cl)) - c.slack, 0.f);
  const float dy = fmaxf(fmaxf(y0 - c.qy, c.qy - (y0 + cl)) - c.slack, 0.f);
  const float dz = fmaxf(fmaxf(z0 - c.qz, c.qz - (z0 + cl)) - c.slack, 0.f);
  return dx * dx + dy * dy + dz * dz;
}

// Scan up to 32 point ranges (lane r < 32 holds (rs, len)) into the replicated best-K list.
template <int K, bool DEDUPE>
__device__ __forceinline__ void scan_ranges(const QueryCtx& c, const float4* __restrict__ sorted, int rs, int len, int lane, u64 (&best)[K]) {
  int pin = len;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    int t = __shfl_up(pin, o, 64);
    if (lane >= o) pin += t;
  }
  const int T = __shfl(pin, 31, 64);
  for (int base = 0; base < T; base += 64) {
    const int i = base + lane;
    int r = 0;  // r = #ranges whose inclusive prefix <= i
#pragma unroll
    for (int b = 16; b > 0; b >>= 1) {
      const int t = r + b;
      const int v = __shfl(pin, t - 1, 64);
      if (i >= v) r = t;
    }
    r = r > 31 ? 31 : r;
    const int pv = __shfl(pin, r > 0 ? r - 1 : 0, 64);   // shuffles must run on all lanes
    const int pe = r > 0 ? pv : 0;
    const int sr = __shfl(rs, r, 64);
    u64 key = ~0ull;
    if (i < T) {
      const float4 p = sorted[sr + (i - pe)];
      const float ddx = c.qx - p.x, ddy = c.qy - p.y, ddz = c.qz - p.z;
      float d = __fmul_rn(ddx, ddx);
      d = __fadd_rn(d, __fmul_rn(ddy, ddy));
      d = __fadd_rn(d, __fmul_rn(ddz, ddz));
      key = ((u64)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
      if (DEDUPE) {   // phase 2 revisits the points phase 1 already selected: keys are unique per point
        bool dup = false;
#pragma unroll
        for (int i = 0; i < K; ++i) dup |= (key == best[i]);
        if (dup) key = ~0ull;
      }
    }
    while (true) {   // wave-level selection
      const bool cont = key < best[K - 1];
      if (__ballot(cont) == 0ull) break;
      const u64 mk = wave_min_u64(cont ? key : ~0ull);
      insert_sorted<K>(best, mk);
      if (key == mk) key = ~0ull;
    }
  }
}

constexpr int FRONT_CAP = 256;   // frontier entries per wave (nodes kept as leaves beyond that)
constexpr int LEAF_CAP = 128;
constexpr int LEAF_COUNT_MAX = 16;  // nodes with <= this many points are scanned instead of expanded

template <int K>
__global__ __launch_bounds__(256) void knn_wave_kernel(const float* __restrict__ q, int N, const NlGridParams* __restrict__ gpp,
                                                       const int* __restrict__ starts, const float4* __restrict__ sorted,
                                                       int Kout, int* __restrict__ idx_out, float* __restrict__ d2_out) {
  __shared__ unsigned s_front[4][2][FRONT_CAP];
  __shared__ int s_leaf_s[4][LEAF_CAP], s_leaf_l[4][LEAF_CAP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = __builtin_amdgcn_readfirstlane(nl_xcd_block() * 4 + wv);
  if (n >= N) return;
  QueryCtx c;
  c.org0 = gpp->origin[0]; c.org1 = gpp->origin[1]; c.org2 = gpp->origin[2];
  c.cell = gpp->cell; c.slack = 1e-3f * c.cell;
  c.qx = q[3 * (size_t)n]; c.qy = q[3 * (size_t)n + 1]; c.qz = q[3 * (size_t)n + 2];

  u64 best[K];
#pragma unroll
  for (int i = 0; i < K; ++i) best[i] = ~0ull;

  // ---------------------------------------------------------------- phase 1: greedy descent -> upper bound U
  unsigned m = 0;
  int L = GRID_BITS;
  while (L > 0) {
    unsigned key = 0xffffffffu;
    if (lane < 8) {
      const unsigned mc = (m << 3) | (unsigned)lane;
      const int sh = 3 * (L - 1);
      const int cnt = starts[(mc + 1) << sh] - starts[mc << sh];
      if (cnt >= K) key = (__float_as_uint(node_mindist2(c, mc, L - 1)) & ~7u) | (unsigned)lane;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o, 64));
    key = (unsigned)__shfl((int)key, 0, 64);
    if (key == 0xffffffffu) break;
    m = (m << 3) | (key & 7u);
    --L;
  }
  {
    const int rs0 = starts[m << (3 * L)], len0 = starts[(m + 1) << (3 * L)] - rs0;
    scan_ranges<K, false>(c, sorted, lane == 0 ? rs0 : 0, lane == 0 ? len0 : 0, lane, best);
  }
  float U = best[K - 1] != ~0ull ? __uint_as_float((unsigned)(best[K - 1] >> 32)) : 3.4e38f;

  // ---------------------------------------------------------------- phase 2: pruned breadth-first descent
  unsigned* front = s_front[wv][0];
  unsigned* nextf = s_front[wv][1];
  int* leaf_s = s_leaf_s[wv];
  int* leaf_l = s_leaf_l[wv];
  int nfront = 1, nleaf = 0;
  if (lane == 0) front[0] = 0u;
  const u64 lt_mask = (1ull << lane) - 1ull;

  auto flush_leaves = [&]() {
    for (int b = 0; b < nleaf; b += 32) {
      const int i = b + lane;
      const bool ok = lane < 32 && i < nleaf;
      scan_ranges<K, true>(c, sorted, ok ? leaf_s[i] : 0, ok ? leaf_l[i] : 0, lane, best);
    }
    nleaf = 0;
    if (best[K - 1] != ~0ull) U = fminf(U, __uint_as_float((unsigned)(best[K - 1] >> 32)));
  };

  for (L = GRID_BITS; L > 0; --L) {
    int nnext = 0;
    const int sh = 3 * (L - 1);
    for (int base = 0; base < nfront; base += 8) {
      const int ni = base + (lane >> 3);
      bool keep = false, leaf = false;
      unsigned mc = 0;
      int rs = 0, cnt = 0;
      if (ni < nfront) {
        mc = (front[ni] << 3) | (unsigned)(lane & 7);
        rs = starts[mc << sh];
        cnt = starts[(mc + 1) << sh] - rs;
        keep = cnt > 0 && node_mindist2(c, mc, L - 1) <= U * 1.000001f;   // '<=' keeps exact ties; margin covers fp32 rounding
        leaf = keep && (L - 1 == 0 || cnt <= LEAF_COUNT_MAX);
      }
      const bool inner = keep && !leaf;
      const u64 mi = __ballot(inner);
      const int ci = __popcll(mi);
      // inner nodes that do not fit the frontier are scanned as (coarse) leaves instead: still exact
      const int room = FRONT_CAP - nnext;
      const int pi = __popcll(mi & lt_mask);
      const bool spill = inner && pi >= room;
      if (inner && !spill) nextf[nnext + pi] = mc;
      nnext += ci < room ? ci : room;
      const bool lf = leaf || spill;
      const u64 ml = __ballot(lf);
      const int cl_ = __popcll(ml);
      if (nleaf + cl_ > LEAF_CAP) flush_leaves();
      if (lf) { const int p = nleaf + __popcll(ml & lt_mask); leaf_s[p] = rs; leaf_l[p] = cnt; }
      nleaf += cl_;
      if (nleaf >= 32) flush_leaves();
    }
    unsigned* t = front; front = nextf; nextf = t;
    nfront = nnext;
    if (nfront == 0) break;
  }
  if (nleaf > 0) flush_leaves();

  if (lane < Kout && lane < K) {
    u64 b = best[0];
#pragma unroll
    for (int k = 1; k < K; ++k) b = lane == k ? best[k] : b;
    const bool ok = b != ~0ull;
    idx_out[(size_t)n * Kout + lane] = ok ? (int)(unsigned)(b & 0xffffffffull) : 0;
    d2_out[(size_t)n * Kout + lane] = ok ? __uint_as_float((unsigned)(b >> 32)) : 0.f;
  }
}

}  // namespace

// ---- host side (called from abi.hip) ------------------------------------------------------------
struct NlKnnGrid {
  NlGridParams* params;  // device
  int* starts;           // device [GRID_CELLS + 1]
  int* counts;           // device [GRID_CELLS]
  int* cursor;           // device [GRID_CELLS]
  int* cell_of;          // device [M]
  float4* sorted;        // device [M]
  int M;
};

size_t nl_knn_grid_bytes(int64_t M) {
  size_t b = 0;
  b += nl_align_up(sizeof(NlGridParams), 256);
  b += nl_align_up(sizeof(int) * (GRID_CELLS + 1), 256);
  b += 2 * nl_align_up(sizeof(int) * GRID_CELLS, 256);
  b += nl_align_up(sizeof(int) * (size_t)(M > 0 ? M : 1), 256);
  b += nl_align_up(sizeof(float4) * (size_t)(M > 0 ? M : 1), 256);
  return b;
}

int nl_knn_grid_build(NlKnnGrid* g, void* mem, const float* xyz, int64_t M, hipStream_t st) {
  char* p = (char*)mem;
  g->params = (NlGridParams*)p; p += nl_align_up(sizeof(NlGridParams), 256);
  g->starts = (int*)p; p += nl_align_up(sizeof(int) * (GRID_CELLS + 1), 256);
  g->counts = (int*)p; p += nl_align_up(sizeof(int) * GRID_CELLS, 256);
  g->cursor = (int*)p; p += nl_align_up(sizeof(int) * GRID_CELLS, 256);
  g->cell_of = (int*)p; p += nl_align_up(sizeof(int) * (size_t)(M > 0 ? M : 1), 256);
  g->sorted = (float4*)p;
  g->M = (int)M;
  if (M <= 0) return NL_OK;
  NL_CHECK_HIP(hipMemsetAsync(g->counts, 0, sizeof(int) * GRID_CELLS, st));
  hipLaunchKernelGGL(knn_bbox_kernel, dim3(1), dim3(256), 0, st, xyz, (int)M, g->params);
  int nb = (int)nl_cdiv(M, 256);
  hipLaunchKernelGGL(knn_count_kernel, dim3(nb), dim3(256), 0, st, xyz, (int)M, g->params, g->counts, g->cell_of);
  hipLaunchKernelGGL(knn_scan_kernel, dim3(1), dim3(1024), 0, st, g->counts, g->starts, g->cursor);
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
  if (K < 1 || K > 8) return NL_ERR_UNSUPPORTED;
  dim3 grid(nl_xcd_grid(nl_cdiv(N, 4)));
  if (K == 1)
    hipLaunchKernelGGL(knn_wave_kernel<1>, grid, dim3(256), 0, st, xyz, (int)N, g->params, g->starts, g->sorted, K, idx, d2);
  else
    hipLaunchKernelGGL(knn_wave_kernel<8>, grid, dim3(256), 0, st, xyz, (int)N, g->params, g->starts, g->sorted, K, idx, d2);
  NL_LAUNCH_CHECK();
  return NL_OK;
}
