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
//            mindist2(q, child box) <= U (boxes widened by a rounding slack); nodes with <= 16 points become leaves
//            and are scanned four at a time (one per 16-lane slot); the K best keys are wave-uniform (SGPRs) and are
//            updated by a selection loop built on DPP wave-min reductions; U tightens as soon as K real candidates
//            are known.
// The kernel is VALU-issue bound (not memory bound): the design goal is few vector instructions per query.
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

// ---- wave-uniform primitives -------------------------------------------------------------------------------------
// Wave-wide unsigned min as 6 DPP-modified v_min_u32 (row_shr 1/2/4/8 scan inside each row of 16, then row_bcast:15 and
// row_bcast:31 carry the row totals up) + one v_readlane: the result is wave-uniform (an SGPR).  The __shfl-based
// butterflies this replaces were ds_bpermute round trips plus address arithmetic.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_umin(unsigned v) {
  const unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, CTRL, ROWMASK, 0xf, false);
  return t < v ? t : v;
}
__device__ __forceinline__ unsigned wave_umin(unsigned v) {
  v = dpp_umin<0x111, 0xf>(v);   // row_shr:1
  v = dpp_umin<0x112, 0xf>(v);   // row_shr:2
  v = dpp_umin<0x114, 0xf>(v);   // row_shr:4
  v = dpp_umin<0x118, 0xf>(v);   // row_shr:8
  v = dpp_umin<0x142, 0xa>(v);   // row_bcast:15 -> rows 1, 3
  v = dpp_umin<0x143, 0xc>(v);   // row_bcast:31 -> rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
  const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
  return t > v ? t : v;
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  v = dpp_umax<0x111, 0xf>(v); v = dpp_umax<0x112, 0xf>(v); v = dpp_umax<0x114, 0xf>(v); v = dpp_umax<0x118, 0xf>(v);
  v = dpp_umax<0x142, 0xa>(v); v = dpp_umax<0x143, 0xc>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// The K best (dist2 bits, idx) keys, ascending, one per LANE (lane k < K holds the k-th best; other lanes hold the "none"
// key).  An insert is then a handful of lane-parallel instructions whatever K is: every lane compares the new key with
// its own entry and with its left neighbour's (one DPP row_shr:1 each) and keeps / takes the neighbour's / takes the new
// key.  The admission threshold (the K-th best) is cached wave-uniformly.
template <int K>
struct BestK {
  unsigned d, i, p;  // this lane's entry; p = position of the point in the sorted array
  unsigned td, ti;   // entry K-1, wave-uniform
  __device__ __forceinline__ void clear() { d = i = td = ti = 0xffffffffu; p = 0; }
  __device__ __forceinline__ bool full() const { return ti != 0xffffffffu; }
  // caller guarantees (nd, ni) < (td, ti); nd, ni wave-uniform
  __device__ __forceinline__ void insert(unsigned nd, unsigned ni, unsigned np) {
    // left neighbour's entry; lane 0 (and every row start) sees (0, 0), which no key sorts before
    const unsigned pd = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0x111, 0xf, 0xf, false);
    const unsigned pi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)i, 0x111, 0xf, 0xf, false);
    const unsigned pp = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p, 0x111, 0xf, 0xf, false);
    const bool lt_me = nd < d || (nd == d && ni < i);
    const bool lt_prev = nd < pd || (nd == pd && ni < pi);
    d = lt_me ? (lt_prev ? pd : nd) : d;
    i = lt_me ? (lt_prev ? pi : ni) : i;
    p = lt_me ? (lt_prev ? pp : np) : p;
    td = (unsigned)__builtin_amdgcn_readlane((int)d, K - 1);
    ti = (unsigned)__builtin_amdgcn_readlane((int)i, K - 1);
  }
  __device__ __forceinline__ unsigned idx_at(int k) const { return (unsigned)__builtin_amdgcn_readlane((int)i, k); }
};

// Merge one candidate per lane (kd = bits(dist2), ki = idx; 0xffffffff/0xffffffff = none) into the best list.
#ifndef KNN_KO
#define KNN_KO 0   // knock-out bits for timing experiments (results are garbage): 1 no selection loop, 2 no leaf candidates, 4 no breadth-first descent
#endif
#ifdef KNN_STATS   // debug build: step counts over one search, printed by nl_knn_search (tools/build_variant.sh stats knn.hip -DKNN_STATS)
__device__ unsigned long long knn_stats[16];
#define KNN_CNT(i, v) do { if (__lane_id() == 0) atomicAdd(&knn_stats[i], (unsigned long long)(v)); } while (0)
#else
#define KNN_CNT(i, v)
#endif
template <int K>
__device__ __forceinline__ void select_into(BestK<K>& best, unsigned kd, unsigned ki, unsigned kp) {
  if (KNN_KO & 1) { best.d ^= kd & 1u; return; }
  while (true) {
    const bool cont = kd < best.td || (kd == best.td && ki < best.ti);
    if (__ballot(cont) == 0ull) break;
    KNN_CNT(2, 1);
    const unsigned dmin = wave_umin(cont ? kd : 0xffffffffu);
    const bool tie = cont && kd == dmin;
    const u64 tm = __ballot(tie);
    unsigned imin;
    int src = __builtin_ctzll(tm);
    if (__popcll(tm) == 1) imin = (unsigned)__builtin_amdgcn_readlane((int)ki, src);
    else {
      imin = wave_umin(tie ? ki : 0xffffffffu);
      src = __builtin_ctzll(__ballot(tie && ki == imin));
    }
    best.insert(dmin, imin, (unsigned)__builtin_amdgcn_readlane((int)kp, src));
    if (tie && ki == imin) { kd = 0xffffffffu; ki = 0xffffffffu; }
  }
}

struct QueryCtx {
  float qx, qy, qz, org0, org1, org2, cell, slack;
};

// squared distance from the query to the (slack-widened) box of the level-L node with integer cell coordinates (cx, cy, cz);
// the search carries node coordinates next to the Morton code, so no code is ever decoded
__device__ __forceinline__ float node_mindist2(const QueryCtx& c, unsigned cx, unsigned cy, unsigned cz, int L) {
  const float cl = c.cell * (float)(1 << L);
  const float x0 = c.org0 + (float)cx * cl, y0 = c.org1 + (float)cy * cl, z0 = c.org2 + (float)cz * cl;
  const float dx = fmaxf(fmaxf(x0 - c.qx, c.qx - (x0 + cl)) - c.slack, 0.f);
  const float dy = fmaxf(fmaxf(y0 - c.qy, c.qy - (y0 + cl)) - c.slack, 0.f);
  const float dz = fmaxf(fmaxf(z0 - c.qz, c.qz - (z0 + cl)) - c.slack, 0.f);
  return dx * dx + dy * dy + dz * dz;
}

// candidate key of sorted[pos] for this lane (valid lanes only), with phase-1 survivors filtered out (DEDUPE: phase 2
// revisits them; indices are unique per point)
template <int K, bool DEDUPE>
__device__ __forceinline__ void candidate(const QueryCtx& c, const float4* __restrict__ sorted, bool valid, int pos, const BestK<K>& best,
                                          unsigned& kd, unsigned& ki, bool dedupe_now = true, unsigned ubits = 0xffffffffu) {
  kd = 0xffffffffu; ki = 0xffffffffu;
  if (valid) {
    const float4 p = sorted[pos];
    const float ddx = c.qx - p.x, ddy = c.qy - p.y, ddz = c.qz - p.z;
    float d = __fmul_rn(ddx, ddx);
    d = __fadd_rn(d, __fmul_rn(ddy, ddy));
    d = __fadd_rn(d, __fmul_rn(ddz, ddz));
    kd = __float_as_uint(d);
    ki = (unsigned)__float_as_int(p.w);
    // U bounds the K-th distance from above: a point beyond it cannot be among the K nearest ('<=' keeps exact ties)
    if (kd > ubits) { kd = 0xffffffffu; ki = 0xffffffffu; }
    if (DEDUPE && dedupe_now) {   // wave-uniform: only a query whose list was pre-filled by phase 1 revisits points
      bool dup = false;
#pragma unroll
      for (int k = 0; k < K; ++k) dup |= (ki == best.idx_at(k));
      if (dup) { kd = 0xffffffffu; ki = 0xffffffffu; }
    }
  }
}

// one contiguous range [rs, rs+len) (wave-uniform), 64 candidates at a time
// (sink(kd, ki, position): what becomes of the batch's candidates — the selection loop, or the deferred list of knn_wave_kernel)
template <int K, bool DEDUPE, class Sink>
__device__ __forceinline__ void scan_range(const QueryCtx& c, const float4* __restrict__ sorted, int rs, int len, int lane, BestK<K>& best, Sink&& sink,
                                           bool dedupe_now = true, unsigned ubits = 0xffffffffu) {
  for (int base = 0; base < len; base += 64) {
    KNN_CNT(3, 1);
    unsigned kd, ki;
    candidate<K, DEDUPE>(c, sorted, base + lane < len, rs + base + lane, best, kd, ki, dedupe_now, ubits);
    sink(kd, ki, (unsigned)(rs + base + lane));
  }
}

#ifndef KNN_FRONT_CAP
#define KNN_FRONT_CAP 128   // (256 -> 128 in round 5: 15 KB of LDS per workgroup instead of 23.5 = 8 waves per SIMD instead of 6: 0.80 -> 0.70 ms at config 2; a fuller frontier spills into coarse leaves, exact either way)
#endif
constexpr int FRONT_CAP = KNN_FRONT_CAP;   // frontier entries per wave (nodes kept as leaves beyond that)
constexpr int LEAF_CAP = 128;
constexpr int LEAF_COUNT_MAX = 8;   // nodes with <= this many points are scanned instead of expanded; also the slot width

template <int K, int SPW>
__global__ __launch_bounds__(256) void knn_wave_kernel(const float* __restrict__ q, int N, const NlGridParams* __restrict__ gpp,
                                                       const int* __restrict__ starts, const float4* __restrict__ sorted,
                                                       int Kout, int* __restrict__ idx_out, float* __restrict__ d2_out) {
  __shared__ unsigned s_front[4][2][FRONT_CAP], s_fxyz[4][2][FRONT_CAP];   // Morton prefix; cell coordinates x | y << 10 | z << 20
  __shared__ int s_leaf_s[4][LEAF_CAP], s_leaf_l[4][LEAF_CAP];
  __shared__ u64 s_ckey[4][64];        // deferred selection: the keys within the bound, in arrival order ...
  __shared__ unsigned s_cpos[4][64];   // ... and their positions in the sorted array
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n0 = __builtin_amdgcn_readfirstlane((nl_xcd_block() * 4 + wv) * SPW);
  QueryCtx c;
  c.org0 = gpp->origin[0]; c.org1 = gpp->origin[1]; c.org2 = gpp->origin[2];
  c.cell = gpp->cell; c.slack = 1e-3f * c.cell;
  const float inv_cell = gpp->inv_cell;
  bool prev_full = false;   // the previous query of this wave found K neighbours, whose sorted-array positions are
  unsigned ppos = 0;        // in ppos of the lanes with pkeep set (lane k < K after a selection-loop query; the lanes that held them after a deferred one)
  bool pkeep = false;
  u64* const ckey = s_ckey[wv];
  unsigned* const cpos = s_cpos[wv];

  // A wave answers SPW consecutive queries (neighbouring samples of a ray).  Only the first one pays for phase 1: the K
  // neighbours of the previous query, re-measured from this one, give the bound instead.
  for (int sq = 0; sq < SPW; ++sq) {
  const int n = n0 + sq;
  if (n >= N) return;
  c.qx = q[3 * (size_t)n]; c.qy = q[3 * (size_t)n + 1]; c.qz = q[3 * (size_t)n + 2];

  BestK<K> best;
  best.clear();
  float U;
  const bool dedupe_now = !prev_full;   // phase 1 below pre-fills the list; otherwise it starts empty and no point is seen twice
  // DEFERRED SELECTION (round 5).  With the bound of the previous query's neighbours (within a few percent of the K-th distance) only ~10 candidates per query pass
  // `dist2 <= U`, and at least K do (those neighbours themselves).  Feeding them to the selection loop one by one (a wave-min, two ballots and an insert each: ~40 vector
  // instructions x 10.4 per query = a third of the kernel's instructions, KNN_STATS build) is replaced by: append each batch's survivors to a list in LDS (one ballot +
  // prefix count per batch), and at the end give every entry its RANK among them (one 64-bit compare per pair: 2 v_readlane + v_cmp_lt_u64 + v_addc per entry) — the
  // entries of rank < K ARE the answer, already in their output slots.  Same keys, same order: the (dist2 bits, idx) tuples are unique.  U does not tighten during such
  // a query (it is tight already).  A list that would pass 64 entries is poured into the selection loop's list and the query continues there (exact either way).
  bool deferred = prev_full;
  int ncand = 0;
  // The list is written at a computed position by one lane and read back at [lane] by another: a cross-lane hand-over through LDS inside one wave.  LDS operations
  // of a wave execute in order and the compiler must assume the store and the load alias, so this was correct without any fence — implicitly.  The
  // release / acquire pair at wavefront scope + the wave barrier state it; they cost nothing in the ISA (no instruction besides the waits that were there).
  auto lds_list_handover = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto pour = [&]() __attribute__((always_inline)) {   // the deferred list -> the selection loop's list; the query goes on undeferred
    unsigned kd = 0xffffffffu, ki = 0xffffffffu, kp = 0;
    lds_list_handover();
    if (lane < ncand) { const u64 k = ckey[lane]; kd = (unsigned)(k >> 32); ki = (unsigned)k; kp = cpos[lane]; }
    select_into<K>(best, kd, ki, kp);
    ncand = 0;
    deferred = false;
  };
  const u64 lt_mask = (1ull << lane) - 1ull;
  auto sink = [&](unsigned kd, unsigned ki, unsigned kp) __attribute__((always_inline)) {
    if (!deferred) { select_into<K>(best, kd, ki, kp); return; }
    const bool v = ki != 0xffffffffu;
    const u64 mv = __ballot(v);
    if (mv == 0ull) return;
    const int cn = __popcll(mv);
    if (ncand + cn > 64) { pour(); select_into<K>(best, kd, ki, kp); return; }
    if (v) { const int pos = ncand + __popcll(mv & lt_mask); ckey[pos] = ((u64)kd << 32) | (u64)ki; cpos[pos] = kp; }
    ncand += cn;
  };
  if (!prev_full) {
    // -------------------------------------------------------------- phase 1: greedy descent -> upper bound U
    unsigned m = 0, mx = 0, my = 0, mz = 0;   // Morton prefix and cell coordinates of the node being descended (wave-uniform)
    int L = GRID_BITS;
    while (L > 0) {
      unsigned key = 0xffffffffu;
      if (lane < 8) {
        const unsigned mc = (m << 3) | (unsigned)lane;
        const int sh = 3 * (L - 1);
        const int cnt = starts[(mc + 1) << sh] - starts[mc << sh];
        if (cnt >= K)
          key = (__float_as_uint(node_mindist2(c, 2 * mx + (lane & 1), 2 * my + ((lane >> 1) & 1), 2 * mz + ((lane >> 2) & 1), L - 1)) & ~7u) | (unsigned)lane;
      }
      key = wave_umin(key);
      if (key == 0xffffffffu) break;
      const unsigned ch = key & 7u;
      m = (m << 3) | ch;
      mx = 2 * mx + (ch & 1); my = 2 * my + ((ch >> 1) & 1); mz = 2 * mz + ((ch >> 2) & 1);
      --L;
    }
    const int rs0 = starts[m << (3 * L)], len0 = starts[(m + 1) << (3 * L)] - rs0;
    scan_range<K, false>(c, sorted, rs0, len0, lane, best, sink);
    U = best.full() ? __uint_as_float(best.td) : 3.4e38f;
  } else {
    // upper bound from the previous query's neighbours (lane k < K re-measures its k-th one): the largest of their K
    // distances to this query bounds this query's K-th distance, and is usually within a few percent of it
    float dprev = 0.f;
    if (pkeep) {
      const float4 pt = sorted[ppos];
      const float ddx = c.qx - pt.x, ddy = c.qy - pt.y, ddz = c.qz - pt.z;
      dprev = __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz));
    }
    U = __uint_as_float(wave_umax(__float_as_uint(dprev)));
  }

  // ---------------------------------------------------------------- phase 2: pruned breadth-first descent
  unsigned* front = s_front[wv][0];
  unsigned* nextf = s_front[wv][1];
  unsigned* fxyz = s_fxyz[wv][0];
  unsigned* nextx = s_fxyz[wv][1];
  int* leaf_s = s_leaf_s[wv];
  int* leaf_l = s_leaf_l[wv];
  int nfront = 1, nleaf = 0;
  // Start level: with a bound U the search region is the ball's bounding box, which at the smallest level where it spans at most two nodes per axis is covered by
  // <= 8 nodes — the frontier starts there instead of at the root (near a surface: level 1-2 of 6; every skipped level is a 60-instruction batch of this wave).
  // (Wave-uniform arithmetic; the box is widened by the rounding slack and one cell, and every child still passes the exact box test below.)
  int Lstart = GRID_BITS;
#ifndef NL_KNN_NO_START_LEVEL
  if (U < 3.0e38f) {
    const float r = sqrtf(U * 1.000001f) + c.slack;
    const int lo0 = min(max((int)floorf((c.qx - r - c.org0) * inv_cell) - 1, 0), GRID_N - 1), hi0 = min(max((int)floorf((c.qx + r - c.org0) * inv_cell) + 1, 0), GRID_N - 1);
    const int lo1 = min(max((int)floorf((c.qy - r - c.org1) * inv_cell) - 1, 0), GRID_N - 1), hi1 = min(max((int)floorf((c.qy + r - c.org1) * inv_cell) + 1, 0), GRID_N - 1);
    const int lo2 = min(max((int)floorf((c.qz - r - c.org2) * inv_cell) - 1, 0), GRID_N - 1), hi2 = min(max((int)floorf((c.qz + r - c.org2) * inv_cell) + 1, 0), GRID_N - 1);
    int Ls = 1;
    while (Ls < GRID_BITS && (((hi0 >> Ls) - (lo0 >> Ls)) > 1 || ((hi1 >> Ls) - (lo1 >> Ls)) > 1 || ((hi2 >> Ls) - (lo2 >> Ls)) > 1)) ++Ls;
    Ls = __builtin_amdgcn_readfirstlane(Ls);
    if (Ls < GRID_BITS) {
      Lstart = Ls;
      const unsigned nx = (unsigned)(lo0 >> Ls) + (lane & 1), ny = (unsigned)(lo1 >> Ls) + ((lane >> 1) & 1), nz = (unsigned)(lo2 >> Ls) + ((lane >> 2) & 1);
      const bool ok = lane < 8 && nx <= (unsigned)(hi0 >> Ls) && ny <= (unsigned)(hi1 >> Ls) && nz <= (unsigned)(hi2 >> Ls);
      const u64 mk = __ballot(ok);
      if (ok) { const int pos = __popcll(mk & lt_mask); front[pos] = morton3(nx, ny, nz); fxyz[pos] = nx | (ny << 10) | (nz << 20); }
      nfront = __popcll(mk);
    }
  }
#endif
  if (Lstart == GRID_BITS && lane == 0) { front[0] = 0u; fxyz[0] = 0u; }
  KNN_CNT(4, Lstart); KNN_CNT(5, prev_full ? 0 : 1); KNN_CNT(6 + Lstart, 1);

  // leaves hold <= 16 points each: four leaves per 64-lane batch, one per 16-lane slot (no prefix sums, no index search)
  auto flush_leaves = [&]() {
    if (KNN_KO & 2) { nleaf = 0; return; }
    constexpr int SL = LEAF_COUNT_MAX;   // lanes per slot
    for (int b = 0; b < nleaf; b += 64 / SL) {
      KNN_CNT(1, 1);
      const int e = b + lane / SL;
      const bool ok = e < nleaf;
      const int rs = ok ? leaf_s[e] : 0, ln = ok ? leaf_l[e] : 0;
      unsigned kd, ki;
      candidate<K, true>(c, sorted, (lane % SL) < ln, rs + (lane % SL), best, kd, ki, dedupe_now, __float_as_uint(U));
      sink(kd, ki, (unsigned)(rs + (lane % SL)));
    }
    nleaf = 0;
    if (best.full()) U = fminf(U, __uint_as_float(best.td));   // (never full while the query is deferred)
  };

  for (int L = (KNN_KO & 4) ? 0 : Lstart; L > 0; --L) {
    int nnext = 0;
    const int sh = 3 * (L - 1);
    for (int base = 0; base < nfront; base += 8) {
      KNN_CNT(0, 1);
      const int ni = base + (lane >> 3);
      bool keep = false, leaf = false;
      unsigned mc = 0, cxyz = 0;
      int rs = 0, cnt = 0;
      if (ni < nfront) {
        mc = (front[ni] << 3) | (unsigned)(lane & 7);
        const unsigned pxyz = fxyz[ni];
        const unsigned cx = 2 * (pxyz & 1023u) + (lane & 1), cy = 2 * ((pxyz >> 10) & 1023u) + ((lane >> 1) & 1), cz = 2 * (pxyz >> 20) + ((lane >> 2) & 1);
        cxyz = cx | (cy << 10) | (cz << 20);
        rs = starts[mc << sh];
        cnt = starts[(mc + 1) << sh] - rs;
        keep = cnt > 0 && node_mindist2(c, cx, cy, cz, L - 1) <= U * 1.000001f;   // '<=' keeps exact ties; margin covers fp32 rounding
        leaf = keep && (L - 1 == 0 || cnt <= LEAF_COUNT_MAX);
      }
      const bool inner = keep && !leaf;
      const u64 mi = __ballot(inner);
      const int ci = __popcll(mi);
      // inner nodes that do not fit the frontier are scanned as (coarse) leaves instead: still exact
      const int room = FRONT_CAP - nnext;
      const int pi = __popcll(mi & lt_mask);
      const bool spill = inner && pi >= room;
      if (inner && !spill) { nextf[nnext + pi] = mc; nextx[nnext + pi] = cxyz; }
      nnext += ci < room ? ci : room;
      // ranges wider than a slot (dense fine cells, spilled inner nodes) are rare: scanned right away, one by one
      const bool lf = leaf || spill;
      u64 mbig = __ballot(lf && cnt > LEAF_COUNT_MAX);
      while (mbig) {
        const int src = __builtin_ctzll(mbig);
        mbig &= mbig - 1;
        scan_range<K, true>(c, sorted, __builtin_amdgcn_readlane(rs, src), __builtin_amdgcn_readlane(cnt, src), lane, best, sink, dedupe_now, __float_as_uint(U));
        if (best.full()) U = fminf(U, __uint_as_float(best.td));
      }
      const bool small = lf && cnt <= LEAF_COUNT_MAX;
      const u64 ml = __ballot(small);
      const int cl_ = __popcll(ml);
      if (nleaf + cl_ > LEAF_CAP) flush_leaves();
      if (small) { const int p = nleaf + __popcll(ml & lt_mask); leaf_s[p] = rs; leaf_l[p] = cnt; }
      nleaf += cl_;
      if (nleaf >= 32) flush_leaves();
    }
    unsigned* t = front; front = nextf; nextf = t;
    t = fxyz; fxyz = nextx; nextx = t;
    nfront = nnext;
    if (nfront == 0) break;
  }
  if (nleaf > 0) flush_leaves();

  if (deferred && ncand < K) pour();   // (cannot happen: the previous neighbours lie within the bound; kept so that the slots past the list are written below)
  if (deferred) {
    lds_list_handover();
    const u64 mine = lane < ncand ? ckey[lane] : ~0ull;
    const unsigned mlo = (unsigned)mine, mhi = (unsigned)(mine >> 32);
    int rank = 0;
    for (int jj = 0; jj < ncand; ++jj) {   // wave-uniform trip count; entry jj as a scalar pair against every lane's own key
      const u64 kj = ((u64)(unsigned)__builtin_amdgcn_readlane((int)mhi, jj) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane((int)mlo, jj);
      rank += kj < mine ? 1 : 0;
    }
    const bool win = lane < ncand && rank < K;
    if (win && rank < Kout) {
      idx_out[(size_t)n * Kout + rank] = (int)mlo;
      d2_out[(size_t)n * Kout + rank] = __uint_as_float(mhi);
    }
    prev_full = true;
    pkeep = win;
    ppos = cpos[lane];
  } else {
    if (lane < Kout && lane < K) {
      const unsigned bd = best.d, bi = best.i;   // lane k holds the k-th best
      const bool ok = bi != 0xffffffffu;
      idx_out[(size_t)n * Kout + lane] = ok ? (int)bi : 0;
      d2_out[(size_t)n * Kout + lane] = ok ? __uint_as_float(bd) : 0.f;
    }
    prev_full = best.full();
    pkeep = lane < K;
    ppos = best.p;
  }
  }   // queries of this wave
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
  // consecutive queries per wave (each takes its bound from the one before): 16 for render-sized batches (-7 % against 4), 4 where the
  // batch would not fill the chip otherwise (descriptor queries: 1024 points)
#ifndef NL_KNN_BIG_LOG2
#define NL_KNN_BIG_LOG2 18
#endif
#ifndef NL_KNN_SPW_SMALL
#define NL_KNN_SPW_SMALL 4
#endif
  const bool big = N >= (1 << NL_KNN_BIG_LOG2);
  const int spw = big ? 16 : NL_KNN_SPW_SMALL;
  dim3 grid(nl_xcd_grid(nl_cdiv(N, 4 * spw)));
#define NL_KNN(KK, SPW) hipLaunchKernelGGL((knn_wave_kernel<KK, SPW>), grid, dim3(256), 0, st, xyz, (int)N, g->params, g->starts, g->sorted, K, idx, d2)
  if (K == 1) { if (big) NL_KNN(1, 16); else NL_KNN(1, NL_KNN_SPW_SMALL); }
  else { if (big) NL_KNN(8, 16); else NL_KNN(8, NL_KNN_SPW_SMALL); }
#undef NL_KNN
  NL_LAUNCH_CHECK();
#ifdef KNN_STATS
  {
    unsigned long long h[16];
    hipStreamSynchronize(st);
    hipMemcpyFromSymbol(h, HIP_SYMBOL(knn_stats), sizeof(h));
    fprintf(stderr, "knn stats per query (N = %lld): frontier batches %.2f, leaf batches %.2f, selection iterations %.2f, range batches %.2f, start level %.2f, phase-1 %.3f; start-level histogram",
            (long long)N, (double)h[0] / N, (double)h[1] / N, (double)h[2] / N, (double)h[3] / N, (double)h[4] / N, (double)h[5] / N);
    for (int l = 0; l <= GRID_BITS; ++l) fprintf(stderr, " %d:%.3f", l, (double)h[6 + l] / N);
    fprintf(stderr, "\n");
    hipMemset(nullptr, 0, 0);
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(knn_stats), z, sizeof(z));
  }
#endif
  return NL_OK;
}
