/* CPU oracle for exact K-nearest-neighbour search.  TEST INFRASTRUCTURE ONLY (see render_oracle.py).
 *
 * Restates /root/reference/nerf_loc/models/ops/knn/src/knn_cpu.cpp:13-64 (KNearestNeighborIdxCpu)
 * followed by the ascending sort of knn_utils.py:60-74, for one cloud (N=1), D=3:
 *   - dist2 is the fp32 chain  d = 0; d += dx*dx; d += dy*dy; d += dz*dz   (knn_cpu.cpp:41-45),
 *     kept free of fma contraction (-ffp-contract=off in the Makefile);
 *   - the reference's max-heap of (dist, idx) tuples with strict-'<' admission while scanning idx
 *     upwards (knn_cpu.cpp:47-52) keeps exactly the K lexicographically smallest (dist2, idx)
 *     tuples and emits them ascending (knn_cpu.cpp:55-61) -- restated here as a sorted insertion
 *     list with the same tuple order;
 *   - output slots k >= P2 stay 0 (torch::full(..., 0), knn_cpu.cpp:23-24).
 * Pinned against the reference's own knn_cpu.cpp compiled in place (oracle/_ref, tests/test_knn_oracle.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KMAX 32

int knn_oracle_f32(const float* q, int64_t n, const float* p, int64_t m, int K,
                   float* out_d2, int64_t* out_idx, int threads) {
  if (K < 1 || K > KMAX) return -1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float bd[KMAX];
    int64_t bi[KMAX];
    int cnt = 0;
    const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
    for (int64_t j = 0; j < m; ++j) {
      float dx = qx - p[3 * j], dy = qy - p[3 * j + 1], dz = qz - p[3 * j + 2];
      float d = 0.f;
      d += dx * dx;
      d += dy * dy;
      d += dz * dz;
      if (cnt < K || d < bd[cnt - 1]) {
        /* j is larger than every stored idx, so among equal dist it sorts last */
        int pos = (cnt < K) ? cnt : K - 1;
        while (pos > 0 && bd[pos - 1] > d) {
          bd[pos] = bd[pos - 1];
          bi[pos] = bi[pos - 1];
          --pos;
        }
        bd[pos] = d;
        bi[pos] = j;
        if (cnt < K) ++cnt;
      }
    }
    for (int k = 0; k < K; ++k) {
      out_d2[i * K + k] = (k < cnt) ? bd[k] : 0.f;
      out_idx[i * K + k] = (k < cnt) ? bi[k] : 0;
    }
  }
  return 0;
}
