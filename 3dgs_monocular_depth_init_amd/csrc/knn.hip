// knn.hip -- F3 / A9: exact K-nearest-neighbour distances for the initial scales.
//
// Replaces `knn(points, 4)` = sklearn NearestNeighbors on the CPU
// (gs_init_compare/utils/runner_utils.py:142-146), called from
// create_splats_with_optimizers (runner.py:88-91) and the scale clamp of
// monocular_depth_init.py:215-223, with 0.3-2 M points.
//
// Points are bucketed into cubic cells of edge h (64-bit cell keys, sorted; the
// caller sorts with torch and passes unique keys + start offsets). One thread
// per query walks the cells ring by ring (Chebyshev radius r = 1, 2, ...) and
// keeps the K smallest squared distances in registers; it stops as soon as the
// K-th distance is <= r*h, which no point outside the visited cube can beat
// (exact result, not approximate). The ring radius is capped (cost grows as
// r^3): queries not proven exact within the cap -- isolated points -- are
// flagged and finished by the brute-force kernel. Cell lookups are binary searches in the
// sorted unique-key array, so memory is O(N) whatever the extent of the cloud.
#include "common.h"

namespace gsr {

constexpr int KNN_MAX_K = 8;

__device__ __forceinline__ uint64_t cell_key(int ix, int iy, int iz) {
  // 21 bits per axis (coordinates are offset to be non-negative by the caller)
  return ((uint64_t)(uint32_t)ix << 42) | ((uint64_t)(uint32_t)iy << 21) | (uint64_t)(uint32_t)iz;
}

__global__ void __launch_bounds__(256)
knn_cell_keys_kernel(int N, const float *__restrict__ pts, const float *__restrict__ origin,
                     float inv_h, int64_t *__restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int ix = (int)floorf((pts[i * 3 + 0] - origin[0]) * inv_h);
  const int iy = (int)floorf((pts[i * 3 + 1] - origin[1]) * inv_h);
  const int iz = (int)floorf((pts[i * 3 + 2] - origin[2]) * inv_h);
  keys[i] = (int64_t)cell_key(max(ix, 0), max(iy, 0), max(iz, 0));
}

template <int K>
__device__ __forceinline__ void topk_insert(float (&best)[K], float d2) {
  if (d2 >= best[K - 1]) return;
  best[K - 1] = d2;
#pragma unroll
  for (int j = K - 1; j > 0; --j) {
    if (best[j] < best[j - 1]) {
      const float t = best[j];
      best[j] = best[j - 1];
      best[j - 1] = t;
    }
  }
}

// N queries (queries[i], result row order[i]) against the points in cell order (sorted_pts).
template <int K>
__global__ void __launch_bounds__(128)
knn_grid_kernel(int N, const float *__restrict__ queries, const float *__restrict__ sorted_pts,
                const int64_t *__restrict__ order,
                const int64_t *__restrict__ ukeys, const int64_t *__restrict__ ustart, int U,
                const float *__restrict__ origin, float h, int max_ring,
                float *__restrict__ out /* [N,K] distances, original order */,
                uint8_t *__restrict__ unresolved /* [N] original order: 1 = not proven exact */) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float px = queries[i * 3], py = queries[i * 3 + 1], pz = queries[i * 3 + 2];
  const float inv_h = 1.0f / h;
  const int cx = max((int)floorf((px - origin[0]) * inv_h), 0);
  const int cy = max((int)floorf((py - origin[1]) * inv_h), 0);
  const int cz = max((int)floorf((pz - origin[2]) * inv_h), 0);
  float best[K];
  bool proven = false;
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = 3.0e38f;
  for (int r = 0; r <= max_ring; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      for (int dy = -r; dy <= r; ++dy) {
        const bool face = (abs(dz) == r) || (abs(dy) == r);
        for (int dx = -r; dx <= r; dx += (face ? 1 : max(2 * r, 1))) {   // shell only
          const int x = cx + dx, y = cy + dy, z = cz + dz;
          if (x < 0 || y < 0 || z < 0) continue;
          const int64_t key = (int64_t)cell_key(x, y, z);
          int lo = 0, hi = U;                 // lower_bound(ukeys, key)
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ukeys[mid] < key) lo = mid + 1; else hi = mid;
          }
          if (lo >= U || ukeys[lo] != key) continue;
          const int64_t s = ustart[lo], e = ustart[lo + 1];
          for (int64_t j = s; j < e; ++j) {
            const float ddx = sorted_pts[j * 3] - px, ddy = sorted_pts[j * 3 + 1] - py,
                        ddz = sorted_pts[j * 3 + 2] - pz;
            topk_insert<K>(best, ddx * ddx + ddy * ddy + ddz * ddz);
          }
        }
      }
    }
    const float reach = (float)r * h;          // everything nearer than this has been seen
    if (r >= 1 && best[K - 1] <= reach * reach) {
      proven = true;
      break;
    }
  }
  const int64_t o = order[i];
  unresolved[o] = proven ? 0 : 1;              // isolated points: finished by the brute-force pass
#pragma unroll
  for (int j = 0; j < K; ++j) out[o * K + j] = sqrtf(best[j]);
}

// Brute force: one workgroup per query, used to size the grid cell from a sample
// and as the small-N path.
template <int K>
__global__ void __launch_bounds__(256)
knn_brute_kernel(int Q, int N, const float *__restrict__ queries, const float *__restrict__ pts,
                 float *__restrict__ out /* [Q,K] */) {
  __shared__ float sbest[256][K];
  const int q = blockIdx.x;
  const float px = queries[q * 3], py = queries[q * 3 + 1], pz = queries[q * 3 + 2];
  float best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = 3.0e38f;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float dx = pts[j * 3] - px, dy = pts[j * 3 + 1] - py, dz = pts[j * 3 + 2] - pz;
    topk_insert<K>(best, dx * dx + dy * dy + dz * dz);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) sbest[threadIdx.x][j] = best[j];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 1; t < 256; ++t)
      for (int j = 0; j < K; ++j) topk_insert<K>(best, sbest[t][j]);
    for (int j = 0; j < K; ++j) out[q * K + j] = sqrtf(best[j]);
  }
}


// ---- K nearest neighbours WITH indices (K <= 64) and the Local Outlier Factor on top ----------
// point_cloud_postprocess/postprocess.py:16-22 runs sklearn's LocalOutlierFactor(n_neighbors=40)
// on the CPU over the whole initial cloud. Same ring search as knn_grid_kernel, but the K best
// (squared distance in fp64 -- sklearn's trees work in float64 --, original index) live in LDS, one
// sorted list per thread laid out [slot][thread] (conflict-free); the query itself is skipped by
// position, as kneighbors() of the fitted data drops it.
constexpr int KNN_IDX_MAX_K = 64;

__global__ void __launch_bounds__(64)
knn_idx_kernel(int Q, int K, const float *__restrict__ queries, const int64_t *__restrict__ self_pos,
               const float *__restrict__ sorted_pts, const int64_t *__restrict__ sorted_ids,
               const int64_t *__restrict__ qorder, const int64_t *__restrict__ ukeys,
               const int64_t *__restrict__ ustart, int U, const float *__restrict__ origin, float h, int max_ring,
               double *__restrict__ out_dist /* [*,K] */, int32_t *__restrict__ out_idx,
               uint8_t *__restrict__ unresolved) {
  __shared__ double sD[KNN_IDX_MAX_K][64];
  __shared__ int32_t sI[KNN_IDX_MAX_K][64];
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 64 + tid;
  if (i >= Q) return;
  const float px = queries[i * 3], py = queries[i * 3 + 1], pz = queries[i * 3 + 2];
  const int64_t self = self_pos ? self_pos[i] : -1;
  const float inv_h = 1.0f / h;
  const int cx = max((int)floorf((px - origin[0]) * inv_h), 0);
  const int cy = max((int)floorf((py - origin[1]) * inv_h), 0);
  const int cz = max((int)floorf((pz - origin[2]) * inv_h), 0);
  int cnt = 0;
  double worst = 1.0e300;
  bool proven = false;
  for (int r = 0; r <= max_ring; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      for (int dy = -r; dy <= r; ++dy) {
        const bool face = (abs(dz) == r) || (abs(dy) == r);
        for (int dx = -r; dx <= r; dx += (face ? 1 : max(2 * r, 1))) {   // shell only
          const int x = cx + dx, y = cy + dy, z = cz + dz;
          if (x < 0 || y < 0 || z < 0) continue;
          const int64_t key = (int64_t)cell_key(x, y, z);
          int lo = 0, hi = U;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ukeys[mid] < key) lo = mid + 1; else hi = mid;
          }
          if (lo >= U || ukeys[lo] != key) continue;
          const int64_t s = ustart[lo], e = ustart[lo + 1];
          for (int64_t j = s; j < e; ++j) {
            if (j == self) continue;
            const double ddx = (double)sorted_pts[j * 3] - (double)px, ddy = (double)sorted_pts[j * 3 + 1] - (double)py,
                         ddz = (double)sorted_pts[j * 3 + 2] - (double)pz;
            const double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
            if (cnt < K || d2 < worst) {
              int pos = (cnt < K) ? cnt : K - 1;
              while (pos > 0 && sD[pos - 1][tid] > d2) {
                sD[pos][tid] = sD[pos - 1][tid];
                sI[pos][tid] = sI[pos - 1][tid];
                --pos;
              }
              sD[pos][tid] = d2;
              sI[pos][tid] = (int32_t)sorted_ids[j];
              if (cnt < K) ++cnt;
              if (cnt == K) worst = sD[K - 1][tid];
            }
          }
        }
      }
    }
    const double reach = (double)r * (double)h;    // everything nearer than this has been seen
    if (r >= 1 && cnt == K && worst <= reach * reach) {
      proven = true;
      break;
    }
  }
  const int64_t o = qorder ? qorder[i] : i;
  unresolved[o] = proven ? 0 : 1;
  for (int j = 0; j < K; ++j) {
    out_dist[o * K + j] = (j < cnt) ? sqrt(sD[j][tid]) : 1.0e300;
    out_idx[o * K + j] = (j < cnt) ? sI[j][tid] : -1;
  }
}

// numpy's mean over the contiguous axis: 8 running sums, combined pairwise, remainder appended
// (numpy/core/src/umath/loops_utils.h pairwise sum for 8 <= n <= 128) -- kept so that a score at
// the threshold falls on the same side as in sklearn
template <typename F>
__device__ __forceinline__ double numpy_mean(int n, F &&at) {
  if (n < 8) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += at(j);
    return s / (double)n;
  }
  double r[8];
  for (int k = 0; k < 8; ++k) r[k] = at(k);
  int j = 8;
  for (; j + 8 <= n; j += 8)
    for (int k = 0; k < 8; ++k) r[k] += at(j + k);
  double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; j < n; ++j) s += at(j);
  return s / (double)n;
}

// local reachability density: 1 / (mean_j max(d(i, j), kdist(j)) + 1e-10)   (sklearn _lof.py)
__global__ void __launch_bounds__(256)
lof_lrd_kernel(int N, int K, const double *__restrict__ dist, const int32_t *__restrict__ idx,
               double *__restrict__ lrd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double m = numpy_mean(K, [&](int j) {
    const double kd = dist[(int64_t)idx[(int64_t)i * K + j] * K + (K - 1)];
    return fmax(dist[(int64_t)i * K + j], kd);
  });
  lrd[i] = 1.0 / (m + 1e-10);
}

// negative_outlier_factor = -mean_j(lrd(j) / lrd(i)); outlier when it is below `offset` (-1.5)
__global__ void __launch_bounds__(256)
lof_score_kernel(int N, int K, const int32_t *__restrict__ idx, const double *__restrict__ lrd, double offset,
                 double *__restrict__ nof, uint8_t *__restrict__ outlier) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double li = lrd[i];
  const double v = -numpy_mean(K, [&](int j) { return lrd[idx[(int64_t)i * K + j]] / li; });
  if (nof) nof[i] = v;
  outlier[i] = v < offset ? 1 : 0;
}

}  // namespace gsr

#define ST ((hipStream_t)stream)

extern "C" int gsr_knn_cell_keys(int N, const float *pts, const float *origin, float h,
                                 int64_t *keys, void *stream) {
  GSR_REQUIRE(N >= 0 && h > 0.f, "knn_cell_keys: bad arguments");
  if (N == 0) return GSR_OK;
  GSR_REQUIRE(pts && origin && keys, "knn_cell_keys: null pointer");
  hipLaunchKernelGGL(gsr::knn_cell_keys_kernel, dim3(gsr::ceil_div(N, 256)), dim3(256), 0, ST, N,
                     pts, origin, 1.0f / h, keys);
  GSR_CHECK_LAUNCH("knn_cell_keys");
  return GSR_OK;
}

extern "C" int gsr_knn_grid(int N, int K, const float *queries, const float *sorted_pts,
                            const int64_t *order,
                            const int64_t *ukeys, const int64_t *ustart, int U,
                            const float *origin, float h, int max_ring, float *out,
                            uint8_t *unresolved, void *stream) {
  GSR_REQUIRE(N >= 0 && U >= 0 && h > 0.f && max_ring >= 1, "knn_grid: bad arguments");
  GSR_REQUIRE(K == 4 || K == 8, "knn_grid: K=%d (built for 4 and 8)", K);
  if (N == 0) return GSR_OK;
  GSR_REQUIRE(queries && sorted_pts && order && ukeys && ustart && origin && out && unresolved,
              "knn_grid: null pointer");
  GSR_REQUIRE(max_ring <= 8, "knn_grid: max_ring %d > 8 (ring cost grows as r^3)", max_ring);
  dim3 grid(gsr::ceil_div(N, 128));
  if (K == 4)
    hipLaunchKernelGGL(gsr::knn_grid_kernel<4>, grid, dim3(128), 0, ST, N, queries, sorted_pts, order,
                       ukeys,
                       ustart, U, origin, h, max_ring, out, unresolved);
  else
    hipLaunchKernelGGL(gsr::knn_grid_kernel<8>, grid, dim3(128), 0, ST, N, queries, sorted_pts, order,
                       ukeys,
                       ustart, U, origin, h, max_ring, out, unresolved);
  GSR_CHECK_LAUNCH("knn_grid");
  return GSR_OK;
}

extern "C" int gsr_knn_brute(int Q, int N, int K, const float *queries, const float *pts,
                             float *out, void *stream) {
  GSR_REQUIRE(Q >= 0 && N >= 0, "knn_brute: bad sizes");
  GSR_REQUIRE(K == 4 || K == 8, "knn_brute: K=%d (built for 4 and 8)", K);
  if (Q == 0) return GSR_OK;
  GSR_REQUIRE(queries && pts && out, "knn_brute: null pointer");
  if (K == 4)
    hipLaunchKernelGGL(gsr::knn_brute_kernel<4>, dim3(Q), dim3(256), 0, ST, Q, N, queries, pts, out);
  else
    hipLaunchKernelGGL(gsr::knn_brute_kernel<8>, dim3(Q), dim3(256), 0, ST, Q, N, queries, pts, out);
  GSR_CHECK_LAUNCH("knn_brute");
  return GSR_OK;
}

extern "C" int gsr_knn_grid_idx(int Q, int K, const float *queries, const int64_t *self_pos,
                                const float *sorted_pts, const int64_t *sorted_ids, const int64_t *qorder,
                                const int64_t *ukeys, const int64_t *ustart, int U, const float *origin,
                                float h, int max_ring, double *out_dist, int32_t *out_idx,
                                uint8_t *unresolved, void *stream) {
  GSR_REQUIRE(Q >= 0 && U >= 0 && h > 0.f && max_ring >= 1 && max_ring <= 8, "knn_grid_idx: bad arguments");
  GSR_REQUIRE(K >= 1 && K <= gsr::KNN_IDX_MAX_K, "knn_grid_idx: K=%d (1..%d)", K, gsr::KNN_IDX_MAX_K);
  if (Q == 0) return GSR_OK;
  GSR_REQUIRE(queries && sorted_pts && sorted_ids && ukeys && ustart && origin && out_dist && out_idx && unresolved,
              "knn_grid_idx: null pointer");
  hipLaunchKernelGGL(gsr::knn_idx_kernel, dim3(gsr::ceil_div(Q, 64)), dim3(64), 0, ST, Q, K, queries, self_pos,
                     sorted_pts, sorted_ids, qorder, ukeys, ustart, U, origin, h, max_ring, out_dist, out_idx,
                     unresolved);
  GSR_CHECK_LAUNCH("knn_grid_idx");
  return GSR_OK;
}

extern "C" int gsr_lof(int N, int K, const double *dist, const int32_t *idx, double offset, double *lrd,
                       double *negative_outlier_factor, uint8_t *outlier, void *stream) {
  GSR_REQUIRE(N >= 0 && K >= 1, "lof: bad sizes");
  if (N == 0) return GSR_OK;
  GSR_REQUIRE(dist && idx && lrd && outlier, "lof: null pointer");
  hipLaunchKernelGGL(gsr::lof_lrd_kernel, dim3(gsr::ceil_div(N, 256)), dim3(256), 0, ST, N, K, dist, idx, lrd);
  hipLaunchKernelGGL(gsr::lof_score_kernel, dim3(gsr::ceil_div(N, 256)), dim3(256), 0, ST, N, K, idx, lrd, offset,
                     negative_outlier_factor, outlier);
  GSR_CHECK_LAUNCH("lof");
  return GSR_OK;
}
