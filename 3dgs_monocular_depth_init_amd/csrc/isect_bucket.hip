// isect_bucket.hip -- A5, second generation: tile lists without per-intersection
// global atomics.
//
// isect.hip buckets intersections per tile with one returning global atomic per
// (tile, Gaussian) in the count pass and another in the emit pass: ~5.4 M
// memory-side atomics per 1080p frame at 1 M Gaussians, which run at the chip's
// fixed ~20 G requests/s (0.2 ms) whatever else the kernels do. Here the first
// radix digit is a BUCKET of 8 horizontally adjacent tiles (1 020 per 1080p
// camera): every 1024-thread workgroup histograms its slice of the Gaussians in
// LDS and touches global memory with one atomic per non-empty bucket (~1e5 per
// frame); each bucket is then sorted INSIDE LDS by the composite key
//   [63:61] tile-in-bucket | [60:30] depth bits (sign dropped) | [29:0] g
// which yields, in one pass, the per-tile start offsets and the depth-sorted
// (ties by g) lists -- the same order as isect.hip / a stable global sort.
#include "common.h"

namespace gsr {

constexpr int BK_TILES = 8;              // tiles per bucket (along x)
constexpr int BK_MAX_BUCKETS = 8192;     // LDS: 2 x 4 B x buckets = 64 KB
constexpr int BK_SORT_CAP = 8192;        // entries sorted in LDS per bucket (64 KB)
constexpr int BK_THREADS = 1024;

__device__ __forceinline__ uint64_t bk_key(int tloc, float depth, uint32_t g) {
  return ((uint64_t)tloc << 61) | ((uint64_t)(__float_as_uint(depth) & 0x7fffffffu) << 30) |
         (uint64_t)(g & 0x3fffffffu);
}

// Walk the buckets a Gaussian's tile rectangle touches: f(bucket, first tile x, last tile x + 1, y)
template <typename F>
__device__ __forceinline__ void for_each_bucket(int c, int x0, int x1, int y0, int y1, int bw,
                                                int tile_h, F &&f) {
  for (int y = y0; y < y1; ++y) {
    const int row = (c * tile_h + y) * bw;
    for (int bx = x0 / BK_TILES; bx <= (x1 - 1) / BK_TILES; ++bx)
      f(row + bx, max(x0, bx * BK_TILES), min(x1, (bx + 1) * BK_TILES), y);
  }
}

// Pass 1: entries per bucket. Each workgroup owns a contiguous slice of the pairs.
__global__ void __launch_bounds__(BK_THREADS)
bucket_count_kernel(int C, int N, const float *__restrict__ means2d,
                    const int32_t *__restrict__ radii, int tile_w, int tile_h, int bw,
                    int n_buckets, int64_t chunk, int32_t *__restrict__ bucket_counts) {
  extern __shared__ int32_t hist[];
  for (int b = threadIdx.x; b < n_buckets; b += BK_THREADS) hist[b] = 0;
  __syncthreads();
  const int64_t total = (int64_t)C * N;
  const int64_t g0 = (int64_t)blockIdx.x * chunk, g1 = min(total, g0 + chunk);
  for (int64_t g = g0 + threadIdx.x; g < g1; g += BK_THREADS) {
    int x0, x1, y0, y1;
    if (!tile_rect(means2d, radii, g, tile_w, tile_h, x0, x1, y0, y1)) continue;
    for_each_bucket(C == 1 ? 0 : (int)(g / N),   /* 64-bit division only with several cameras */ x0, x1, y0, y1, bw, tile_h,
                    [&](int b, int xa, int xb, int) { atomicAdd(&hist[b], xb - xa); });
  }
  __syncthreads();
  for (int b = threadIdx.x; b < n_buckets; b += BK_THREADS)
    if (hist[b] > 0) atomicAdd(&bucket_counts[b], hist[b]);
}

// Pass 2: scatter the composite keys into their buckets (unordered inside a bucket).
__global__ void __launch_bounds__(BK_THREADS)
bucket_emit_kernel(int C, int N, const float *__restrict__ means2d,
                   const int32_t *__restrict__ radii, const float *__restrict__ depths,
                   int tile_w, int tile_h, int bw, int n_buckets, int64_t chunk,
                   const int32_t *__restrict__ bucket_offsets, int32_t *__restrict__ bucket_cursor,
                   uint64_t *__restrict__ keys, int64_t capacity) {
  extern __shared__ int32_t lds[];
  int32_t *hist = lds, *base = lds + n_buckets;
  for (int b = threadIdx.x; b < n_buckets; b += BK_THREADS) hist[b] = 0;
  __syncthreads();
  const int64_t total = (int64_t)C * N;
  const int64_t g0 = (int64_t)blockIdx.x * chunk, g1 = min(total, g0 + chunk);
  for (int64_t g = g0 + threadIdx.x; g < g1; g += BK_THREADS) {
    int x0, x1, y0, y1;
    if (!tile_rect(means2d, radii, g, tile_w, tile_h, x0, x1, y0, y1)) continue;
    for_each_bucket(C == 1 ? 0 : (int)(g / N),   /* 64-bit division only with several cameras */ x0, x1, y0, y1, bw, tile_h,
                    [&](int b, int xa, int xb, int) { atomicAdd(&hist[b], xb - xa); });
  }
  __syncthreads();
  // reserve this workgroup's range in every bucket it feeds (one atomic per bucket)
  for (int b = threadIdx.x; b < n_buckets; b += BK_THREADS) {
    const int h = hist[b];
    base[b] = (h > 0) ? bucket_offsets[b] + atomicAdd(&bucket_cursor[b], h) : 0;
    hist[b] = 0;   // becomes the local cursor
  }
  __syncthreads();
  for (int64_t g = g0 + threadIdx.x; g < g1; g += BK_THREADS) {
    int x0, x1, y0, y1;
    if (!tile_rect(means2d, radii, g, tile_w, tile_h, x0, x1, y0, y1)) continue;
    const float d = depths[g];
    for_each_bucket(C == 1 ? 0 : (int)(g / N),   /* 64-bit division only with several cameras */ x0, x1, y0, y1, bw, tile_h, [&](int b, int xa, int xb, int) {
      const int n = xb - xa;
      const int64_t p = (int64_t)base[b] + atomicAdd(&hist[b], n);
      for (int k = 0; k < n; ++k)
        if (p + k < capacity) keys[p + k] = bk_key((xa + k) & (BK_TILES - 1), d, (uint32_t)g);
    });
  }
}

// All-ascending bitonic network with virtual +inf padding (positions >= L). Pair
// index t of a step with stride j touches elements inside [128*(t/64), +128) whenever
// j <= 64, i.e. inside the chunk owned by ONE wave: those steps need no workgroup
// barrier (a wave's LDS accesses complete in order), only the wide strides do.
__device__ __forceinline__ void bk_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename Ptr>
__device__ __forceinline__ void bk_cmpx(Ptr data, int lo, int hi, int L) {
  if (hi < L) {
    const uint64_t a = data[lo], b = data[hi];
    if (a > b) {
      data[lo] = b;
      data[hi] = a;
    }
  }
}
template <bool WAVE_LOCAL_OK, typename Ptr>
__device__ __forceinline__ void bk_bitonic(Ptr data, int L, int tid) {
  int n_pad = 1;
  while (n_pad < L) n_pad <<= 1;
  const int n_pairs = n_pad >> 1;
  for (int lk = 1; (1 << lk) <= n_pad; ++lk) {
    const int k = 1 << lk, lhalf = lk - 1;
    // mirror step: partners span the whole k-block
    for (int t = tid; t < n_pairs; t += BK_THREADS) {
      const int blk = t >> lhalf, off = t & ((1 << lhalf) - 1);
      bk_cmpx(data, (blk << lk) + off, (blk << lk) + k - 1 - off, L);
    }
    if (WAVE_LOCAL_OK && k <= 128) bk_wave_sync(); else __syncthreads();
    for (int lj = lk - 2; lj >= 0; --lj) {
      const int j = 1 << lj;
      for (int t = tid; t < n_pairs; t += BK_THREADS) {
        const int blk = t >> lj, off = t & (j - 1);
        const int lo = (blk << (lj + 1)) + off;
        bk_cmpx(data, lo, lo + j, L);
      }
      // wave-only ordering suffices iff THIS step and the NEXT one (stride j/2, or the next
      // stage's mirror over 2k) both stay inside the wave's own 128-element chunks
      const bool both_local = WAVE_LOCAL_OK && ((j > 1) ? (j <= 64) : (2 * k <= 128));
      if (both_local) bk_wave_sync(); else __syncthreads();
    }
  }
  __syncthreads();
}

// One bitonic network per tile segment, run by a group of GT threads; all groups walk
// the same (k, j) schedule up to n_pad_max so that the workgroup barriers line up.
template <int GT>
__device__ __forceinline__ void bk_bitonic_segments(uint64_t *seg, int L, int n_pad_max, int gtid) {
  int n_pad = 1;
  while (n_pad < L) n_pad <<= 1;
  const int n_pairs = n_pad >> 1;
  // strides are powers of two: block / offset of a pair by shift and mask (a runtime integer
  // division per compare-exchange cost more than the exchange itself)
  for (int lk = 1; (1 << lk) <= n_pad_max; ++lk) {
    const int k = 1 << lk, lhalf = lk - 1;
    if (k <= n_pad)
      for (int t = gtid; t < n_pairs; t += GT) {
        const int blk = t >> lhalf, off = t & ((1 << lhalf) - 1);
        const int base = blk << lk;
        bk_cmpx(seg, base + off, base + k - 1 - off, L);
      }
    if (k <= 128) bk_wave_sync(); else __syncthreads();
    for (int lj = lk - 2; lj >= 0; --lj) {
      const int j = 1 << lj;
      if (k <= n_pad)
        for (int t = gtid; t < n_pairs; t += GT) {
          const int blk = t >> lj, off = t & (j - 1);
          const int lo = (blk << (lj + 1)) + off;
          bk_cmpx(seg, lo, lo + j, L);
        }
      const bool both_local = (j > 1) ? (j <= 64) : (2 * k <= 128);
      if (both_local) bk_wave_sync(); else __syncthreads();
    }
  }
  __syncthreads();
}

// tile order (longest list first, 64 length classes) from finished tile offsets, by ONE
// workgroup of `nthreads` threads. (Doing this in the last-finishing workgroup of the sort
// kernel instead of a launch of its own was tried: the device-scope fences it needs write
// back the XCD's L2 in every workgroup and cost 0.33 ms.)
constexpr int ORD_BUCKETS = 256;   // classes of 8 entries (64 classes of 32: compositing backward +2.5 %)
__device__ __forceinline__ void tile_order_body(int n, const int32_t *tile_offsets,
                                                int32_t *__restrict__ tile_order, int tid,
                                                int nthreads) {
  __shared__ int32_t hist[ORD_BUCKETS];
  auto ld = [&](int i) { return tile_offsets[i]; };
  auto cls = [&](int t) {
    const int len = ld(t + 1) - ld(t);
    return ORD_BUCKETS - 1 - min(ORD_BUCKETS - 1, (len + 7) >> 3);
  };
  if (tid < ORD_BUCKETS) hist[tid] = 0;
  __syncthreads();
  for (int t = tid; t < n; t += nthreads) atomicAdd(&hist[cls(t)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 0; b < ORD_BUCKETS; ++b) {
      const int c = hist[b];
      hist[b] = run;
      run += c;
    }
  }
  __syncthreads();
  for (int t = tid; t < n; t += nthreads) tile_order[atomicAdd(&hist[cls(t)], 1)] = t;
}
__global__ void __launch_bounds__(1024)
tile_order_kernel(int n, const int32_t *__restrict__ tile_offsets, int32_t *__restrict__ tile_order) {
  tile_order_body(n, tile_offsets, tile_order, threadIdx.x, 1024);
}

// Pass 3: one workgroup per bucket. The keys are split by tile-in-bucket while they are
// loaded into LDS (8-bin counting sort), then the 8 tile segments are depth-sorted side
// by side, each by its own 128-thread group; flatten_ids and the tile offsets follow.
__global__ void __launch_bounds__(BK_THREADS)
bucket_sort_kernel(int n_buckets, int tile_w, int bw, const int32_t *__restrict__ bucket_offsets,
                   const int32_t *__restrict__ bucket_order, uint64_t *__restrict__ keys,
                   int32_t *__restrict__ flatten_ids, int32_t *__restrict__ tile_offsets,
                   int n_tiles, int capacity, int32_t *__restrict__ clear_counts) {
  __shared__ uint64_t sk[BK_SORT_CAP];
  __shared__ int seg_cnt[BK_TILES], seg_start[BK_TILES + 1], seg_cur[BK_TILES], npad_max_s;
  const int tid = threadIdx.x;
  const int b = bucket_order ? bucket_order[blockIdx.x] : (int)blockIdx.x;
  // `capacity` = entries the key / id buffers hold. The caller may size them from the
  // previous frame without waiting for this frame's total: everything is clamped so an
  // overflowing frame yields truncated (then discarded) lists, never an out-of-bounds access.
  const int s = min(bucket_offsets[b], capacity), e = min(bucket_offsets[b + 1], capacity);
  const int L = e - s;
  const int row = b / bw, bx = b - row * bw;          // row = cam*tile_h + ty
  if (L <= BK_SORT_CAP) {
    if (tid < BK_TILES) seg_cnt[tid] = 0;
    __syncthreads();
    for (int t = tid; t < L; t += BK_THREADS) atomicAdd(&seg_cnt[(int)(keys[s + t] >> 61)], 1);
    __syncthreads();
    if (tid == 0) {
      int run = 0, mx = 0;
      for (int q = 0; q < BK_TILES; ++q) {
        seg_start[q] = run;
        seg_cur[q] = run;
        run += seg_cnt[q];
        mx = max(mx, seg_cnt[q]);
      }
      seg_start[BK_TILES] = run;
      int np = 1;
      while (np < mx) np <<= 1;
      npad_max_s = np;
    }
    __syncthreads();
    for (int t = tid; t < L; t += BK_THREADS) {
      const uint64_t k = keys[s + t];
      sk[atomicAdd(&seg_cur[(int)(k >> 61)], 1)] = k;
    }
    __syncthreads();
    const int grp = tid >> 7;                          // 8 groups of 128 threads
    bk_bitonic_segments<128>(sk + seg_start[grp], seg_cnt[grp], npad_max_s, tid & 127);
    for (int t = tid; t < L; t += BK_THREADS) {
      const uint64_t k = sk[t];
      keys[s + t] = k;
      flatten_ids[s + t] = (int32_t)(k & 0x3fffffffu);
    }
    if (tid < BK_TILES && bx * BK_TILES + tid < tile_w)
      tile_offsets[row * tile_w + bx * BK_TILES + tid] = s + seg_start[tid];
  } else {   // longer than the LDS sorter: one composite-key network in global memory (slow, exact)
    bk_bitonic<false>(keys + s, L, tid);
    for (int t = tid; t < L; t += BK_THREADS)
      flatten_ids[s + t] = (int32_t)(keys[s + t] & 0x3fffffffu);
    __syncthreads();
    // start offset of each tile = first entry whose tile-in-bucket >= t
    if (tid < BK_TILES && bx * BK_TILES + tid < tile_w) {
      int lo = 0, hi = L;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)(keys[s + mid] >> 61) < tid) lo = mid + 1; else hi = mid;
      }
      tile_offsets[row * tile_w + bx * BK_TILES + tid] = s + lo;
    }
  }
  if (b == n_buckets - 1 && tid == 0) tile_offsets[n_tiles] = e;
  if (clear_counts && tid == 0) clear_counts[b] = 0;   // the emit cursor of this bucket: zero for the next frame
}

// tile_order (longest list first) from finished tile offsets; one workgroup.
static inline int bk_grid(int64_t total, int64_t *chunk) {
  int g = (int)ceil_div64(total, 4096);
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  *chunk = ceil_div64(total, g);
  return g;
}

}  // namespace gsr

extern "C" int gsr_bucket_layout(int C, int tile_w, int tile_h, int *bw_out, int *n_buckets_out) {
  const int bw = gsr::ceil_div(tile_w, gsr::BK_TILES);
  if (bw_out) *bw_out = bw;
  if (n_buckets_out) *n_buckets_out = C * tile_h * bw;
  return (int64_t)C * tile_h * bw <= gsr::BK_MAX_BUCKETS ? GSR_OK : GSR_ECAPACITY;
}

extern "C" int gsr_bucket_count(int C, int N, const float *means2d, const int32_t *radii,
                                int tile_w, int tile_h, int32_t *bucket_counts, int assume_zero,
                                void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && tile_w > 0 && tile_h > 0 && bucket_counts, "bucket_count: bad arguments");
  int bw, nb;
  if (gsr_bucket_layout(C, tile_w, tile_h, &bw, &nb) != GSR_OK) {
    gsr::set_error("bucket_count: %d buckets exceed the LDS histogram (%d)", nb, gsr::BK_MAX_BUCKETS);
    return GSR_ECAPACITY;
  }
  GSR_REQUIRE((int64_t)C * N < (1LL << 30), "bucket_count: C*N must be < 2^30 (composite key)");
  if (nb > 0 && !assume_zero)
    GSR_CHECK_HIP(hipMemsetAsync(bucket_counts, 0, sizeof(int32_t) * nb, (hipStream_t)stream));
  const int64_t total = (int64_t)C * N;
  if (total == 0 || nb == 0) return GSR_OK;
  GSR_REQUIRE(means2d && radii, "bucket_count: null pointer");
  int64_t chunk;
  const int grid = gsr::bk_grid(total, &chunk);
  hipLaunchKernelGGL(gsr::bucket_count_kernel, dim3(grid), dim3(gsr::BK_THREADS),
                     sizeof(int32_t) * nb, (hipStream_t)stream, C, N, means2d, radii, tile_w, tile_h,
                     bw, nb, chunk, bucket_counts);
  GSR_CHECK_LAUNCH("bucket_count");
  return GSR_OK;
}

extern "C" int gsr_bucket_emit(int C, int N, const float *means2d, const int32_t *radii,
                               const float *depths, int tile_w, int tile_h,
                               const int32_t *bucket_offsets, int32_t *bucket_cursor,
                               uint64_t *keys, int64_t capacity, int assume_zero, void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && tile_w > 0 && tile_h > 0 && capacity >= 0, "bucket_emit: bad sizes");
  int bw, nb;
  if (gsr_bucket_layout(C, tile_w, tile_h, &bw, &nb) != GSR_OK) {
    gsr::set_error("bucket_emit: %d buckets exceed the LDS histogram", nb);
    return GSR_ECAPACITY;
  }
  const int64_t total = (int64_t)C * N;
  if (total == 0 || nb == 0) return GSR_OK;
  GSR_REQUIRE(means2d && radii && depths && bucket_offsets && bucket_cursor && (keys || capacity == 0),
              "bucket_emit: null pointer");
  if (!assume_zero)
    GSR_CHECK_HIP(hipMemsetAsync(bucket_cursor, 0, sizeof(int32_t) * nb, (hipStream_t)stream));
  int64_t chunk;
  const int grid = gsr::bk_grid(total, &chunk);
  hipLaunchKernelGGL(gsr::bucket_emit_kernel, dim3(grid), dim3(gsr::BK_THREADS),
                     2 * sizeof(int32_t) * nb, (hipStream_t)stream, C, N, means2d, radii, depths,
                     tile_w, tile_h, bw, nb, chunk, bucket_offsets, bucket_cursor, keys, capacity);
  GSR_CHECK_LAUNCH("bucket_emit");
  return GSR_OK;
}

extern "C" int gsr_bucket_sort(int C, int tile_w, int tile_h, const int32_t *bucket_offsets,
                               const int32_t *bucket_order, uint64_t *keys, int32_t *flatten_ids,
                               int32_t *tile_offsets, int32_t *tile_order, int64_t capacity,
                               int32_t *clear_counts, void *stream) {
  GSR_REQUIRE(C >= 0 && tile_w > 0 && tile_h > 0, "bucket_sort: bad sizes");
  int bw, nb;
  if (gsr_bucket_layout(C, tile_w, tile_h, &bw, &nb) != GSR_OK) {
    gsr::set_error("bucket_sort: %d buckets exceed the LDS histogram", nb);
    return GSR_ECAPACITY;
  }
  if (nb == 0) return GSR_OK;
  GSR_REQUIRE(bucket_offsets && tile_offsets && capacity >= 0, "bucket_sort: bad arguments");
  const int n_tiles = C * tile_w * tile_h;
  hipLaunchKernelGGL(gsr::bucket_sort_kernel, dim3(nb), dim3(gsr::BK_THREADS), 0,
                     (hipStream_t)stream, nb, tile_w, bw, bucket_offsets, bucket_order, keys,
                     flatten_ids, tile_offsets, n_tiles,
                     (int)(capacity < 2147483647LL ? capacity : 2147483647LL), clear_counts);
  GSR_CHECK_LAUNCH("bucket_sort");
  if (tile_order) {
    hipLaunchKernelGGL(gsr::tile_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_tiles,
                       tile_offsets, tile_order);
    GSR_CHECK_LAUNCH("tile_order");
  }
  return GSR_OK;
}
