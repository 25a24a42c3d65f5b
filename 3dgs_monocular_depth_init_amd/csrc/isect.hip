// isect.hip -- A5: per-tile depth-sorted Gaussian lists.
//
// Replaces gsplat's isect_tiles + global 64-bit radix sort + offset encode
// (reached from gs_init_compare/runner.py:341). MI355X-first design: instead
// of sorting n_isects 64-bit (cam|tile|depth) keys globally (6 radix passes,
// ~150 B of HBM traffic per intersection), bucket the intersections by tile
// with a counting pass (the tile IS the high key digit), then depth-sort every
// bucket inside LDS (160 KB/CU). HBM traffic: 8 B written + 8 B read + 4 B
// written per intersection. The result (per-tile order = ascending depth, ties
// by flat Gaussian index) is identical to the stable global sort.
#include "common.h"

namespace gsr {

__global__ void __launch_bounds__(256)
isect_count_kernel(int C, int N, const float *__restrict__ means2d,
                   const int32_t *__restrict__ radii, int tile_w, int tile_h,
                   int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ tile_counts) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)C * N) return;
  int x0, x1, y0, y1;
  bool vis = tile_rect(means2d, radii, g, tile_w, tile_h, x0, x1, y0, y1);
  if (tiles_per_gauss) tiles_per_gauss[g] = vis ? (x1 - x0) * (y1 - y0) : 0;
  if (!vis) return;
  int c = (int)(g / N);
  int32_t *tc = tile_counts + (int64_t)c * tile_w * tile_h;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) atomicAdd(&tc[y * tile_w + x], 1);
}

__global__ void __launch_bounds__(256)
isect_emit_kernel(int C, int N, const float *__restrict__ means2d,
                  const int32_t *__restrict__ radii, const float *__restrict__ depths, int tile_w,
                  int tile_h, const int32_t *__restrict__ tile_offsets,
                  int32_t *__restrict__ tile_cursor, uint64_t *__restrict__ keys,
                  int64_t capacity) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)C * N) return;
  int x0, x1, y0, y1;
  if (!tile_rect(means2d, radii, g, tile_w, tile_h, x0, x1, y0, y1)) return;
  int c = (int)(g / N);
  int64_t tbase = (int64_t)c * tile_w * tile_h;
  uint64_t key = ((uint64_t)__float_as_uint(depths[g]) << 32) | (uint64_t)(uint32_t)g;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      int64_t t = tbase + y * tile_w + x;
      int64_t pos = (int64_t)tile_offsets[t] + atomicAdd(&tile_cursor[t], 1);
      if (pos < capacity) keys[pos] = key;
    }
}

// Exclusive scan of n int32 counts by ONE workgroup of 1024 threads, 8 items
// per thread per sweep (n is the tile count: 8 160 per 1080p camera).
//
// It also emits tile_order: the tile ids bucketed by list length, longest
// first (64 length classes of 32 entries). The compositing kernels take work
// in this order, so the hardware dispatcher hands the long tiles out first
// and the short ones fill the tail (longest-processing-time-first balance).
constexpr int ORDER_BUCKETS = 64;
__device__ __forceinline__ int order_bucket(int len) {
  return ORDER_BUCKETS - 1 - min(ORDER_BUCKETS - 1, (len + 31) >> 5);   // 0 = longest
}

// CLEAR: the counts are zeroed once they have been read (the bucketed builder reuses the
// buffer as its emit cursor and keeps it across frames: no memset launches in between).
template <bool CLEAR>
__global__ void __launch_bounds__(1024)
scan_kernel(int n, int32_t *__restrict__ in, int32_t *__restrict__ out,
            int32_t *__restrict__ tile_order, int32_t *__restrict__ total_host = nullptr) {
  __shared__ int32_t wave_tot[16];
  __shared__ int32_t carry_s;
  __shared__ int32_t hist[ORDER_BUCKETS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 8192) {
    int idx = base + tid * 8;
    int v[8];
    int local = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = (idx + k < n) ? in[idx + k] : 0;
      local += v[k];
    }
    // inclusive wave scan of `local`
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_prefix = 0;
    for (int w = 0; w < wave; ++w) wave_prefix += wave_tot[w];
    int carry = carry_s;
    int run = carry + wave_prefix + incl - local;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (idx + k < n) out[idx + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + wave_prefix + incl;
    __syncthreads();
  }
  if (tid == 0) {
    out[n] = carry_s;
    // the host's copy of the total, stored straight into pinned host memory (no copy launch
    // between this kernel and the emit kernel); visible once an event recorded after this
    // kernel has completed
    if (total_host) __hip_atomic_store(total_host, carry_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!tile_order) {
    if (CLEAR) {
      __syncthreads();
      for (int t = tid; t < n; t += 1024) in[t] = 0;
    }
    return;
  }
  if (tid < ORDER_BUCKETS) hist[tid] = 0;
  __syncthreads();
  for (int t = tid; t < n; t += 1024) atomicAdd(&hist[order_bucket(in[t])], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 0; b < ORDER_BUCKETS; ++b) {
      int c = hist[b];
      hist[b] = run;
      run += c;
    }
  }
  __syncthreads();
  for (int t = tid; t < n; t += 1024) {
    int pos = atomicAdd(&hist[order_bucket(in[t])], 1);
    tile_order[pos] = t;
  }
  if (CLEAR) {
    __syncthreads();
    for (int t = tid; t < n; t += 1024) in[t] = 0;
  }
}

// ---- per-tile bitonic sort (all-ascending network, virtual +inf padding) ----
template <typename Ptr>
__device__ __forceinline__ void bitonic_sort(Ptr data, int L, int nthreads, int tid) {
  int n_pad = 1;
  while (n_pad < L) n_pad <<= 1;
  for (int k = 2; k <= n_pad; k <<= 1) {
    int half = k >> 1;
    for (int t = tid; t < (n_pad >> 1); t += nthreads) {
      int blk = t / half, off = t - blk * half;
      int lo = blk * k + off, hi = blk * k + k - 1 - off;
      if (hi < L) {
        uint64_t a = data[lo], b = data[hi];
        if (a > b) {
          data[lo] = b;
          data[hi] = a;
        }
      }
    }
    __syncthreads();
    for (int j = k >> 2; j >= 1; j >>= 1) {
      for (int t = tid; t < (n_pad >> 1); t += nthreads) {
        int blk = t / j, off = t - blk * j;
        int lo = 2 * j * blk + off, hi = lo + j;
        if (hi < L) {
          uint64_t a = data[lo], b = data[hi];
          if (a > b) {
            data[lo] = b;
            data[hi] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

constexpr int SORT_SMALL_CAP = 2048;   // 16 KB LDS, 256 threads
constexpr int SORT_LARGE_CAP = 8192;   // 64 KB LDS, 1024 threads

// One 256-thread workgroup per tile; tiles longer than SORT_SMALL_CAP are
// queued for the large kernel.
__global__ void __launch_bounds__(256)
tile_sort_small_kernel(int n_tiles, const int32_t *__restrict__ tile_offsets,
                       const int32_t *__restrict__ tile_order, uint64_t *__restrict__ keys,
                       int32_t *__restrict__ flatten_ids,
                       int32_t *__restrict__ big_list /* [0]=count, then tile ids */) {
  __shared__ uint64_t sk[SORT_SMALL_CAP];
  if ((int)blockIdx.x >= n_tiles) return;
  int tile = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
  int s = tile_offsets[tile], e = tile_offsets[tile + 1];
  int L = e - s;
  if (L <= 0) return;
  if (L > SORT_SMALL_CAP) {
    if (threadIdx.x == 0) {
      int slot = atomicAdd(&big_list[0], 1);
      big_list[1 + slot] = tile;
    }
    return;
  }
  for (int t = threadIdx.x; t < L; t += 256) sk[t] = keys[s + t];
  __syncthreads();
  bitonic_sort(sk, L, 256, threadIdx.x);
  for (int t = threadIdx.x; t < L; t += 256) {
    uint64_t k = sk[t];
    keys[s + t] = k;
    flatten_ids[s + t] = (int32_t)(uint32_t)k;
  }
}

// Persistent grid over the queued long tiles: LDS up to SORT_LARGE_CAP
// entries, in-place global-memory network beyond that (slow, but any list
// length is sorted correctly).
__global__ void __launch_bounds__(1024)
tile_sort_large_kernel(const int32_t *__restrict__ tile_offsets, uint64_t *__restrict__ keys,
                       int32_t *__restrict__ flatten_ids, const int32_t *__restrict__ big_list) {
  __shared__ uint64_t sk[SORT_LARGE_CAP];
  int n_big = big_list[0];
  for (int w = blockIdx.x; w < n_big; w += gridDim.x) {
    int tile = big_list[1 + w];
    int s = tile_offsets[tile], e = tile_offsets[tile + 1];
    int L = e - s;
    if (L <= SORT_LARGE_CAP) {
      for (int t = threadIdx.x; t < L; t += 1024) sk[t] = keys[s + t];
      __syncthreads();
      bitonic_sort(sk, L, 1024, threadIdx.x);
      for (int t = threadIdx.x; t < L; t += 1024) {
        uint64_t k = sk[t];
        keys[s + t] = k;
        flatten_ids[s + t] = (int32_t)(uint32_t)k;
      }
    } else {
      bitonic_sort(keys + s, L, 1024, threadIdx.x);
      for (int t = threadIdx.x; t < L; t += 1024)
        flatten_ids[s + t] = (int32_t)(uint32_t)keys[s + t];
    }
    __syncthreads();
  }
}

}  // namespace gsr

extern "C" int gsr_isect_count(int C, int N, const float *means2d, const int32_t *radii,
                               int tile_w, int tile_h, int32_t *tiles_per_gauss,
                               int32_t *tile_counts, void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && tile_w > 0 && tile_h > 0, "isect_count: bad sizes");
  GSR_REQUIRE(tile_counts, "isect_count: null tile_counts");
  int64_t n_tiles = (int64_t)C * tile_w * tile_h;
  GSR_REQUIRE(n_tiles < (1LL << 30), "isect_count: too many tiles");
  if (n_tiles > 0)
    GSR_CHECK_HIP(hipMemsetAsync(tile_counts, 0, n_tiles * sizeof(int32_t), (hipStream_t)stream));
  int64_t total = (int64_t)C * N;
  if (total == 0) return GSR_OK;
  GSR_REQUIRE(means2d && radii, "isect_count: null pointer");
  GSR_REQUIRE(total < (1LL << 32), "isect_count: C*N must fit 32 bits");
  hipLaunchKernelGGL(gsr::isect_count_kernel, dim3((unsigned)gsr::ceil_div64(total, 256)),
                     dim3(256), 0, (hipStream_t)stream, C, N, means2d, radii, tile_w, tile_h,
                     tiles_per_gauss, tile_counts);
  GSR_CHECK_LAUNCH("isect_count");
  return GSR_OK;
}

extern "C" int gsr_isect_scan(int n_tiles, const int32_t *tile_counts, int32_t *tile_offsets,
                              int32_t *tile_order, void *stream) {
  GSR_REQUIRE(n_tiles >= 0 && tile_counts && tile_offsets, "isect_scan: bad arguments");
  hipLaunchKernelGGL(gsr::scan_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_tiles,
                     const_cast<int32_t *>(tile_counts), tile_offsets, tile_order);
  GSR_CHECK_LAUNCH("isect_scan");
  return GSR_OK;
}

extern "C" int gsr_isect_scan_clear(int n, int32_t *counts, int32_t *offsets, int32_t *order,
                                    int32_t *total_host, void *stream) {
  GSR_REQUIRE(n >= 0 && counts && offsets, "isect_scan_clear: bad arguments");
  hipLaunchKernelGGL(gsr::scan_kernel<true>, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, counts,
                     offsets, order, total_host);
  GSR_CHECK_LAUNCH("isect_scan_clear");
  return GSR_OK;
}

extern "C" int gsr_isect_emit(int C, int N, const float *means2d, const int32_t *radii,
                              const float *depths, int tile_w, int tile_h,
                              const int32_t *tile_offsets, int32_t *tile_cursor,
                              uint64_t *isect_keys, int64_t capacity, void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && tile_w > 0 && tile_h > 0 && capacity >= 0,
              "isect_emit: bad sizes");
  int64_t n_tiles = (int64_t)C * tile_w * tile_h;
  int64_t total = (int64_t)C * N;
  if (total == 0 || n_tiles == 0) return GSR_OK;
  GSR_REQUIRE(means2d && radii && depths && tile_offsets && tile_cursor &&
                  (isect_keys || capacity == 0),
              "isect_emit: null pointer");
  GSR_CHECK_HIP(hipMemsetAsync(tile_cursor, 0, n_tiles * sizeof(int32_t), (hipStream_t)stream));
  hipLaunchKernelGGL(gsr::isect_emit_kernel, dim3((unsigned)gsr::ceil_div64(total, 256)),
                     dim3(256), 0, (hipStream_t)stream, C, N, means2d, radii, depths, tile_w,
                     tile_h, tile_offsets, tile_cursor, isect_keys, capacity);
  GSR_CHECK_LAUNCH("isect_emit");
  return GSR_OK;
}

extern "C" int gsr_tile_sort(int n_tiles, const int32_t *tile_offsets,
                             const int32_t *tile_order, uint64_t *isect_keys,
                             int32_t *flatten_ids, int32_t *big_list, void *stream) {
  GSR_REQUIRE(n_tiles >= 0, "tile_sort: bad n_tiles");
  if (n_tiles == 0) return GSR_OK;
  GSR_REQUIRE(tile_offsets && big_list, "tile_sort: null pointer");
  GSR_CHECK_HIP(hipMemsetAsync(big_list, 0, sizeof(int32_t), (hipStream_t)stream));
  hipLaunchKernelGGL(gsr::tile_sort_small_kernel, dim3(n_tiles), dim3(256), 0,
                     (hipStream_t)stream, n_tiles, tile_offsets, tile_order, isect_keys,
                     flatten_ids, big_list);
  GSR_CHECK_LAUNCH("tile_sort_small");
  hipLaunchKernelGGL(gsr::tile_sort_large_kernel, dim3(256), dim3(1024), 0, (hipStream_t)stream,
                     tile_offsets, isect_keys, flatten_ids, big_list);
  GSR_CHECK_LAUNCH("tile_sort_large");
  return GSR_OK;
}
