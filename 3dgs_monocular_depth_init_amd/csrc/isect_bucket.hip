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
//   [63:61] tile-in-bucket | [60:30] depth bits (sign dropped) | [29:5] g | [4] clamp | [3:0] quadrant mask
// which yields, in one pass, the per-tile start offsets and the depth-sorted
// (ties by g) lists -- the same order as isect.hip / a stable global sort (the mask is a
// function of (tile, g), so it never decides a comparison).
//
// Round 3: the pair's quadrant mask (raster_common.h) is computed HERE, once (in the emit pass,
// under the latency of its scatter), and handed to the compositing kernels in the pair words; with
// `tight` lists a pair whose ellipse (alpha >= 1/255) misses all four quadrants of the tile is not
// listed -- gsplat's bounding-rectangle rule lists it, and 17.5 % of the c4 pairs are of that kind
// (profiles/r03_pair_stats.jsonl). The count pass keeps the rectangle rule (integer arithmetic;
// evaluating the exact test there as well cost 30 us): its counts size the bucket regions, the emit
// pass fills what it does not use with sentinels, and the sort pass compacts. Tight lists render the same image and gradients (the
// dropped pairs contribute alpha < 1/255 everywhere, which both rules discard); the default
// keeps gsplat's lists bit for bit.
#include <type_traits>
#include "raster_common.h"

namespace gsr {

constexpr int BK_TILES = 8;              // tiles per bucket (along x)
constexpr int BK_MAX_BUCKETS = 8192;     // LDS: 2 x 4 B x buckets = 64 KB
constexpr int BK_SORT_CAP = 8192;        // entries sorted in LDS per bucket (64 KB)
constexpr int BK_THREADS = 1024;

constexpr int BK_G_BITS = 25;            // g = camera * N + Gaussian < 2^25 in the composite key
// low 5 bits: [4] clamp flag (opacity > 0.999), [3:0] quadrant mask -- functions of (tile, g), so they
// never decide a comparison
__device__ __forceinline__ uint64_t bk_key(int tloc, float depth, uint32_t g, int mask5) {
  return ((uint64_t)tloc << 61) | ((uint64_t)(__float_as_uint(depth) & 0x7fffffffu) << 30) |
         ((uint64_t)(g & 0x1ffffffu) << 5) | (uint64_t)(mask5 & 31);
}
__device__ __forceinline__ uint32_t bk_key_g(uint64_t k) { return (uint32_t)(k >> 5) & 0x1ffffffu; }
__device__ __forceinline__ uint32_t bk_key_pair(uint64_t k) {
  return bk_key_g(k) | ((uint32_t)(k & 15) << PAIR_MASK_SHIFT) | ((k & 16) ? PAIR_CLAMP_BIT : 0u);
}

// What the exact pair test needs of a projected Gaussian.
typedef PairConic PairGauss;
__device__ __forceinline__ PairGauss load_pair_gauss(const float *__restrict__ means2d,
                                                     const float *__restrict__ conics,
                                                     const float *__restrict__ opacities,
                                                     int opac_per_camera, int64_t g, int N, int C) {
  return make_pair_conic(means2d[g * 2], means2d[g * 2 + 1], conics[g * 3], conics[g * 3 + 1],
                         conics[g * 3 + 2], opacities[(opac_per_camera || C == 1) ? g : (g % N)]);
}
__device__ __forceinline__ int pair_mask_of(const PairGauss &p, int tx, int ty) {
  return pair_quadrant_mask(p, (float)(tx * GSR_TILE), (float)(ty * GSR_TILE));
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

// Pass 1: entries per bucket under gsplat's rule (bounding rectangle x tile grid: integer
// arithmetic only). For tight lists this is an UPPER bound that sizes the bucket regions; the emit
// pass, which evaluates the exact pair test anyway, leaves the surplus slots as sentinels.
// Also clears the two per-frame counters the emit pass uses (nothing reads them before it).
__global__ void __launch_bounds__(BK_THREADS)
bucket_count_kernel(int C, int N, const float *__restrict__ means2d,
                    const int32_t *__restrict__ radii, int tile_w, int tile_h, int bw, int n_buckets,
                    int64_t chunk, int32_t *__restrict__ bucket_counts,
                    int32_t *__restrict__ clear_a, int32_t *__restrict__ clear_b,
                    int32_t *__restrict__ wg_hist) {
  extern __shared__ int32_t hist[];
  for (int b = threadIdx.x; b < n_buckets; b += BK_THREADS) hist[b] = 0;
  if (clear_a)
    for (int b = blockIdx.x * BK_THREADS + threadIdx.x; b < n_buckets; b += gridDim.x * BK_THREADS) {
      clear_a[b] = 0;
      clear_b[b] = 0;
    }
  __syncthreads();
  const int64_t total = (int64_t)C * N;
  const int64_t g0 = (int64_t)blockIdx.x * chunk, g1 = min(total, g0 + chunk);
  constexpr int UB = 4;   // four Gaussians per thread and trip, their loads issued together
  for (int64_t gb = g0 + threadIdx.x; gb < g1; gb += UB * BK_THREADS) {
    float2 m2[UB];
    int2 rd[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t g = min(gb + (int64_t)u * BK_THREADS, g1 - 1);
      m2[u] = *reinterpret_cast<const float2 *>(means2d + g * 2);
      rd[u] = *reinterpret_cast<const int2 *>(radii + g * 2);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t g = gb + (int64_t)u * BK_THREADS;
      int x0, x1, y0, y1;
      if (g >= g1 || !tile_rect_v(m2[u].x, m2[u].y, rd[u].x, rd[u].y, tile_w, tile_h, x0, x1, y0, y1)) continue;
      for_each_bucket(C == 1 ? 0 : (int)(g / N),   /* 64-bit division only with several cameras */ x0, x1, y0, y1, bw, tile_h,
                      [&](int b, int xa, int xb, int) { atomicAdd(&hist[b], xb - xa); });
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < n_buckets; b += BK_THREADS) {
    if (hist[b] > 0) atomicAdd(&bucket_counts[b], hist[b]);
    // this workgroup's own counts, for the emit pass (same grid, same chunks): it reserves its ranges
    // from them instead of walking its Gaussians a second time
    if (wg_hist) wg_hist[(int64_t)blockIdx.x * n_buckets + b] = hist[b];
  }
}

// Exclusive scan of a[0..n) in LDS by the whole workgroup (n <= 8192); returns the total.
// Every thread owns 8 consecutive elements.
__device__ __forceinline__ int bk_block_exclusive_scan(int32_t *a, int n, int32_t *wave_tot /* [17] */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int v[8], local = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = tid * 8 + k;
    v[k] = i < n ? a[i] : 0;
    local += v[k];
  }
  int incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int prefix = 0, tot = 0;
  for (int w = 0; w < BK_THREADS / 64; ++w) {
    const int t = wave_tot[w];
    if (w < wave) prefix += t;
    tot += t;
  }
  int run = prefix + incl - local;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = tid * 8 + k;
    if (i < n) a[i] = run;
    run += v[k];
  }
  __syncthreads();
  return tot;
}

#ifdef GSR_EMIT_TIMELINE
// diagnostic build: s_memtime at the phase boundaries of bucket_emit_kernel, per workgroup
__device__ unsigned long long *g_emit_timeline = nullptr;
#define EMIT_STAMP(k)                                                                     \
  do {                                                                                    \
    __syncthreads();                                                                      \
    if (threadIdx.x == 0 && g_emit_timeline) {                                            \
      unsigned long long t_;                                                              \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      g_emit_timeline[8 * blockIdx.x + (k)] = t_;                                         \
    }                                                                                     \
  } while (0)
#else
#define EMIT_STAMP(k)
#endif
constexpr uint64_t BK_SENTINEL = ~0ull;   // a reserved slot no pair was written to (sorts last, skipped)
constexpr int BK_ORDER_CLASSES = 64;

// Pass 2: scatter the composite keys into their buckets (unordered inside a bucket). Every
// workgroup scans the bucket counts itself (no scan launch in between; workgroup 0 publishes the
// offsets, the bucket work order and the total for the sort pass and the host). The exact pair test
// runs here, once per pair, under the scatter's own latency; tight lists drop the pairs whose mask
// is 0 and fill the slots reserved for them with sentinels. real_counts[b] = pairs actually listed.
template <bool TIGHT>
__global__ void __launch_bounds__(BK_THREADS)
bucket_emit_kernel(int C, int N, const float *__restrict__ means2d,
                   const int32_t *__restrict__ radii, const float *__restrict__ depths,
                   const float *__restrict__ conics, const float *__restrict__ opacities,
                   int opac_per_camera, int tile_w, int tile_h, int bw, int n_buckets, int64_t chunk,
                   const int32_t *__restrict__ bucket_counts, int32_t *__restrict__ bucket_cursor,
                   int32_t *__restrict__ real_counts, int32_t *__restrict__ bucket_offsets,
                   int32_t *__restrict__ bucket_order, int32_t *__restrict__ tile_order,
                   int32_t *__restrict__ total_host, uint64_t *__restrict__ keys, int64_t capacity,
                   const int32_t *__restrict__ wg_hist) {
  extern __shared__ int32_t lds[];
  int32_t *hist = lds, *base = lds + n_buckets, *resv = lds + 2 * n_buckets;
  __shared__ int32_t wave_tot[BK_THREADS / 64 + 1];
  __shared__ int32_t cls[BK_ORDER_CLASSES], clt[BK_ORDER_CLASSES];
  const int tid = threadIdx.x, lane = tid & 63;
  EMIT_STAMP(0);
  // Both passes over the workgroup's Gaussians take them FOUR per thread and trip. With the usual
  // grid that is ONE trip, and its values are loaded right here, ahead of the scan, the histogram
  // and the reservation, whose latencies they then hide under (loaded where they are used they
  // cost 7 us of exposed round trips, profiles/r03_emit_knockouts.log).
  constexpr int UB = 4;
  const int64_t total = (int64_t)C * N;
  const int64_t g0 = (int64_t)blockIdx.x * chunk, g1 = min(total, g0 + chunk);
  const int64_t gw0 = g0 + (tid & ~63);
  const bool one_trip = g0 + (int64_t)UB * BK_THREADS >= g1;
  float2 m2[UB];
  int2 rd[UB];
  float dd[UB], op[UB], cn[UB][3];
  auto load_rects = [&](int64_t gb) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t g = min(gb + (int64_t)u * BK_THREADS, g1 - 1);
      m2[u] = *reinterpret_cast<const float2 *>(means2d + g * 2);
      rd[u] = *reinterpret_cast<const int2 *>(radii + g * 2);
    }
  };
  auto load_rest = [&](int64_t gb) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t g = min(gb + (int64_t)u * BK_THREADS, g1 - 1);
#if defined(GSR_EMIT_KO) && GSR_EMIT_KO == 5
      dd[u] = m2[u].x; cn[u][0] = 0.01f; cn[u][1] = 0.f; cn[u][2] = 0.01f; op[u] = 0.5f;
#else
      dd[u] = depths[g];
      cn[u][0] = conics[g * 3];
      cn[u][1] = conics[g * 3 + 1];
      cn[u][2] = conics[g * 3 + 2];
      op[u] = opacities[(opac_per_camera || C == 1) ? g : (g % N)];
#endif
    }
  };
  const bool owns = blockIdx.x != gridDim.x - 1 && g0 < g1;
  if (owns) {
    load_rects(gw0 + lane);
    if (one_trip) load_rest(gw0 + lane);
  }
  for (int b = tid; b < n_buckets; b += BK_THREADS) {
    hist[b] = 0;
    base[b] = bucket_counts[b];
  }
  __syncthreads();
  // The LAST workgroup of the grid owns no Gaussians: it publishes what the sort pass and the host
  // need (work orders, offsets, slot total) while the others scatter. (On workgroup 0, beside its
  // share of the Gaussians, that job made it the kernel's critical path: +12 k cycles of 93 k.)
  const bool publisher = blockIdx.x == gridDim.x - 1;
  if (publisher && (bucket_order || tile_order)) {
    // work orders, longest first, in 64 length classes of the rectangle-rule counts: of the buckets
    // for the sort pass, and of the TILES for the compositing kernels (bucket by bucket: the 8 tiles
    // of a bucket are neighbours with lists of similar length; a tile-exact order needed a launch
    // of its own after the sort pass, 8 us on one workgroup)
    auto cl = [&](int len) { return BK_ORDER_CLASSES - 1 - min(BK_ORDER_CLASSES - 1, (len + 127) >> 7); };
    auto ntile = [&](int b) { return min(BK_TILES, tile_w - (b % bw) * BK_TILES); };   // tiles of bucket b
    if (tid < BK_ORDER_CLASSES) cls[tid] = clt[tid] = 0;
    __syncthreads();
    for (int b = tid; b < n_buckets; b += BK_THREADS) {
      atomicAdd(&cls[cl(base[b])], 1);
      atomicAdd(&clt[cl(base[b])], ntile(b));
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0, runt = 0;
      for (int k = 0; k < BK_ORDER_CLASSES; ++k) {
        const int c = cls[k], t = clt[k];
        cls[k] = run;
        clt[k] = runt;
        run += c;
        runt += t;
      }
    }
    __syncthreads();
    for (int b = tid; b < n_buckets; b += BK_THREADS) {
      const int k = cl(base[b]);
      if (bucket_order) bucket_order[atomicAdd(&cls[k], 1)] = b;
      if (tile_order) {
        const int nt = ntile(b), at = atomicAdd(&clt[k], nt);
        const int first = (b / bw) * tile_w + (b % bw) * BK_TILES;   // row = cam*tile_h + ty
        for (int j = 0; j < nt; ++j) tile_order[at + j] = first + j;
      }
    }
    __syncthreads();
  }
  const int total_slots = bk_block_exclusive_scan(base, n_buckets, wave_tot);   // base = bucket offsets
  if (publisher) {
    for (int b = tid; b < n_buckets; b += BK_THREADS) bucket_offsets[b] = base[b];
    if (tid == 0) {
      bucket_offsets[n_buckets] = total_slots;
      // the host's copy, stored straight into pinned host memory; visible once an event recorded
      // after this kernel has completed
      if (total_host) __hip_atomic_store(total_host, total_slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  EMIT_STAMP(1);
  if (publisher) return;
  if (wg_hist) {   // the count pass left this workgroup's histogram behind (same grid, same chunks)
    if (g0 < g1)   // (a workgroup without Gaussians -- an empty input launches no count pass -- keeps its zeros)
      for (int b = tid; b < n_buckets; b += BK_THREADS) hist[b] = wg_hist[(int64_t)blockIdx.x * n_buckets + b];
  } else {
    for (int64_t gw = gw0; gw < g1; gw += UB * BK_THREADS) {
      const int64_t gb = gw + lane;
      if (gw != gw0) load_rects(gb);
  #pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int64_t g = gb + (int64_t)u * BK_THREADS;
        int x0, x1, y0, y1;
        if (g >= g1 || !tile_rect_v(m2[u].x, m2[u].y, rd[u].x, rd[u].y, tile_w, tile_h, x0, x1, y0, y1)) continue;
        for_each_bucket(C == 1 ? 0 : (int)(g / N), x0, x1, y0, y1, bw, tile_h,
                        [&](int b, int xa, int xb, int) { atomicAdd(&hist[b], xb - xa); });
      }
    }
  }
  __syncthreads();
  EMIT_STAMP(2);
  // reserve this workgroup's range in every bucket it feeds (one returning atomic per bucket; the
  // workgroups walk the buckets from different starting points, so that at any moment they queue
  // at different words)
  for (int i = tid; i < n_buckets; i += BK_THREADS) {
    int b = i + (int)blockIdx.x * 4;
    b = b % n_buckets;
    const int h = hist[b];
    resv[b] = h;
    base[b] = (h > 0) ? base[b] + atomicAdd(&bucket_cursor[b], h) : 0;
    hist[b] = 0;   // becomes the local cursor
  }
  __syncthreads();
  EMIT_STAMP(3);
  // Pass 2 proper. A Gaussian touches 2.7 tiles on average but a few touch dozens: a lane that
  // walked its own (rows x tiles) nest kept the other 63 waiting for the largest one of the wave
  // (measured: 49 k of the workgroup's 81 k cycles). So the wave POOLS its pairs: an exclusive
  // scan of the lanes' pair counts numbers them, lane l takes pairs l, l+64, ... of the pool and
  // fetches the owning Gaussian's values from the owner's registers (ds_bpermute).
#if defined(GSR_EMIT_KO) && GSR_EMIT_KO == 4
  if (keys == nullptr)
#endif
  for (int64_t gw = gw0; gw < g1; gw += UB * BK_THREADS) {   // wave-uniform trip count
    const int64_t gb = gw + lane;
    if (!one_trip) {
      load_rects(gb);
      load_rest(gb);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t g = gb + (int64_t)u * BK_THREADS;
      int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
      const bool ok = g < g1 && tile_rect_v(m2[u].x, m2[u].y, rd[u].x, rd[u].y, tile_w, tile_h, x0, x1, y0, y1);
      const int nx = ok ? x1 - x0 : 0, n = ok ? nx * (y1 - y0) : 0;
      int incl = n;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      const int excl = incl - n, pool = __shfl(incl, 63, 64);
      const PairConic own = make_pair_conic(m2[u].x, m2[u].y, cn[u][0], cn[u][1], cn[u][2], op[u]);   // once per Gaussian
      const int xy = x0 | (y0 << 16) | (own.clamp ? (int)0x80000000 : 0);   // tile_h < 2^15
      for (int j0 = 0; j0 < pool; j0 += 64) {
        const int j = j0 + lane;
        int L = 0;   // owner = the LAST lane whose first pair is <= j (lanes without pairs share their successor's start)
#pragma unroll
        for (int step = 32; step; step >>= 1) {
          const int e = __shfl(excl, L + step, 64);
          if (e <= j) L += step;
        }
        const int k = j - __shfl(excl, L, 64);
        const int oxy = __shfl(xy, L, 64), onx = __shfl(nx, L, 64);
        PairConic p;
        p.mx = __shfl(own.mx, L, 64);
        p.my = __shfl(own.my, L, 64);
        p.ha = __shfl(own.ha, L, 64);
        p.hc = __shfl(own.hc, L, 64);
        p.sx = __shfl(own.sx, L, 64);
        p.sy = __shfl(own.sy, L, 64);
        p.kx = __shfl(own.kx, L, 64);
        p.ky = __shfl(own.ky, L, 64);
        p.tau_m = __shfl(own.tau_m, L, 64);
        const float odc = __shfl(dd[u], L, 64);
        p.clamp = oxy < 0;
        if (j >= pool) continue;
        int row = (int)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)onx));
        row -= (row * onx > k);
        row += ((row + 1) * onx <= k);
        const int x = (oxy & 0xffff) + (k - row * onx), y = ((oxy >> 16) & 0x7fff) + row;
        const int64_t og = gw + (int64_t)u * BK_THREADS + L;
#if defined(GSR_EMIT_KO) && GSR_EMIT_KO == 3
        const int m = 15;
#else
        const int m = pair_quadrant_mask(p, (float)(x * GSR_TILE), (float)(y * GSR_TILE));
#endif
        if (TIGHT && m == 0) continue;
        const int c = C == 1 ? 0 : (int)(og / N);
        const int b = (c * tile_h + y) * bw + x / BK_TILES;
#if defined(GSR_EMIT_KO) && GSR_EMIT_KO == 2
        const int64_t q = (int64_t)base[b];
#else
        const int64_t q = (int64_t)base[b] + atomicAdd(&hist[b], 1);
#endif
        const uint64_t key = bk_key(x & (BK_TILES - 1), odc, (uint32_t)og, m | (p.clamp ? 16 : 0));
#if defined(GSR_EMIT_KO) && (GSR_EMIT_KO == 1 || GSR_EMIT_KO == 2)
        if (key == 0x1234567ull) keys[q] = key;   // knock-out: never true, keeps the key computation alive
#else
        if (q < capacity) keys[q] = key;
#endif
      }
    }
  }
  __syncthreads();
  EMIT_STAMP(4);
  for (int b = tid; b < n_buckets; b += BK_THREADS) {
    const int h = resv[b], r = hist[b];
    if (h == 0) continue;
    if (TIGHT)
      for (int k = r; k < h; ++k)
        if ((int64_t)base[b] + k < capacity) keys[(int64_t)base[b] + k] = BK_SENTINEL;
    // pairs actually WRITTEN: in a frame that outgrows `capacity` (its lists are discarded and
    // rebuilt) the count must still match what the sort pass finds in the buffer, or the compacted
    // lists would have gaps of uninitialised pair words that the compositing kernels then follow
    const int64_t room = capacity - (int64_t)base[b];
    const int written = (int)(room <= 0 ? 0 : (room < r ? room : r));
    if (written > 0) atomicAdd(&real_counts[b], written);
  }
  EMIT_STAMP(5);
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
  // exclusive prefix over the 256 classes: a scan inside each of the four waves that hold them, then the waves'
  // totals (every thread adding up the classes before its own took up to 255 LDS reads)
  __shared__ int32_t wtot[ORD_BUCKETS / 64];
  int run = 0;
  if (tid < ORD_BUCKETS) {
    const int lane = tid & 63, own = hist[tid];
    int incl = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[tid >> 6] = incl;
    run = incl - own;
  }
  __syncthreads();
  if (tid < ORD_BUCKETS) {
    for (int w = 0; w < (tid >> 6); ++w) run += wtot[w];
    hist[tid] = run;
  }
  __syncthreads();
  for (int t = tid; t < n; t += nthreads) tile_order[atomicAdd(&hist[cls(t)], 1)] = t;
}
__global__ void __launch_bounds__(1024)
tile_order_kernel(int n, const int32_t *__restrict__ tile_offsets, int32_t *__restrict__ tile_order) {
  tile_order_body(n, tile_offsets, tile_order, threadIdx.x, 1024);
}

#ifdef GSR_SORT_TIMELINE
__device__ unsigned long long *g_sort_timeline = nullptr;
#define SORT_STAMP(k)                                                                     \
  do {                                                                                    \
    __syncthreads();                                                                      \
    if (threadIdx.x == 0 && g_sort_timeline) {                                            \
      unsigned long long t_;                                                              \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      g_sort_timeline[8 * blockIdx.x + (k)] = t_;                                         \
    }                                                                                     \
  } while (0)
#else
#define SORT_STAMP(k)
#endif
#ifdef GSR_SORT_COUNT_PATHS
// Diagnostic build (tools/build_variants.sh sortpaths "-DGSR_SORT_COUNT_PATHS=1"; tools/sort_paths.py): buckets by the
// path their sort took: [0] all, [1] equalised bins on keys parked in LDS, [2] equalised bins on streamed keys,
// [3] more than one group, [4] networks instead of ranks, [5] global-memory network.
static __device__ unsigned long long g_sort_paths[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SORT_PATH(i)                                                   \
  do {                                                                 \
    if (threadIdx.x == 0) atomicAdd(&g_sort_paths[i], 1ull);           \
  } while (0)
#else
#define SORT_PATH(i)
#endif
// Pass 3: one workgroup per bucket. The bucket's region holds its listed pairs (and, for tight
// lists, sentinels in the slots the rectangle rule reserved for dropped pairs); the lists are written
// COMPACTED: the bucket's output position is the sum of the real counts of the buckets before it.
//
// Round 4: the depth sort is no longer a network. Round 3's kernel split the keys by tile (8 segments) and
// ran a bitonic network on each: 45-55 compare-exchange steps, every one an LDS read -> compare -> LDS
// write -> sync round trip, 76 % of a workgroup's 26.6 us (profiles/r03_sort_timeline.json). Now the
// keys are split ONCE by (tile, depth bin) -- the bin a monotone function of the depth bits over the
// bucket's own [min, max], with about eight keys per bin -- and every key then finds its rank inside
// its bin by comparing with the bin's few members: position = bin start + rank. Bins are ordered by
// (tile, depth) and the in-bin rank uses the whole composite key, so the result is exactly the order of
// the full sort. A bucket in which some bin holds more than BK_BIN_MAX keys (many equal depths) takes the
// round-3 networks on the tile segments instead (which the split has already laid out).
constexpr int BK_MAX_BINS = 960;       // 8 tiles x up to 120 depth bins (four tables of them beside the 64 KB of keys: 2 workgroups per CU)
#ifndef GSR_BIN_TARGET
#define GSR_BIN_TARGET 4
#endif
constexpr int BK_BIN_TARGET = GSR_BIN_TARGET;   // keys per bin aimed at
constexpr int BK_BIN_MAX = 160;        // longer bins: fall back to the networks
constexpr int BK_BIN_EQ = 24;          // a bin this long (uniform depths: <= 15) -> the bins are equalised, see below
constexpr int BK_KPT = BK_SORT_CAP / BK_THREADS;   // keys per thread when a whole bucket is in flight (8)

// (<= 64 VGPRs: two 1024-thread workgroups per CU)
__global__ void __launch_bounds__(BK_THREADS, 8)
bucket_sort_kernel(int n_buckets, int tile_w, int bw, const int32_t *__restrict__ bucket_offsets,
                   const int32_t *__restrict__ bucket_order, const int32_t *__restrict__ real_counts,
                   uint64_t *__restrict__ keys, uint64_t *__restrict__ keys_sorted,
                   int32_t *__restrict__ flatten_ids, int32_t *__restrict__ pair_ids,
                   int32_t *__restrict__ tile_offsets, int n_tiles, int capacity,
                   int32_t *__restrict__ clear_counts, int32_t *__restrict__ total_host,
                   int32_t *__restrict__ done_host, int seq) {
  __shared__ uint64_t sk[BK_SORT_CAP];
  __shared__ int32_t bins[BK_MAX_BINS + 8], cur[BK_MAX_BINS], cdf[BK_MAX_BINS + 8];
  __shared__ uint32_t occ[BK_MAX_BINS];
  __shared__ int32_t wave_tot[BK_THREADS / 64 + 1];
  __shared__ int seg_cnt[BK_TILES], seg_start[BK_TILES + 1], npad_max_s;
  __shared__ int32_t red[BK_THREADS / 64], out_base_s, maxbin_s;
  __shared__ uint32_t dmin_s, dmax_s;
  __shared__ float sumsq_s;
  const int tid = threadIdx.x;
  SORT_STAMP(0);
  const int b = bucket_order ? bucket_order[blockIdx.x] : (int)blockIdx.x;
  // `capacity` = entries the key / id buffers hold. The caller may size them from the
  // previous frame without waiting for this frame's total: everything is clamped so an
  // overflowing frame yields truncated (then discarded) lists, never an out-of-bounds access.
  const int s = min(bucket_offsets[b], capacity), e = min(bucket_offsets[b + 1], capacity);
  const int LA = e - s;                               // slots of the region (real keys + sentinels)
  const int row = b / bw, bx = b - row * bw;          // row = cam*tile_h + ty
  const int Lr = real_counts[b];                      // listed pairs (sizes the bins; the lists use what the keys say)
  // the bucket's keys: in registers when the region fits (one global read), else re-read per pass
  const bool in_regs = LA <= BK_SORT_CAP;
  uint64_t kreg[BK_KPT];
  uint32_t mn = 0xffffffffu, mx = 0u;
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < BK_KPT; ++u) {
      const int t = tid + u * BK_THREADS;
      kreg[u] = t < LA ? keys[s + t] : BK_SENTINEL;
    }
  }
  auto depth_of = [](uint64_t k) { return (uint32_t)(k >> 30) & 0x7fffffffu; };
  auto for_keys = [&](auto &&f) {
    if (in_regs) {
#pragma unroll
      for (int u = 0; u < BK_KPT; ++u)
        if (kreg[u] != BK_SENTINEL) f(kreg[u]);
    } else {   // (eight loads in flight per thread: one at a time, each pass over a 12 000-key region cost 15 round trips)
      for (int base = 0; base < LA; base += BK_SORT_CAP) {
        uint64_t kk[BK_KPT];
#pragma unroll
        for (int u = 0; u < BK_KPT; ++u) {
          int t = base + tid + u * BK_THREADS;
            kk[u] = t < LA ? keys[s + t] : BK_SENTINEL;
        }
#pragma unroll
        for (int u = 0; u < BK_KPT; ++u)
          if (kk[u] != BK_SENTINEL) f(kk[u]);
      }
    }
  };
  // output position of this bucket: real pairs of all buckets before it; and the depth range of its keys
  {
    int acc = 0;
    for (int i = tid; i < b; i += BK_THREADS) acc += real_counts[i];
    acc = wave_sum_i32_dpp(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    if (tid == 0) {
      dmin_s = 0xffffffffu;
      dmax_s = 0u;
      maxbin_s = 0;
      sumsq_s = 0.f;
    }
    for (int i = tid; i < BK_MAX_BINS + 8; i += BK_THREADS) bins[i] = 0;
    if (tid < BK_MAX_BINS) occ[tid] = 0u;
    __syncthreads();
    if (tid < 64) {     // (a wave sum, not one thread adding the sixteen: that block's wide LDS reads were the kernel's
                        // register peak)
      const int t = wave_sum_i32_dpp(tid < BK_THREADS / 64 ? red[tid] : 0);
      if (tid == 0) out_base_s = t;
    }
    for_keys([&](uint64_t k) {
      const uint32_t d = depth_of(k);
      mn = min(mn, d);
      mx = max(mx, d);
    });
    mn = wave_min_u32_dpp(mn);
    mx = wave_max_u32_dpp(mx);
    if ((tid & 63) == 0 && mn <= mx) {
      atomicMin(&dmin_s, mn);
      atomicMax(&dmax_s, mx);
    }
    __syncthreads();
  }
  SORT_STAMP(1);
  const int out_base = out_base_s;
  // depth bins per tile: about BK_BIN_TARGET keys per (tile, bin) if depths were uniform over [min, max]
  int NB = (Lr + BK_TILES * BK_BIN_TARGET - 1) / (BK_TILES * BK_BIN_TARGET);
  NB = max(1, min(NB, BK_MAX_BINS / BK_TILES));
  const float fmin = __uint_as_float(dmin_s), fmax = __uint_as_float(dmax_s);   // depths are positive floats
  const float span = fmax - fmin;
  // (a span too small for NB / span to be finite -- equal or nearly equal depths -- means one bin per tile)
  const float bscale = (span > 1e-30f && span < 3.0e38f) ? (float)NB / span : 0.f;
  // monotone in the depth bits: float subtraction and multiplication by a positive constant round
  // monotonically, truncation is monotone, the clamps keep both edges inside (and send a NaN, which no
  // finite depth produces here, to bin 0 instead of an out-of-range index)
  //
  // Equalised bins. Real scenes put a tile's Gaussians on a few surfaces: depths in clusters, gaps between them, the
  // range set by a floater -- linear bins over [min, max] then hold dozens of keys each and the in-bin ranks go
  // quadratic (c4's Gaussians on two thin shells: 0.028 -> 0.061 ms, tools/sort_depth_clusters.sh). When a level-1 bin
  // exceeds BK_BIN_EQ keys the bins are re-drawn: every level-1 bin records which of its 32 sub-intervals are occupied
  // (`occ`), and a key's position inside its tile is taken as (keys of the tile in earlier level-1 bins) + (its bin's
  // count) x (occupied sub-intervals before its own + its fraction of its own) / (occupied sub-intervals) -- the
  // tile's empirical distribution, resolved to 1 / (32 NB) of the range; the level-2 bin is that position scaled to
  // NB bins. Monotone in the depth bits like level 1: inside a sub-interval, across sub-intervals and across level-1
  // bins the position never decreases (counts are integers below 2^24, every float operation rounds monotonically).
  const int nbins = BK_TILES * NB;
  auto level1 = [&](uint64_t k, float &t) {      // level-1 depth bin of a key, and its coordinate t in bins
    t = (__uint_as_float(depth_of(k)) - fmin) * bscale;
    return t >= 0.f ? (t < (float)NB ? (int)t : NB - 1) : 0;
  };
  auto sub_of = [&](float t, int d1, float &us) {   // which of the 32 sub-intervals of its level-1 bin
    us = fminf(fmaxf(t - (float)d1, 0.f), 1.f) * 32.f;
    return min(31, (int)us);
  };
  auto bin1 = [&](uint64_t k) {
    float t;
    return (int)(k >> 61) * NB + level1(k, t);
  };
  auto bin2 = [&](uint64_t k) {
    float t, us;
    const int q = (int)(k >> 61), d1 = level1(k, t), b1 = q * NB + d1, sub = sub_of(t, d1, us);
    const uint32_t m = occ[b1];
    const float w = ((float)__popc(m & ((1u << sub) - 1u)) + fminf(us - (float)sub, 1.f)) * __builtin_amdgcn_rcpf((float)max(1, __popc(m)));
    const int t0 = cdf[q * NB], c0 = cdf[b1], c1 = cdf[b1 + 1], t1 = cdf[(q + 1) * NB];
    const float pos = (float)(c0 - t0) + fminf(w, 1.f) * (float)(c1 - c0);
    const int d2 = (int)(pos * ((float)NB * __builtin_amdgcn_rcpf((float)max(1, t1 - t0))));   // (a positive constant per tile)
    return q * NB + min(NB - 1, max(0, d2));
  };
  auto stream_keys = [&](auto &&f) {      // the bucket's keys from global memory (L2), eight loads in flight per thread:
                                          // one at a time, each pass over a 12 000-key region cost 15 round trips
    for (int base = 0; base < LA; base += BK_SORT_CAP) {
      uint64_t kk[BK_KPT];
#pragma unroll
      for (int u = 0; u < BK_KPT; ++u) {
        const int t = base + tid + u * BK_THREADS;
        kk[u] = t < LA ? keys[s + t] : BK_SENTINEL;
      }
#pragma unroll
      for (int u = 0; u < BK_KPT; ++u)
        if (kk[u] != BK_SENTINEL) f(kk[u]);
    }
  };
  auto histogram = [&](auto &&bin_fn, auto &&walk) {      // bins = keys per bin; returns the largest
    walk([&](uint64_t k) { atomicAdd(&bins[bin_fn(k)], 1); });
    __syncthreads();
    int m = 0;
    for (int i = tid; i < nbins; i += BK_THREADS) m = max(m, bins[i]);
    m = wave_max_i32_dpp(m);
    if ((tid & 63) == 0 && m > 0) atomicMax(&maxbin_s, m);
    __syncthreads();
    return maxbin_s;
  };
  auto put = [&](int t, uint64_t k) {
    const int64_t o = (int64_t)out_base + t;
    if (o < capacity) {
      flatten_ids[o] = (int32_t)bk_key_g(k);
      pair_ids[o] = (int32_t)bk_key_pair(k);
      if (keys_sorted) keys_sorted[o] = k;
    }
  };
  int L = 0;
  // everything after the histogram: scan, split, ranks (or networks), lists. Two instances: level-1 bins with the keys
  // in registers when they fit, and equalised bins on keys parked in LDS (or streamed, long buckets)
  auto finish = [&](int maxbin, auto &&bin_fn, auto &&resident_keys) {
    L = bk_block_exclusive_scan(bins, nbins, wave_tot);   // bins = start of every (tile, depth bin); L = real pairs
    if (tid == 0) bins[nbins] = L;
    for (int i = tid; i < nbins; i += BK_THREADS) cur[i] = bins[i];
    __syncthreads();
    // Groups of consecutive bins that fit the LDS sorter together: ONE group when the whole bucket does (c4: 2 200
    // pairs per bucket); otherwise group j = the bins that start inside [j C, (j + 1) C) of the bucket's sorted order,
    // fewer than C + (largest bin) keys. A dense scene (2 M Gaussians seeded from depth maps at 1080p: 12 000 pairs per
    // bucket, 2 000 per tile) takes two groups; round 4 sent such buckets to the global-memory network at the bottom:
    // 1.25 ms per frame, now 0.118. A single tile list may be longer than the sorter, too. Only a BIN that does not
    // fit (thousands of equal depths in one tile) leaves no grouping (n_grp 0).
    const bool by_rank = maxbin <= BK_BIN_MAX, whole = L <= BK_SORT_CAP;
    const int C = by_rank ? BK_SORT_CAP - BK_BIN_MAX : BK_SORT_CAP / 2;
    const int n_grp = whole ? 1 : (maxbin <= BK_SORT_CAP - C ? (L + C - 1) / C : 0);
    if (n_grp > 1) SORT_PATH(3);
    if (n_grp > 0 && !by_rank) SORT_PATH(4);
    if (n_grp == 0) SORT_PATH(5);
    auto first_bin_from = [&](int pos) {     // first bin whose start is >= pos (uniform: every thread searches)
      int lo = 0, hi = nbins;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bins[mid] < pos) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    // one group: bins [b0, b1) = the slice [g0, g0 + Lg) of the bucket's sorted order; `walk` hands over the bucket's keys
    auto sort_group = [&](int b0, int b1, auto &&walk) {
      const int g0 = bins[b0], Lg = bins[b1] - g0;
      walk([&](uint64_t k) {
        const int bn = bin_fn(k);
        if (bn >= b0 && bn < b1) sk[atomicAdd(&cur[bn], 1) - g0] = k;
      });
      __syncthreads();
      SORT_STAMP(3);
      if (by_rank) {
        // every key's rank inside its bin gives its final position: written straight to the lists (a wave's 64
        // consecutive keys sit in neighbouring bins, so its stores fall into the same few lines). Ranks first, then
        // a permutation inside LDS and coalesced stores -- round 4's first version -- holds eight keys and positions
        // per thread across a barrier: 0.037 vs 0.036 ms at c4, and kept live beside the group loop it cost the second
        // workgroup per CU (> 64 VGPRs) or spilled (dense scene 0.138 vs 0.118 ms).
        for (int p = tid; p < Lg; p += BK_THREADS) {
          const uint64_t k = sk[p];
          const int bn = bin_fn(k);
          const int s0 = bins[bn] - g0, e0 = bins[bn + 1] - g0;
          int rank = 0;
          for (int j = s0; j < e0; ++j) rank += sk[j] < k ? 1 : 0;
          put(g0 + s0 + rank, k);
        }
        SORT_STAMP(4);
        SORT_STAMP(5);
        return;
      }
      // many equal depths in one bin: compare-exchange networks instead of ranks
      if (whole) {      // the round-3 networks on the eight tile segments the split has laid out
        if (tid < BK_TILES) {
          seg_start[tid] = bins[tid * NB];
          seg_cnt[tid] = bins[(tid + 1) * NB] - bins[tid * NB];
        }
        __syncthreads();
        if (tid == 0) {
          int mxs = 0;
          for (int q = 0; q < BK_TILES; ++q) mxs = max(mxs, seg_cnt[q]);
          int np = 1;
          while (np < mxs) np <<= 1;
          npad_max_s = np;
        }
        __syncthreads();
        const int grp = tid >> 7;                          // 8 groups of 128 threads
        bk_bitonic_segments<128>(sk + seg_start[grp], seg_cnt[grp], npad_max_s, tid & 127);
      } else {          // one network over the group (the tile is the top of the key)
        bk_bitonic<true>(sk, Lg, tid);
      }
      SORT_STAMP(4);
      for (int t = tid; t < Lg; t += BK_THREADS) put(g0 + t, sk[t]);
      SORT_STAMP(5);
    };
    if (in_regs) {       // the whole region fits the sorter: always one group, from the keys the workgroup holds
      sort_group(0, nbins, resident_keys);
    } else {
      for (int g = 0; g < n_grp; ++g) {
        const int b0 = whole ? 0 : first_bin_from(g * C), b1 = (whole || g + 1 == n_grp) ? nbins : first_bin_from((g + 1) * C);
        sort_group(b0, b1, stream_keys);
        if (g + 1 < n_grp) __syncthreads();                // the next group's split overwrites sk
      }
    }
    if (n_grp == 0) {   // one BIN alone exceeds what a group may hold: one composite-key network in global memory (slow,
                        // exact); sentinels are the largest key and end up behind the L real ones
      bk_bitonic<false>(keys + s, LA, tid);
      for (int t = tid; t < L; t += BK_THREADS) put(t, keys[s + t]);
    }
  };
  const int maxbin1 = histogram(bin1, for_keys);
  SORT_STAMP(2);
  // equalise when the ranks would cost (sum n^2 / keys = comparisons per key) well over what balanced bins cost plus
  // what the extra passes do, about 20 comparisons' worth (a dense scene's long buckets, a dozen keys per bin and 20-30
  // comparisons per key, lost 30 % to equalising at every long bin)
  bool equalise = false;
  if (maxbin1 > BK_BIN_EQ && bscale > 0.f) {
    float sq = 0.f;
    for (int i = tid; i < nbins; i += BK_THREADS) sq += (float)bins[i] * (float)bins[i];
    sq = wave_sum(sq);
    if ((tid & 63) == 0) atomicAdd(&sumsq_s, sq);
    __syncthreads();
    // (long buckets: the bins cannot get finer than 120 per tile and every extra pass streams the keys again)
    equalise = sumsq_s / (float)max(1, Lr) > (in_regs ? 1.5f : 4.f) * (float)Lr / (float)nbins + 24.f;
  }
  SORT_PATH(0);
  if (equalise) {
    SORT_PATH(in_regs ? 1 : 2);
    // short buckets park their keys in LDS (the sorter's array is free until the split) and read them from there:
    // with the wider bin function, eight keys in registers across the passes do not fit 64 VGPRs
    if (in_regs) {
#pragma unroll
      for (int u = 0; u < BK_KPT; ++u) sk[tid + u * BK_THREADS] = kreg[u];
    }
    for (int i = tid; i < nbins; i += BK_THREADS) cdf[i] = bins[i];
    __syncthreads();
    const int n1 = bk_block_exclusive_scan(cdf, nbins, wave_tot);
    auto parked_keys = [&](auto &&f) {
#pragma unroll
      for (int u = 0; u < BK_KPT; ++u) {
        const uint64_t k = sk[tid + u * BK_THREADS];
        if (k != BK_SENTINEL) f(k);
      }
    };
    auto parked_keys_for_split = [&](auto &&f) {      // (the split writes into the array the keys are parked in)
      uint64_t kk[BK_KPT];
#pragma unroll
      for (int u = 0; u < BK_KPT; ++u) kk[u] = sk[tid + u * BK_THREADS];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < BK_KPT; ++u)
        if (kk[u] != BK_SENTINEL) f(kk[u]);
    };
    auto mark = [&](uint64_t k) {
      float t, us;
      const int d1 = level1(k, t);
      atomicOr(&occ[(int)(k >> 61) * NB + d1], 1u << sub_of(t, d1, us));
    };
    if (in_regs) parked_keys(mark); else stream_keys(mark);
    for (int i = tid; i < nbins; i += BK_THREADS) bins[i] = 0;
    if (tid == 0) {
      cdf[nbins] = n1;
      maxbin_s = 0;
    }
    __syncthreads();
    const int maxbin2 = in_regs ? histogram(bin2, parked_keys) : histogram(bin2, stream_keys);
    finish(maxbin2, bin2, parked_keys_for_split);
  } else {
    finish(maxbin1, bin1, [&](auto &&f) {      // the keys still in registers (and dead afterwards: kept live across the
                                               // group loop they cost the second workgroup per CU)
#pragma unroll
      for (int u = 0; u < BK_KPT; ++u)
        if (kreg[u] != BK_SENTINEL) f(kreg[u]);
    });
  }
  if (tid < BK_TILES && bx * BK_TILES + tid < tile_w)
    tile_offsets[row * tile_w + bx * BK_TILES + tid] = min(out_base + bins[tid * NB], capacity);
  if (b == n_buckets - 1 && tid == 0) {
    tile_offsets[n_tiles] = min(out_base + L, capacity);
    if (total_host)
      __hip_atomic_store(total_host, out_base + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (done_host) {
      // both totals, then the frame's sequence number LAST with release: a host polling [2] for `seq` reads the
      // totals of THIS frame as soon as this workgroup is through (it need not wait for the launch, and no event
      // packet sits in the queue in front of the compositing forward: ~6 us of idle GPU per frame)
      __hip_atomic_store(done_host + 0, bucket_offsets[n_buckets], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(done_host + 1, out_base + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(done_host + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (clear_counts && tid == 0) clear_counts[b] = 0;   // the count of this bucket: zero for the next frame
}

// tile_order (longest list first) from finished tile offsets; one workgroup.
static inline int bk_grid(int64_t total, int64_t *chunk) {
  int g = (int)ceil_div64(total, 4096);
  if (g > 256) g = 256;   // (512 / 1024 workgroups measured: no difference, profiles/r03_emit_knockouts.log)
  if (g < 1) g = 1;
  *chunk = ceil_div64(total, g);
  return g;
}

}  // namespace gsr

#ifdef GSR_SORT_COUNT_PATHS
extern "C" int gsr_debug_sort_paths(unsigned long long *out8, int reset) {
  GSR_CHECK_HIP(hipDeviceSynchronize());
  GSR_CHECK_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(gsr::g_sort_paths), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_sort_paths), z, sizeof(z)));
  }
  return GSR_OK;
}
#endif
#ifdef GSR_SORT_TIMELINE
extern "C" int gsr_debug_set_sort_timeline(void *buf) {
  unsigned long long *p = (unsigned long long *)buf;
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_sort_timeline), &p, sizeof(p)));
  return GSR_OK;
}
#endif
#ifdef GSR_EMIT_TIMELINE
extern "C" int gsr_debug_set_emit_timeline(void *buf) {
  unsigned long long *p = (unsigned long long *)buf;
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_emit_timeline), &p, sizeof(p)));
  return GSR_OK;
}
#endif

extern "C" int gsr_bucket_layout(int C, int tile_w, int tile_h, int *bw_out, int *n_buckets_out) {
  const int bw = gsr::ceil_div(tile_w, gsr::BK_TILES);
  if (bw_out) *bw_out = bw;
  if (n_buckets_out) *n_buckets_out = C * tile_h * bw;
  return (int64_t)C * tile_h * bw <= gsr::BK_MAX_BUCKETS ? GSR_OK : GSR_ECAPACITY;
}

extern "C" int gsr_bucket_count(int C, int N, const float *means2d, const int32_t *radii,
                                int tile_w, int tile_h, int32_t *bucket_counts, int32_t *clear_a,
                                int32_t *clear_b, int assume_zero, int32_t *wg_hist, void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && tile_w > 0 && tile_h > 0 && bucket_counts, "bucket_count: bad arguments");
  GSR_REQUIRE((clear_a == nullptr) == (clear_b == nullptr), "bucket_count: clear_a and clear_b go together");
  int bw, nb;
  if (gsr_bucket_layout(C, tile_w, tile_h, &bw, &nb) != GSR_OK) {
    gsr::set_error("bucket_count: %d buckets exceed the LDS histogram (%d)", nb, gsr::BK_MAX_BUCKETS);
    return GSR_ECAPACITY;
  }
  GSR_REQUIRE((int64_t)C * N < (1LL << gsr::BK_G_BITS), "bucket_count: C*N must be < 2^25 (composite key)");
  if (nb > 0 && !assume_zero)
    GSR_CHECK_HIP(hipMemsetAsync(bucket_counts, 0, sizeof(int32_t) * nb, (hipStream_t)stream));
  const int64_t total = (int64_t)C * N;
  if (nb == 0) return GSR_OK;
  if (total == 0) {
    if (clear_a) {
      GSR_CHECK_HIP(hipMemsetAsync(clear_a, 0, sizeof(int32_t) * nb, (hipStream_t)stream));
      GSR_CHECK_HIP(hipMemsetAsync(clear_b, 0, sizeof(int32_t) * nb, (hipStream_t)stream));
    }
    return GSR_OK;
  }
  GSR_REQUIRE(means2d && radii, "bucket_count: null pointer");
  int64_t chunk;
  const int grid = gsr::bk_grid(total, &chunk);
  hipLaunchKernelGGL(gsr::bucket_count_kernel, dim3(grid), dim3(gsr::BK_THREADS),
                     sizeof(int32_t) * nb, (hipStream_t)stream, C, N, means2d, radii, tile_w, tile_h,
                     bw, nb, chunk, bucket_counts, clear_a, clear_b, wg_hist);
  GSR_CHECK_LAUNCH("bucket_count");
  return GSR_OK;
}

extern "C" int gsr_bucket_emit(int C, int N, const float *means2d, const int32_t *radii,
                               const float *depths, const float *conics, const float *opacities,
                               int opac_per_camera, int tile_w, int tile_h, int tight,
                               const int32_t *bucket_counts, int32_t *bucket_cursor,
                               int32_t *real_counts, int32_t *bucket_offsets, int32_t *bucket_order,
                               int32_t *tile_order, int32_t *total_host, uint64_t *keys,
                               int64_t capacity, const int32_t *wg_hist, void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && tile_w > 0 && tile_h > 0 && capacity >= 0, "bucket_emit: bad sizes");
  int bw, nb;
  if (gsr_bucket_layout(C, tile_w, tile_h, &bw, &nb) != GSR_OK) {
    gsr::set_error("bucket_emit: %d buckets exceed the LDS histogram", nb);
    return GSR_ECAPACITY;
  }
  GSR_REQUIRE((int64_t)C * N < (1LL << gsr::BK_G_BITS), "bucket_emit: C*N must be < 2^25 (composite key)");
  const int64_t total = (int64_t)C * N;
  if (nb == 0) return GSR_OK;
  GSR_REQUIRE(bucket_counts && bucket_cursor && real_counts && bucket_offsets && (keys || capacity == 0),
              "bucket_emit: null pointer");
  GSR_REQUIRE(total == 0 || (means2d && radii && depths && conics && opacities), "bucket_emit: null pointer");
  int64_t chunk;
  const int grid = gsr::bk_grid(total, &chunk) + 1;   // + the publisher workgroup (see the kernel)
  const size_t lds = 3 * sizeof(int32_t) * nb;
  if (tight)
    hipLaunchKernelGGL(gsr::bucket_emit_kernel<true>, dim3(grid), dim3(gsr::BK_THREADS), lds,
                       (hipStream_t)stream, C, N, means2d, radii, depths, conics, opacities,
                       opac_per_camera, tile_w, tile_h, bw, nb, chunk, bucket_counts, bucket_cursor,
                       real_counts, bucket_offsets, bucket_order, tile_order, total_host, keys, capacity, wg_hist);
  else
    hipLaunchKernelGGL(gsr::bucket_emit_kernel<false>, dim3(grid), dim3(gsr::BK_THREADS), lds,
                       (hipStream_t)stream, C, N, means2d, radii, depths, conics, opacities,
                       opac_per_camera, tile_w, tile_h, bw, nb, chunk, bucket_counts, bucket_cursor,
                       real_counts, bucket_offsets, bucket_order, tile_order, total_host, keys, capacity, wg_hist);
  GSR_CHECK_LAUNCH("bucket_emit");
  return GSR_OK;
}

extern "C" int gsr_bucket_sort(int C, int tile_w, int tile_h, const int32_t *bucket_offsets,
                               const int32_t *bucket_order, const int32_t *real_counts,
                               uint64_t *keys, uint64_t *keys_sorted, int32_t *flatten_ids,
                               int32_t *pair_ids, int32_t *tile_offsets, int32_t *tile_order,
                               int64_t capacity, int32_t *clear_counts, int32_t *total_host,
                               int32_t *done_host, int seq, void *stream) {
  GSR_REQUIRE(C >= 0 && tile_w > 0 && tile_h > 0, "bucket_sort: bad sizes");
  int bw, nb;
  if (gsr_bucket_layout(C, tile_w, tile_h, &bw, &nb) != GSR_OK) {
    gsr::set_error("bucket_sort: %d buckets exceed the LDS histogram", nb);
    return GSR_ECAPACITY;
  }
  if (nb == 0) return GSR_OK;
  GSR_REQUIRE(bucket_offsets && real_counts && tile_offsets && capacity >= 0 &&
                  ((flatten_ids && pair_ids && keys) || capacity == 0),
              "bucket_sort: bad arguments");
  const int n_tiles = C * tile_w * tile_h;
  hipLaunchKernelGGL(gsr::bucket_sort_kernel, dim3(nb), dim3(gsr::BK_THREADS), 0,
                     (hipStream_t)stream, nb, tile_w, bw, bucket_offsets, bucket_order, real_counts, keys,
                     keys_sorted, flatten_ids, pair_ids, tile_offsets, n_tiles,
                     (int)(capacity < 2147483647LL ? capacity : 2147483647LL), clear_counts, total_host,
                     done_host, seq);
  GSR_CHECK_LAUNCH("bucket_sort");
  if (tile_order) {
    hipLaunchKernelGGL(gsr::tile_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_tiles,
                       tile_offsets, tile_order);
    GSR_CHECK_LAUNCH("tile_order");
  }
  return GSR_OK;
}

namespace gsr {
// Pair words for tile lists built elsewhere (isect.hip, a caller's own lists): one workgroup
// per tile evaluates the quadrant masks of its entries.
__global__ void __launch_bounds__(256)
pair_masks_kernel(int C, int N, int tile_w, int tile_h, const int32_t *__restrict__ tile_offsets,
                  const int32_t *__restrict__ flatten_ids, const float *__restrict__ means2d,
                  const float *__restrict__ conics, const float *__restrict__ opacities,
                  int opac_per_camera, int32_t *__restrict__ pair_ids) {
  const int tile = blockIdx.x;
  const int tin = tile % (tile_w * tile_h);
  const int ty = tin / tile_w, tx = tin - ty * tile_w;
  const int s = tile_offsets[tile], e = tile_offsets[tile + 1];
  for (int i = s + (int)threadIdx.x; i < e; i += 256) {
    const uint32_t g = (uint32_t)flatten_ids[i];
    const PairGauss p = load_pair_gauss(means2d, conics, opacities, opac_per_camera, (int64_t)g, N, C);
    pair_ids[i] = (int32_t)(g | ((uint32_t)pair_mask_of(p, tx, ty) << PAIR_MASK_SHIFT) |
                            (p.clamp ? PAIR_CLAMP_BIT : 0u));
  }
}
}  // namespace gsr

extern "C" int gsr_pair_masks(int C, int N, int tile_w, int tile_h, const int32_t *tile_offsets,
                              const int32_t *flatten_ids, const float *means2d, const float *conics,
                              const float *opacities, int opac_per_camera, int32_t *pair_ids,
                              void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && tile_w > 0 && tile_h > 0, "pair_masks: bad sizes");
  GSR_REQUIRE((int64_t)C * N <= (int64_t)gsr::PAIR_ID_MASK, "pair_masks: C*N must be < 2^27 (pair word)");
  const int n_tiles = C * tile_w * tile_h;
  if (n_tiles == 0 || (int64_t)C * N == 0) return GSR_OK;
  GSR_REQUIRE(tile_offsets && flatten_ids && means2d && conics && opacities && pair_ids,
              "pair_masks: null pointer");
  hipLaunchKernelGGL(gsr::pair_masks_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, C, N,
                     tile_w, tile_h, tile_offsets, flatten_ids, means2d, conics, opacities,
                     opac_per_camera, pair_ids);
  GSR_CHECK_LAUNCH("pair_masks");
  return GSR_OK;
}
