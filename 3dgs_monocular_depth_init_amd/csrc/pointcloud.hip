// pointcloud.hip -- F4: the reference's native point-cloud subsampler, MI355X-native.
//
// Reference (C++20 / Eigen / pybind11, off by default, point_cloud_postprocess/config.py:15-17):
//   native_modules/subsampling/src/impl.cpp:70-126   compute_minimal_gaussian_extents
//   native_modules/subsampling/src/impl.cpp:313-426  subsample_pointcloud_impl (explicit-stack
//       top-down splitting of a cube at its spatial median, axes cycling Y, Z, X, ...; a node is
//       merged into its mean when min(aspect(node box), aspect(tight box)) <= max_bbox_aspect_ratio
//       and the tight box's longest side <= min_extent_multiplier * mean extent of its points)
//   binding: native_modules/subsampling/src/pointcloud_subsampling.cpp:22-67
//
// Not a translation of that stack machine: the splits are midpoints of a cube, so the whole tree is
// fixed by each point's bit path (bit l = which half at level l). The path is computed per point
// with the reference's own float arithmetic (split = (min + max) / 2, repeated), the points are
// radix-sorted by path (sort_pairs_u64 below), every node is then a contiguous range, and the tree is walked
// LEVEL-SYNCHRONOUSLY: per level one statistics launch (count, sum of extents / positions / colours
// in fp64, tight box by ordered-integer atomic min/max -- all order-independent) and one decision
// launch over the points that are still unresolved. 63 levels at most; no recursion, no host sync.
#include <hip/hip_runtime.h>

#include <cstring>


#include "common.h"

namespace gsr {
namespace pc {

constexpr int MAX_LEVELS = 63;

// impl.cpp:17-35 + 86-108: extent = 2 * depth / min(fx, fy), minimum over the cameras that see
// the point (depth > 0, 0 <= u < W, 0 <= v < H); -1 when no camera does.
__global__ void __launch_bounds__(256)
min_extents_kernel(int N, int C, const float *__restrict__ points, const float *__restrict__ Ks,
                   const float *__restrict__ Ps, const int32_t *__restrict__ sizes,
                   float *__restrict__ extents) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float x = points[i * 3], y = points[i * 3 + 1], z = points[i * 3 + 2];
  float best = 3.402823466e+38f;
  for (int c = 0; c < C; ++c) {
    const float *P = Ps + c * 12;
    const float px = P[0] * x + P[1] * y + P[2] * z + P[3];
    const float py = P[4] * x + P[5] * y + P[6] * z + P[7];
    const float d = P[8] * x + P[9] * y + P[10] * z + P[11];
    if (d <= 0.f) continue;
    const float u = px / d, v = py / d;
    if (u < 0.f || u >= (float)sizes[c * 2] || v < 0.f || v >= (float)sizes[c * 2 + 1]) continue;
    const float f = fminf(Ks[c * 9], Ks[c * 9 + 4]);
    best = fminf(best, 2.0f * (d / f));
  }
  extents[i] = best == 3.402823466e+38f ? -1.0f : best;
}

// order-preserving float <-> int for atomicMin / atomicMax
__device__ __forceinline__ int f2o(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float o2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void __launch_bounds__(256)
bbox_kernel(int N, const float *__restrict__ points, int *__restrict__ mm /* [6]: min xyz, max xyz */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int o = f2o(points[i * 3 + a]);
    atomicMin(&mm[a], o);
    atomicMax(&mm[3 + a], o);
  }
}

// geometry.h:53-60 cube_from_points: centre +- half of the longest side
__global__ void cube_kernel(const int *__restrict__ mm, float *__restrict__ cube /* [6] */) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mn[3], mx[3], md = -3.402823466e+38f;
  for (int a = 0; a < 3; ++a) {
    mn[a] = o2f(mm[a]);
    mx[a] = o2f(mm[3 + a]);
    md = fmaxf(md, mx[a] - mn[a]);
  }
  const float half = md / 2.0f;
  for (int a = 0; a < 3; ++a) {
    const float c = (mn[a] + mx[a]) / 2.0f;
    cube[a] = c - half;
    cube[3 + a] = c + half;
  }
}

// bit path of a point: level l splits axis (l + 1) % 3 at (min + max) / 2, right half when
// !(pos < split) (impl.cpp:217-228); bit l is stored at position 62 - l
__global__ void __launch_bounds__(256)
codes_kernel(int N, const float *__restrict__ points, const float *__restrict__ cube,
             uint64_t *__restrict__ codes, uint32_t *__restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float mn[3] = {cube[0], cube[1], cube[2]}, mx[3] = {cube[3], cube[4], cube[5]};
  const float p[3] = {points[i * 3], points[i * 3 + 1], points[i * 3 + 2]};
  uint64_t code = 0;
  for (int l = 0; l < MAX_LEVELS; ++l) {
    const int a = (l + 1) % 3;
    const float split = (mn[a] + mx[a]) / 2.0f;
    const bool right = !(p[a] < split);
    if (right) {
      mn[a] = split;
      code |= 1ull << (62 - l);
    } else {
      mx[a] = split;
    }
  }
  codes[i] = code;
  idx[i] = (uint32_t)i;
}

struct Work {       // everything in sorted (path) order, one slot per point
  int N;
  const uint64_t *code;
  const uint32_t *idx;
  float *pos, *rgb, *ext;        // gathered inputs
  float *bmin, *bmax;            // the node box each unresolved point currently sits in
  int *lo, *hi;                  // its node's range
  int *state;                    // 0 unresolved, 1 emit self, 2 emit merged (head only), 3 absorbed
  int *cnt;                      // per-node statistics, stored at the node's first index
  double *s_ext, *s_pos, *s_rgb;
  int *t_min, *t_max;
  float *m_pos, *m_rgb;          // merged outputs (at the head)
  int *emit, *offs;
};

__global__ void __launch_bounds__(256)
gather_kernel(Work w, const float *__restrict__ points, const float *__restrict__ rgbs,
              const float *__restrict__ extents, const float *__restrict__ cube) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w.N) return;
  const uint32_t s = w.idx[i];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    w.pos[i * 3 + a] = points[s * 3 + a];
    w.rgb[i * 3 + a] = rgbs[s * 3 + a];
    w.bmin[i * 3 + a] = cube[a];
    w.bmax[i * 3 + a] = cube[3 + a];
  }
  w.ext[i] = extents[s];
  w.lo[i] = 0;
  w.hi[i] = w.N;
  w.state[i] = 0;
}

__global__ void __launch_bounds__(256)
clear_stats_kernel(Work w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w.N) return;
  w.cnt[i] = 0;
  w.s_ext[i] = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    w.s_pos[i * 3 + a] = 0.0;
    w.s_rgb[i * 3 + a] = 0.0;
    w.t_min[i * 3 + a] = 0x7fffffff;
    w.t_max[i * 3 + a] = (int)0x80000000;
  }
}

__global__ void __launch_bounds__(256)
level_stats_kernel(Work w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < w.N && w.state[i] == 0;
  const int h = active ? w.lo[i] : -1;
  // Near the root every lane of a wave sits in the same node: reduce in the wave and issue one
  // atomic per wave and statistic instead of 64 on one address (sorted order makes this the
  // common case for nodes of more than a few hundred points).
  const int hmax = wave_max_i32(h);
  if (hmax < 0) return;                                            // no active lane
  const bool uniform = __all(!active || h == hmax);
  float p[3] = {0.f, 0.f, 0.f}, c[3] = {0.f, 0.f, 0.f};
  float e = 0.f;
  if (active) {
    e = w.ext[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      p[a] = w.pos[i * 3 + a];
      c[a] = w.rgb[i * 3 + a];
    }
  }
  if (uniform) {
    const int n = wave_sum_i32(active ? 1 : 0);
    const double se = wave_sum_f64(active ? (double)e : 0.0);
    double sp[3], sc[3];
    int mn[3], mx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      sp[a] = wave_sum_f64(active ? (double)p[a] : 0.0);
      sc[a] = wave_sum_f64(active ? (double)c[a] : 0.0);
      mx[a] = wave_max_i32(active ? f2o(p[a]) : (int)0x80000000);
      mn[a] = -wave_max_i32(active ? -f2o(p[a]) : -0x7fffffff);
    }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&w.cnt[hmax], n);
      atomicAdd(&w.s_ext[hmax], se);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        atomicAdd(&w.s_pos[hmax * 3 + a], sp[a]);
        atomicAdd(&w.s_rgb[hmax * 3 + a], sc[a]);
        atomicMin(&w.t_min[hmax * 3 + a], mn[a]);
        atomicMax(&w.t_max[hmax * 3 + a], mx[a]);
      }
    }
    return;
  }
  if (!active) return;
  atomicAdd(&w.cnt[h], 1);
  atomicAdd(&w.s_ext[h], (double)e);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    atomicAdd(&w.s_pos[h * 3 + a], (double)p[a]);
    atomicAdd(&w.s_rgb[h * 3 + a], (double)c[a]);
    atomicMin(&w.t_min[h * 3 + a], f2o(p[a]));
    atomicMax(&w.t_max[h * 3 + a], f2o(p[a]));
  }
}

__device__ __forceinline__ float cmax3(const float *d) { return fmaxf(d[0], fmaxf(d[1], d[2])); }
__device__ __forceinline__ float cmin3(const float *d) { return fminf(d[0], fminf(d[1], d[2])); }

// impl.cpp:352-421 for every unresolved point: all points of a node read the same statistics and
// take the same decision
__global__ void __launch_bounds__(256)
level_decide_kernel(Work w, int level, float max_aspect, float min_mult) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w.N || w.state[i] != 0) return;
  const int h = w.lo[i], n = w.cnt[h];
  if (n == 1 || level >= MAX_LEVELS) {          // (a node that never resolves: emit its points)
    w.state[i] = 1;
    return;
  }
  const float avg = (float)(w.s_ext[h] / (double)n);
  float td[3], od[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    td[a] = o2f(w.t_max[h * 3 + a]) - o2f(w.t_min[h * 3 + a]);
    od[a] = w.bmax[i * 3 + a] - w.bmin[i * 3 + a];
  }
  const float orig_ar = cmax3(od) / cmin3(od);
  const float tight_ar = cmax3(td) / cmin3(td);
  const float ar = (tight_ar < orig_ar) ? tight_ar : orig_ar;       // std::min(orig, tight): NaN -> orig
  const float thr = cmax3(td);
  if (ar <= max_aspect && thr <= min_mult * avg) {                   // merge (impl.cpp:370-386)
    if (i == h) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        w.m_pos[i * 3 + a] = (float)(w.s_pos[h * 3 + a] / (double)n);
        w.m_rgb[i * 3 + a] = (float)(w.s_rgb[h * 3 + a] / (double)n);
      }
      w.state[i] = 2;
    } else {
      w.state[i] = 3;
    }
    return;
  }
  if (n <= 2) {                                                       // impl.cpp:388-395
    w.state[i] = 1;
    return;
  }
  // descend: the children are the zero / one runs of this level's bit inside [lo, hi)
  const int bit = 62 - level, a = (level + 1) % 3;
  int l = h, r = w.hi[i];
  const int hi0 = r;
  while (l < r) {                       // first index whose bit is set
    const int m = (l + r) >> 1;
    if ((w.code[m] >> bit) & 1ull) r = m;
    else l = m + 1;
  }
  const float split = (w.bmin[i * 3 + a] + w.bmax[i * 3 + a]) / 2.0f;
  if ((w.code[i] >> bit) & 1ull) {
    w.lo[i] = l;
    w.hi[i] = hi0;
    w.bmin[i * 3 + a] = split;
  } else {
    w.hi[i] = l;
    w.bmax[i * 3 + a] = split;
  }
}

__global__ void __launch_bounds__(256)
emit_flags_kernel(Work w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w.N) return;
  const int s = w.state[i];
  w.emit[i] = (s == 1 || s == 2 || s == 0) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
emit_kernel(Work w, float *__restrict__ out_pos, float *__restrict__ out_rgb, int *__restrict__ out_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w.N) return;
  if (i == w.N - 1) *out_count = w.offs[i] + w.emit[i];
  if (!w.emit[i]) return;
  const int o = w.offs[i];
  const bool merged = w.state[i] == 2;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    out_pos[o * 3 + a] = merged ? w.m_pos[i * 3 + a] : w.pos[i * 3 + a];
    out_rgb[o * 3 + a] = merged ? w.m_rgb[i * 3 + a] : w.rgb[i * 3 + a];
  }
}

struct Carver {     // bump allocator over the caller's workspace
  char *base;
  size_t off = 0;
  template <typename T>
  T *take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// ---- key sort and prefix sum (hand-written; rocPRIM served here until round 3) ----------------------
// Stable LSD radix sort of (uint64 key, uint32 value) pairs, 8 bits per pass, eight passes over the 64 key
// bits, ping-pong between the caller's two buffer pairs (an even number of passes: the result ends in the
// buffers it started in). Per pass:
//   rs_hist     tile of 2048 keys per workgroup -> 256 digit counts (LDS atomics) -> hist[digit][tile]
//   scan_i32    exclusive prefix sum of hist, read digit-major: the global offset of (digit, tile)
//   rs_scatter  the tile again in eight rounds of 256 keys in index order; a key's position is
//               offset[digit][tile] + keys of that digit in earlier rounds (LDS counters)
//               + keys of that digit in earlier waves of the round + its rank in the wave (match-any
//               by eight ballots) -- every term counts keys that precede it in the input: stable.
constexpr int RS_THREADS = 256, RS_ROUNDS = 8, RS_TILE = RS_THREADS * RS_ROUNDS, RS_DIGITS = 256;

__global__ void __launch_bounds__(RS_THREADS)
rs_hist_kernel(int64_t n, const uint64_t *__restrict__ keys, int shift, int n_tiles, int *__restrict__ hist) {
  __shared__ int h[RS_DIGITS];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t i = base + r * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255)], 1);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive prefix sum of a[0..n) in place, one workgroup walking the array in chunks of 4096
__global__ void __launch_bounds__(1024)
scan_i32_kernel(int64_t n, const int *in, int *out) {   // (in == out allowed: a chunk is read before it is written)
  __shared__ int wsum[16], carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 4096) {
    const int64_t i0 = base + (int64_t)tid * 4;
    int v[4], local = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = i0 + k < n ? in[i0 + k] : 0;
      local += v[k];
    }
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int prefix = carry_s, total = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wave) prefix += wsum[w];
      total += wsum[w];
    }
    int run = prefix + incl - local;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) out[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == 0) carry_s += total;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(RS_THREADS)
rs_scatter_kernel(int64_t n, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                  uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int shift, int n_tiles,
                  const int *__restrict__ offsets) {
  __shared__ int run[RS_DIGITS], wcount[RS_THREADS / 64][RS_DIGITS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  run[tid] = offsets[(int64_t)tid * n_tiles + blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_ROUNDS; ++r) {
#pragma unroll
    for (int w = 0; w < RS_THREADS / 64; ++w) wcount[w][tid] = 0;
    __syncthreads();
    const int64_t i = base + r * RS_THREADS + tid;
    const bool live = i < n;
    const uint64_t key = live ? keys[i] : ~0ull;
    const int d = (int)((key >> shift) & 255);
    // lanes of this wave with the same digit (dead lanes, at the very end of the input, match only each other)
    uint64_t same = __ballot(live);
    same = live ? same : ~same;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bal = __ballot((d >> b) & 1);
      same &= ((d >> b) & 1) ? bal : ~bal;
    }
    const int rank = __popcll(same & ((1ull << lane) - 1));
    if (live && rank == 0) wcount[wave][d] = __popcll(same);
    __syncthreads();
    if (live) {
      int pos = run[d] + rank;
      for (int w = 0; w < wave; ++w) pos += wcount[w][d];
      keys_out[pos] = key;
      vals_out[pos] = vals[i];
    }
    __syncthreads();
    int add = 0;
#pragma unroll
    for (int w = 0; w < RS_THREADS / 64; ++w) add += wcount[w][tid];
    run[tid] += add;
    __syncthreads();
  }
}

static inline int rs_tiles(int64_t n) { return (int)ceil_div64(n > 0 ? n : 1, RS_TILE); }
static inline int64_t rs_hist_ints(int64_t n) { return (int64_t)RS_DIGITS * rs_tiles(n); }

// sorts (keys, vals) by key, ascending and stable; keys_alt / vals_alt / hist are scratch
static int sort_pairs_u64(int64_t n, uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                          int *hist, hipStream_t st) {
  if (n <= 0) return GSR_OK;
  const int tiles = rs_tiles(n);
  uint64_t *ka = keys, *kb = keys_alt;
  uint32_t *va = vals, *vb = vals_alt;
  for (int pass = 0; pass < 8; ++pass) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3(tiles), dim3(RS_THREADS), 0, st, n, ka, pass * 8, tiles, hist);
    hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, st, rs_hist_ints(n), hist, hist);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(tiles), dim3(RS_THREADS), 0, st, n, ka, va, kb, vb, pass * 8, tiles, hist);
    uint64_t *tk = ka; ka = kb; kb = tk;
    uint32_t *tv = va; va = vb; vb = tv;
  }
  return GSR_OK;
}

struct Layout {
  uint64_t *codes_in, *codes_out;
  uint32_t *idx_in, *idx_out;
  int *mm;
  float *cube;
  Work w;
  int *sort_hist;
  size_t total;
};

static Layout carve(void *ws, int N) {
  Layout L;
  Carver c{(char *)ws};
  const size_t n = (size_t)(N > 0 ? N : 1);
  L.codes_in = c.take<uint64_t>(n);
  L.codes_out = c.take<uint64_t>(n);
  L.idx_in = c.take<uint32_t>(n);
  L.idx_out = c.take<uint32_t>(n);
  L.mm = c.take<int>(8);
  L.cube = c.take<float>(8);
  Work &w = L.w;
  w.N = N;
  w.code = L.codes_in;     // the eight-pass sort ends in the buffers it started in
  w.idx = L.idx_in;
  w.pos = c.take<float>(3 * n);
  w.rgb = c.take<float>(3 * n);
  w.ext = c.take<float>(n);
  w.bmin = c.take<float>(3 * n);
  w.bmax = c.take<float>(3 * n);
  w.lo = c.take<int>(n);
  w.hi = c.take<int>(n);
  w.state = c.take<int>(n);
  w.cnt = c.take<int>(n);
  w.s_ext = c.take<double>(n);
  w.s_pos = c.take<double>(3 * n);
  w.s_rgb = c.take<double>(3 * n);
  w.t_min = c.take<int>(3 * n);
  w.t_max = c.take<int>(3 * n);
  w.m_pos = c.take<float>(3 * n);
  w.m_rgb = c.take<float>(3 * n);
  w.emit = c.take<int>(n);
  w.offs = c.take<int>(n);
  L.sort_hist = c.take<int>((size_t)rs_hist_ints((int64_t)n));
  L.total = c.off + 256;
  return L;
}

}  // namespace pc
}  // namespace gsr

using namespace gsr::pc;

extern "C" int gsr_pc_min_extents(int N, int C, const float *points, const float *Ks, const float *Ps,
                                  const int32_t *image_sizes, float *extents, void *stream) {
  GSR_REQUIRE(N >= 0 && C >= 0, "pc_min_extents: bad sizes");
  if (N == 0) return GSR_OK;
  GSR_REQUIRE(points && extents && (C == 0 || (Ks && Ps && image_sizes)), "pc_min_extents: null pointer");
  hipLaunchKernelGGL(min_extents_kernel, dim3((unsigned)gsr::ceil_div(N, 256)), dim3(256), 0,
                     (hipStream_t)stream, N, C, points, Ks, Ps, image_sizes, extents);
  GSR_CHECK_LAUNCH("pc_min_extents");
  return GSR_OK;
}

// The sort by itself (tests; the subsampler calls it on its Morton codes). keys / vals are sorted in place,
// keys_alt / vals_alt [n] and hist [256 * ceil(n / 2048)] int32 are scratch.
extern "C" int gsr_sort_pairs_u64(int64_t n, uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                                  int32_t *hist, int64_t hist_ints, void *stream) {
  GSR_REQUIRE(n >= 0, "sort_pairs_u64: bad n");
  if (n == 0) return GSR_OK;
  GSR_REQUIRE(keys && keys_alt && vals && vals_alt && hist && hist_ints >= rs_hist_ints(n),
              "sort_pairs_u64: null pointer or %lld < %lld histogram words", (long long)hist_ints, (long long)rs_hist_ints(n));
  sort_pairs_u64(n, keys, keys_alt, vals, vals_alt, hist, (hipStream_t)stream);
  GSR_CHECK_LAUNCH("sort_pairs_u64");
  return GSR_OK;
}

extern "C" int64_t gsr_pc_subsample_workspace_bytes(int N) {
  if (N < 0) return -1;
  return (int64_t)carve(nullptr, N).total;
}

extern "C" int gsr_pc_subsample(int N, const float *points, const float *rgbs, const float *extents,
                                float max_bbox_aspect_ratio, float min_extent_multiplier, void *workspace,
                                int64_t workspace_bytes, float *out_points, float *out_rgbs,
                                int32_t *out_count, void *stream) {
  GSR_REQUIRE(N >= 0, "pc_subsample: bad size");
  GSR_REQUIRE(out_count, "pc_subsample: null out_count");
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    GSR_CHECK_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
    return GSR_OK;
  }
  GSR_REQUIRE(points && rgbs && extents && workspace && out_points && out_rgbs, "pc_subsample: null pointer");
  Layout L = carve(workspace, N);
  GSR_REQUIRE((int64_t)L.total <= workspace_bytes, "pc_subsample: workspace %lld < %lld bytes",
              (long long)workspace_bytes, (long long)L.total);
  const dim3 grid((unsigned)gsr::ceil_div(N, 256)), block(256);
  const int init[8] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0};
  GSR_CHECK_HIP(hipMemcpyAsync(L.mm, init, sizeof(init), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(bbox_kernel, grid, block, 0, st, N, points, L.mm);
  hipLaunchKernelGGL(cube_kernel, dim3(1), dim3(64), 0, st, L.mm, L.cube);
  hipLaunchKernelGGL(codes_kernel, grid, block, 0, st, N, points, L.cube, L.codes_in, L.idx_in);
  sort_pairs_u64(N, L.codes_in, L.codes_out, L.idx_in, L.idx_out, L.sort_hist, st);
  hipLaunchKernelGGL(gather_kernel, grid, block, 0, st, L.w, points, rgbs, extents, L.cube);
  for (int level = 0; level <= MAX_LEVELS; ++level) {
    hipLaunchKernelGGL(clear_stats_kernel, grid, block, 0, st, L.w);
    hipLaunchKernelGGL(level_stats_kernel, grid, block, 0, st, L.w);
    hipLaunchKernelGGL(level_decide_kernel, grid, block, 0, st, L.w, level, max_bbox_aspect_ratio,
                       min_extent_multiplier);
  }
  hipLaunchKernelGGL(emit_flags_kernel, grid, block, 0, st, L.w);
  hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, st, (int64_t)N, L.w.emit, L.w.offs);
  hipLaunchKernelGGL(emit_kernel, grid, block, 0, st, L.w, out_points, out_rgbs, out_count);
  GSR_CHECK_LAUNCH("pc_subsample");
  return GSR_OK;
}
