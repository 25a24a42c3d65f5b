// raster_fwd.hip -- A6: front-to-back alpha compositing.
//
// Replaces gsplat's rasterize_to_pixels forward reached from
// gs_init_compare/runner.py:341. CDNA4 shape: ONE wave64 per 16x16 tile, every
// lane owns one pixel of each 8x8 quadrant (4 independent blend chains per lane,
// 4x fewer LDS broadcast reads per pixel-Gaussian pair than one pixel per thread,
// no cross-wave barriers, wave-uniform skips per quadrant and per row of
// quadrants). The tile's depth-sorted list is streamed through LDS in batches of
// 64 Gaussians; the gather of batch k+1 is in flight while batch k is composited.
// 64 VGPRs -> 8 waves/SIMD, so that all 8160 tiles of a 1080p frame are resident
// at once (measured: 0.178 ms against 0.19 ms at 5-7 waves/SIMD, where a second
// round of tiles trails the first).
#include <type_traits>

#include "raster_common.h"

namespace gsr {

#ifdef GSR_FWD_TIMELINE
// Diagnostic build only (tools/fwd_timeline.py): per tile, cycles spent in the segments of a
// batch, summed over the tile's batches: [0] waiting for the gathered records, [1] building
// the LDS image (make_rec, ds_write, barrier), [2] issuing the next batch's loads, [3] the
// compositing loop, [4] whole kernel, [5] batches, [6] list length. s_memtime stamps, as
// cdna_hip_programming.md section 7 prescribes (one asm statement with its lgkmcnt(0),
// sched barriers around it); the values leave through a buffer nothing else reads.
__device__ unsigned long long *g_fwd_timeline = nullptr;
#define GSR_STAMP(t)                                                        \
  do {                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                      \
  } while (0)
#endif

template <int CH>
__global__ void __launch_bounds__(64, (CH <= 3) ? 8 : 5)
raster_fwd_kernel(int n_tiles, const float *__restrict__ records,
                  const float *__restrict__ backgrounds, int width, int height, int tile_w,
                  int tile_h, const int32_t *__restrict__ tile_offsets,
                  const int32_t *__restrict__ tile_order,
                  const int32_t *__restrict__ flatten_ids, float *__restrict__ render_colors,
                  float *__restrict__ render_alphas, int32_t *__restrict__ last_ids) {
  // staged batch: [0] = {mx, my, ha, bb}, [1] = {hc, opacity, col0, col1},
  // [2] = {col2, col3, col4, quadrant mask}. 3 KB per wave; a single buffer is enough
  // because the workgroup IS one wave: its LDS writes for the next batch follow its reads
  // of the current one in program order.
  __shared__ float4 sRec[1][3][64];

  if ((int)blockIdx.x >= n_tiles) return;
  const int tile = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
  const int tiles_per_cam = tile_w * tile_h;
  const int cam = tile / tiles_per_cam;
  const int tin = tile - cam * tiles_per_cam;
  const int ty = tin / tile_w, tx = tin - ty * tile_w;
  const int lane = threadIdx.x;
  const int lx = lane & 7, ly = lane >> 3;
  const int tx0 = tx * GSR_TILE, ty0 = ty * GSR_TILE;
  // tile origin as floats held in SCALAR registers (wave-uniform; a VGPR copy would be
  // hoisted out of the loops and cost the compositing loops registers)
  const float ftx0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)tx0)));
  const float fty0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)ty0)));

  // pixel q of this lane: (tx0 + 8*(q&1) + lx, ty0 + 8*(q>>1) + ly)
  float px[4], py[2], T[4], acc[4][CH];
  int last[4];
  unsigned outside = 0;
  py[0] = (float)(ty0 + ly) + 0.5f;
  py[1] = py[0] + 8.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = tx0 + 8 * (q & 1) + lx, y = ty0 + 8 * (q >> 1) + ly;
    px[q] = (float)x + 0.5f;
    T[q] = 1.0f;
    last[q] = 0x7fffffff;   // "never terminated": resolved to the end of the list at the end
    if (x >= width || y >= height) {
      outside |= 1u << q;
      px[q] = PIX_DONE;
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[q][k] = 0.f;
  }

  const int s = tile_offsets[tile], e = tile_offsets[tile + 1];
  // software pipeline: ids two batches ahead, record rows one batch ahead (in flight
  // during the compositing loop), LDS image built after the loop
  RawRec<CH> raw;
  TileRec<CH> rec;
  // (loads are unconditional with clamped indices: a predicated load would merge old and
  // new register values, and the copies that merge needs wait for the load right away)
  if (e <= s) {
    raw.r0 = raw.r1 = raw.r2 = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    load_raw<CH>(flatten_ids[min(s + lane, e - 1)], records, raw);
  }
  int id_next = (e > s) ? flatten_ids[min(s + 64 + lane, e - 1)] : 0;
  constexpr int buf = 0;
  unsigned live = 0xfu;   // wave-uniform: quadrants that still have an unfinished pixel
#ifdef GSR_FWD_TIMELINE
  unsigned long long tl_t0, tl_a, tl_b, tl_c, tl_d, tl_e;
  unsigned long long tl_wait = 0, tl_rec = 0, tl_issue = 0, tl_loop = 0, tl_batches = 0;
  GSR_STAMP(tl_t0);
#endif
  for (int base = s; base < e; base += 64) {
    // drop quadrants whose 64 pixels are all finished (or outside the image)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (!__any(px[q] != PIX_DONE)) live &= ~(1u << q);
    if (live == 0) break;
    const int n = min(64, e - base);
    bool can_clamp = false;   // opacity > 0.999: alpha may hit the clamp
#ifdef GSR_FWD_TIMELINE
    GSR_STAMP(tl_a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GSR_STAMP(tl_b);
#endif
    if (lane < n) {
      make_rec<CH>(raw, ftx0, fty0, rec);
      can_clamp = rec.b.y > gs::ALPHA_MAX;
      sRec[buf][0][lane] = rec.a;
      sRec[buf][1][lane] = rec.b;
      sRec[buf][2][lane] = rec.c;
    }
    __syncthreads();
#ifdef GSR_FWD_TIMELINE
    GSR_STAMP(tl_c);
#endif
    load_raw<CH>(id_next, records, raw);
    id_next = flatten_ids[min(base + 128 + lane, e - 1)];
#ifdef GSR_FWD_TIMELINE
    GSR_STAMP(tl_d);
#endif

    // One Gaussian against the tile. NOCLAMP (wave-uniform per batch): no opacity of the
    // batch exceeds 0.999, so min(0.999, .) is compiled out.
    auto composite = [&](auto noclamp_tag, int j) {
      constexpr bool NOCLAMP = decltype(noclamp_tag)::value;
      const unsigned qm =
          (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(sRec[buf][2][j].w)) & live;
      if (qm == 0) return;
      const float4 Ac = sRec[buf][0][j], Bc = sRec[buf][1][j];
      float col[CH];
      col[0] = Bc.z;
      if (CH > 1) col[1] = Bc.w;
      if (CH > 2) col[2] = sRec[buf][2][j].x;
      if (CH > 3) col[3] = sRec[buf][2][j].y;
      if (CH > 4) col[4] = sRec[buf][2][j].z;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (qm & (3u << (2 * r))) {   // scalar: this row of quadrants is touched
          float Br, Cr;
          sigma_row_terms(Ac.w, Bc.x, Ac.y - py[r], Br, Cr);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int q = 2 * r + h;
            if (qm & (1u << q)) {   // scalar branch
              const float dx = Ac.x - px[q];
              const float sg = sigma_l2(Ac.z, dx, Br, Cr);
              const float ov = Bc.y * __builtin_amdgcn_exp2f(-sg);
              // valid <=> sigma >= 0 and alpha >= 1/255 (<=> ov >= 1/255): one compare
              const bool ok = with_sign_of(ov, sg) >= gs::ALPHA_THRESHOLD;
              float a = ok ? (NOCLAMP ? ov : fminf(gs::ALPHA_MAX, ov)) : 0.f;
              float nT = fmaf(-a, T[q], T[q]);
              if (__any(nT <= gs::T_THRESHOLD)) {   // rare: a pixel finishes at this Gaussian
                const bool stop = nT <= gs::T_THRESHOLD;   // only possible when ok
                a = stop ? 0.f : a;
                nT = stop ? T[q] : nT;
                px[q] = stop ? PIX_DONE : px[q];
                last[q] = stop ? base + j - 1 : last[q];   // this one is NOT blended
              }
              const float w = a * T[q];
              T[q] = nT;
#pragma unroll
              for (int k = 0; k < CH; ++k) acc[q][k] = fmaf(col[k], w, acc[q][k]);
            }
          }
        }
      }
    };
    if (!__any(can_clamp)) {
      for (int j = 0; j < n; ++j) composite(std::true_type{}, j);
    } else {
      for (int j = 0; j < n; ++j) composite(std::false_type{}, j);
    }
#ifdef GSR_FWD_TIMELINE
    GSR_STAMP(tl_e);
    tl_wait += tl_b - tl_a;
    tl_rec += tl_c - tl_b;
    tl_issue += tl_d - tl_c;
    tl_loop += tl_e - tl_d;
    tl_batches += 1;
#endif
  }
#ifdef GSR_FWD_TIMELINE
  GSR_STAMP(tl_e);
  if (g_fwd_timeline && lane == 0) {
    unsigned long long *o = g_fwd_timeline + 8 * (size_t)blockIdx.x;
    o[0] = tl_wait; o[1] = tl_rec; o[2] = tl_issue; o[3] = tl_loop;
    o[4] = tl_e - tl_t0; o[5] = tl_batches; o[6] = (unsigned long long)(e - s); o[7] = tl_t0;
  }
#endif

#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if ((outside >> q) & 1u) continue;
    const int x = tx0 + 8 * (q & 1) + lx, y = ty0 + 8 * (q >> 1) + ly;
    const int64_t pix = ((int64_t)cam * height + y) * width + x;
    float *out = render_colors + pix * CH;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      float v = acc[q][k];
      if (backgrounds) v += T[q] * backgrounds[cam * CH + k];
      out[k] = v;
    }
    render_alphas[pix] = 1.0f - T[q];
    // list position after which nothing is blended into this pixel: where it terminated, else
    // the end of the tile's list (the backward re-tests alpha >= 1/255 per pair itself, so the
    // exact position of the last contributor is not needed and not tracked)
    last_ids[pix] = (last[q] == 0x7fffffff) ? e - 1 : last[q];
  }
}

template <int CH>
static int launch_fwd(int n_tiles, const float *records, const float *backgrounds, int width,
                      int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                      const int32_t *tile_order, const int32_t *flatten_ids, float *render_colors,
                      float *render_alphas, int32_t *last_ids, hipStream_t stream) {
  hipLaunchKernelGGL(raster_fwd_kernel<CH>, dim3(n_tiles), dim3(64), 0, stream, n_tiles, records,
                     backgrounds, width, height, tile_w, tile_h, tile_offsets, tile_order,
                     flatten_ids, render_colors, render_alphas, last_ids);
  GSR_CHECK_LAUNCH("rasterize_fwd");
  return GSR_OK;
}

// Pack caller-supplied per-pair arrays into compositing records (the projection
// kernel writes the records itself on the SH path).
__global__ void __launch_bounds__(256)
pack_records_kernel(int64_t total, int N, int CH, const float *__restrict__ means2d,
                    const float *__restrict__ conics, const float *__restrict__ colors,
                    int color_stride, const float *__restrict__ opacities, int opac_per_camera,
                    float *__restrict__ records) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const float *cl = colors + g * color_stride;
  float c[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < CH; ++k) c[k] = cl[k];
  float4 *row = reinterpret_cast<float4 *>(records + g * REC_FLOATS);
  row[0] = make_float4(means2d[g * 2], means2d[g * 2 + 1], conics[g * 3], conics[g * 3 + 1]);
  row[1] = make_float4(conics[g * 3 + 2], opacities[opac_per_camera ? g : (g % N)], c[0], c[1]);
  row[2] = make_float4(c[2], c[3], c[4], 0.f);
}

}  // namespace gsr

#ifdef GSR_FWD_TIMELINE
extern "C" int gsr_debug_set_fwd_timeline(void *buf) {
  unsigned long long *p = (unsigned long long *)buf;
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_fwd_timeline), &p, sizeof(p)));
  return GSR_OK;
}
#endif

extern "C" int gsr_pack_records(int C, int N, int CH, const float *means2d, const float *conics,
                                const float *colors, int color_stride, const float *opacities,
                                int opac_per_camera, float *records, void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && CH >= 1 && CH <= 5 && color_stride >= CH, "pack_records: bad sizes");
  const int64_t total = (int64_t)C * N;
  if (total == 0) return GSR_OK;
  GSR_REQUIRE(means2d && conics && colors && opacities && records, "pack_records: null pointer");
  hipLaunchKernelGGL(gsr::pack_records_kernel, dim3((unsigned)gsr::ceil_div64(total, 256)),
                     dim3(256), 0, (hipStream_t)stream, total, N, CH, means2d, conics, colors,
                     color_stride, opacities, opac_per_camera, records);
  GSR_CHECK_LAUNCH("pack_records");
  return GSR_OK;
}

extern "C" int gsr_rasterize_fwd(int C, int CH, const float *records, const float *backgrounds,
                                 int width, int height, int tile_w, int tile_h,
                                 const int32_t *tile_offsets, const int32_t *tile_order,
                                 const int32_t *flatten_ids, float *render_colors,
                                 float *render_alphas, int32_t *last_ids, void *stream) {
  GSR_REQUIRE(C >= 0 && width > 0 && height > 0, "rasterize_fwd: bad sizes");
  GSR_REQUIRE(tile_w == gsr::ceil_div(width, GSR_TILE) && tile_h == gsr::ceil_div(height, GSR_TILE),
              "rasterize_fwd: tile grid %dx%d does not match %dx%d image", tile_w, tile_h, width,
              height);
  GSR_REQUIRE(CH >= 1 && CH <= 5, "rasterize_fwd: CH=%d", CH);
  if (C == 0) return GSR_OK;
  GSR_REQUIRE(tile_offsets && render_colors && render_alphas && last_ids,
              "rasterize_fwd: null pointer");
  int n_tiles = C * tile_w * tile_h;
  hipStream_t st = (hipStream_t)stream;
#define GSR_FWD_CASE(K)                                                                         \
  case K:                                                                                       \
    return gsr::launch_fwd<K>(n_tiles, records, backgrounds, width, height, tile_w, tile_h,     \
                              tile_offsets, tile_order, flatten_ids, render_colors,             \
                              render_alphas, last_ids, st);
  switch (CH) {
    GSR_FWD_CASE(1)
    GSR_FWD_CASE(2)
    GSR_FWD_CASE(3)
    GSR_FWD_CASE(4)
    GSR_FWD_CASE(5)
  }
#undef GSR_FWD_CASE
  return GSR_EINVAL;
}
