// raster_fwd.hip -- A6: front-to-back alpha compositing.
//
// Replaces gsplat's rasterize_to_pixels forward reached from
// gs_init_compare/runner.py:341. CDNA4 shape: ONE wave64 per 16x16 tile, every
// lane owns one pixel of each 8x8 quadrant (4 independent blend chains per lane,
// 4x fewer LDS broadcast reads per pixel-Gaussian pair than one pixel per thread,
// no cross-wave barriers, wave-uniform skips per quadrant and per row of
// quadrants). The tile's depth-sorted list is streamed through LDS in batches of
// 32 Gaussians: the 64-byte records of batch k+1 are copied global -> LDS by LDS-DMA
// (no registers, no arithmetic: the records are stored in the loop's own units and
// the quadrant masks arrive with the pair words) while batch k is composited.
// Round 3 rewrite: the round-2 kernel re-derived the masks while staging
// (min_sigma_rect x 4 per pair), which cost 18 % of the wave time and, through its
// register pressure at the 64-VGPR / 8-waves budget, put the prefetched record
// registers into scratch -- every batch then waited for its own prefetch
// (profiles/r03_fwd_timeline_before.json). Also built and rejected in round 3: reading each record
// with scalar loads into SGPRs instead of staging it (no LDS, 54 VGPRs): 0.198 ms against 0.145 ms,
// every record is a scalar-cache miss (profiles/r03_ab_scalar_records.log).

#include <type_traits>

#include "raster_common.h"

#ifndef GSR_FWD_WAVES
#define GSR_FWD_WAVES 8   // waves per SIMD the CH <= 3 instantiation is register-allocated for
#endif

namespace gsr {

template <int CH>
__global__ void __launch_bounds__(64, (CH <= 3) ? GSR_FWD_WAVES : 6)
raster_fwd_kernel(int n_tiles, const float *__restrict__ records,
                  const float *__restrict__ backgrounds, int width, int height, int tile_w,
                  int tile_h, const int32_t *__restrict__ tile_offsets,
                  const int32_t *__restrict__ tile_order,
                  const int32_t *__restrict__ pair_ids, float *__restrict__ render_colors,
                  float *__restrict__ render_alphas, int32_t *__restrict__ last_ids,
                  float4 *__restrict__ zero_rows, int64_t n_zero16, const float *__restrict__ l1_target,
                  float l1_scale, double *__restrict__ l1_partials, int planar) {
  // staged batches: row r = {mx, my, ha, bb | hc, opacity, col0, col1 | col2, col3, col4, - | -}.
  // 2 x 2 KB per wave; the workgroup IS one wave, so no barriers: the DMA into buffer (k+1)&1 is
  // issued after the loop over batch k-1 has consumed its last read of that buffer.
  __shared__ __attribute__((aligned(16))) float4 sRec[2][RBATCH][4];
  __shared__ uint32_t sPw[PW_SLOTS][RBATCH];

  if ((int)blockIdx.x >= n_tiles) return;
  // The backward's gradient rows ([C*N, 16] floats) must start at zero. This kernel is bound by
  // instruction issue, not by HBM, so every wave clears its slice of them on the side (a handful of
  // fire-and-forget 1 KB stores) instead of a 64 MB fill launch in front of the backward.
  if (zero_rows) {
    const int64_t per = (n_zero16 + n_tiles - 1) / n_tiles;           // 16-byte pieces per workgroup
    const int64_t z0 = (int64_t)blockIdx.x * per, z1 = min(n_zero16, z0 + per);
    for (int64_t z = z0 + threadIdx.x; z < z1; z += 64) zero_rows[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int tile = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
  const int tiles_per_cam = tile_w * tile_h;
  const int cam = tile / tiles_per_cam;
  const int tin = tile - cam * tiles_per_cam;
  const int ty = tin / tile_w, tx = tin - ty * tile_w;
  const int lane = threadIdx.x;
  const int lx = lane & 7, ly = lane >> 3;
  const int tx0 = tx * GSR_TILE, ty0 = ty * GSR_TILE;

  // pixel q of this lane: (tx0 + 8*(q&1) + lx, ty0 + 8*(q>>1) + ly)
  float px[4], py[2], T[4], acc[4][CH];
  unsigned outside = 0;
  py[0] = (float)(ty0 + ly) + 0.5f;
  py[1] = py[0] + 8.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = tx0 + 8 * (q & 1) + lx, y = ty0 + 8 * (q >> 1) + ly;
    px[q] = (float)x + 0.5f;
    T[q] = 1.0f;
    if (x >= width || y >= height) {
      outside |= 1u << q;
      px[q] = PIX_DONE;
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[q][k] = 0.f;
  }
  // linear pixel index of quadrant q's pixel of this lane; recomputed where it is needed (a rare
  // path and the epilogue) instead of living in registers across the loops
  auto pix_of = [&](int q) {
    const int l = opaque(lane);
    return ((int64_t)cam * height + (ty0 + 8 * (q >> 1) + (l >> 3))) * width + (tx0 + 8 * (q & 1) + (l & 7));
  };

  const int s = tile_offsets[tile], e = tile_offsets[tile + 1];
  const int last = e - 1;
  unsigned live = 0xfu;   // wave-uniform: quadrants that still have an unfinished pixel
  TL_DECL();
  if (e > s) {
    // pipeline: pair words two batches ahead, records one batch ahead, both by LDS-DMA
    dma_pair_words<1>(pair_ids, s, s, last, lane, sPw[0]);
    dma_pair_words<1>(pair_ids, s + RBATCH, s, last, lane, sPw[1]);
    GSR_WAIT_VMEM();
    dma_stage_batch(records, sPw[0], lane, sRec[0]);
    int buf = 0, slot = 0;   // slot = batch number % 3
    for (int base = s; base < e; base += RBATCH, buf ^= 1, slot = (slot == 2) ? 0 : slot + 1) {
      // drop quadrants whose 64 pixels are all finished (or outside the image)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (!__any(px[q] != PIX_DONE)) live &= ~(1u << q);
      if (live == 0) break;
      const int n = min(RBATCH, e - base);
      TL_MARK(3);                            // (3) = everything outside the three segments below
      GSR_WAIT_VMEM();                       // this batch's records and the next batch's words are in LDS
      TL_MARK(0);                            // (0) waiting for the staged batch
      if (base + RBATCH < e) {
        const int s1 = (slot == 2) ? 0 : slot + 1, s2 = (s1 == 2) ? 0 : s1 + 1;
        dma_stage_batch(records, sPw[s1], lane, sRec[buf ^ 1]);
        dma_pair_words<1>(pair_ids, base + 2 * RBATCH, s, last, lane, sPw[s2]);
      }
      TL_MARK(1);                            // (1) issuing the next batch's DMAs
#ifdef GSR_RASTER_TIMELINE
      ++tl_batches;
#endif
      const uint32_t pw = sPw[slot][lane & 31];   // lane j < n: pair j's word (mask in the top bits)
      // opacity > 0.999 somewhere in the batch (the pair words' clamp flags): alpha may hit the clamp
      const bool can_clamp = (lane < n) && (pw & PAIR_CLAMP_BIT);
      const float4(*rec)[4] = sRec[buf];

      // One Gaussian against the tile. NOCLAMP (wave-uniform per batch): no opacity of the
      // batch exceeds 0.999, so min(0.999, .) is compiled out.
      auto composite = [&](auto noclamp_tag, int j) {
        constexpr bool NOCLAMP = decltype(noclamp_tag)::value;
        const unsigned qm =
            ((unsigned)__builtin_amdgcn_readlane((int)pw, j) >> PAIR_MASK_SHIFT) & live;
        if (qm == 0) return;
        const float4 Ac = rec[j][0], Bc = rec[j][1];
        float col[CH];
        col[0] = Bc.z;
        if (CH > 1) col[1] = Bc.w;
        if (CH > 2) col[2] = rec[j][2].x;
        if (CH > 3) col[3] = rec[j][2].y;
        if (CH > 4) col[4] = rec[j][2].z;
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
                if (__builtin_expect(__any(nT <= gs::T_THRESHOLD), 0)) {   // rare: a pixel finishes at this Gaussian
                  const bool stop = nT <= gs::T_THRESHOLD;   // only possible when ok
                  a = stop ? 0.f : a;
                  nT = stop ? T[q] : nT;
                  px[q] = stop ? PIX_DONE : px[q];
                  // this Gaussian is NOT blended: the pixel's last list position is the one before
                  if (stop) last_ids[pix_of(q)] = base + j - 1;
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
      TL_MARK(2);                            // (2) the compositing loop
    }
    GSR_WAIT_VMEM();   // a DMA issued for a batch the early exit skipped must land before the wave ends
  }
  TL_STORE(0, tile, e - s);

  // l1_target (gsr_rasterize_fwd_l1, the training step with the plain L1 loss, runner.py:506): the loss is taken while
  // the finished pixels are still in registers -- sum |render - target| per tile into l1_partials, and `render_colors`
  // receives d mean|render - target| / d render = sign(render - target) * l1_scale instead of the render (which
  // nothing else of that step reads): the 75 MB pass of gsr_l1_fwd disappears from the step.
  float l1_part = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if ((outside >> q) & 1u) continue;
    const int64_t pix = pix_of(q);
    // render_colors is [C,H,W,CH], or (planar: what the SSIM kernels read three times faster) [C,CH,H,W]
    const int64_t hw = (int64_t)height * width;
    float *out = planar ? render_colors + (int64_t)cam * (CH - 1) * hw + pix : render_colors + pix * CH;
    const int64_t kstride = planar ? hw : 1;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      float v = acc[q][k];
      if (backgrounds) v += T[q] * backgrounds[cam * CH + k];
      if (l1_target) {
        const float d = v - l1_target[pix * CH + k];
        l1_part += fabsf(d);
        v = d > 0.f ? l1_scale : (d < 0.f ? -l1_scale : 0.f);
      }
      out[k * kstride] = v;
    }
    render_alphas[pix] = 1.0f - T[q];
    // list position after which nothing is blended into this pixel: where it terminated (stored
    // at that moment), else the end of the tile's list (the backward re-tests alpha >= 1/255 per
    // pair itself, so the exact position of the last contributor is not needed and not tracked)
    if (px[q] != PIX_DONE) last_ids[pix] = last;
  }
  if (l1_target) {
    const float tile_sum = wave_sum(l1_part);
    if (lane == 0) l1_partials[blockIdx.x] = (double)tile_sum;
  }
}

// mean_out[0] = (sum of the per-tile partial sums) * inv_n
__global__ void __launch_bounds__(1024)
l1_tiles_finalize_kernel(int n_partials, const double *__restrict__ partials, double inv_n, float *__restrict__ mean_out) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += 1024) acc += partials[i];
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    mean_out[0] = (float)(t * inv_n);
  }
}

// Experiment knob (tools/ab_env.sh, profiles/r03_ab_occupancy.log; builds with -DGSR_EXPERIMENT_KNOBS only): extra
// dynamic LDS per workgroup lowers the number of resident waves without touching the code object:
// GSR_FWD_LDS_PAD / GSR_BWD_LDS_PAD = bytes. The product build has the constant 0 here.
static unsigned occupancy_pad(const char *var) { return (unsigned)gsr_knob_int(var, 0); }

template <int CH>
static int launch_fwd(int n_tiles, const float *records, const float *backgrounds, int width,
                      int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                      const int32_t *tile_order, const int32_t *pair_ids, float *render_colors,
                      float *render_alphas, int32_t *last_ids, float *zero_rows, int64_t n_zero_rows,
                      hipStream_t stream, const float *l1_target = nullptr, float l1_scale = 0.f,
                      double *l1_partials = nullptr, int planar = 0) {
  hipLaunchKernelGGL(raster_fwd_kernel<CH>, dim3(n_tiles), dim3(64), occupancy_pad("GSR_FWD_LDS_PAD"), stream, n_tiles, records,
                     backgrounds, width, height, tile_w, tile_h, tile_offsets, tile_order,
                     pair_ids, render_colors, render_alphas, last_ids, reinterpret_cast<float4 *>(zero_rows),
                     n_zero_rows * (GSR_GRAD_ROW / 4), l1_target, l1_scale, l1_partials, planar);
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
  write_record(records, g, means2d[g * 2], means2d[g * 2 + 1], conics[g * 3], conics[g * 3 + 1],
               conics[g * 3 + 2], opacities[opac_per_camera ? g : (g % N)], c);
}

}  // namespace gsr

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

static int rasterize_fwd_impl(int C, int CH, const float *records, const float *backgrounds,
                              int width, int height, int tile_w, int tile_h,
                              const int32_t *tile_offsets, const int32_t *tile_order,
                              const int32_t *pair_ids, float *render_colors,
                              float *render_alphas, int32_t *last_ids, float *zero_rows,
                              int64_t n_zero_rows, int planar, void *stream) {
  GSR_REQUIRE(C >= 0 && width > 0 && height > 0, "rasterize_fwd: bad sizes");
  GSR_REQUIRE(tile_w == gsr::ceil_div(width, GSR_TILE) && tile_h == gsr::ceil_div(height, GSR_TILE),
              "rasterize_fwd: tile grid %dx%d does not match %dx%d image", tile_w, tile_h, width,
              height);
  GSR_REQUIRE(CH >= 1 && CH <= 5, "rasterize_fwd: CH=%d", CH);
  if (C == 0) return GSR_OK;
  GSR_REQUIRE(tile_offsets && render_colors && render_alphas && last_ids,
              "rasterize_fwd: null pointer");
  GSR_REQUIRE(((uintptr_t)records & 15) == 0, "rasterize_fwd: records must be 16-byte aligned");
  GSR_REQUIRE(!zero_rows || (n_zero_rows >= 0 && ((uintptr_t)zero_rows & 15) == 0),
              "rasterize_fwd: zero_rows must be 16-byte aligned");
  int n_tiles = C * tile_w * tile_h;
  hipStream_t st = (hipStream_t)stream;
#define GSR_FWD_CASE(K)                                                                         \
  case K:                                                                                       \
    return gsr::launch_fwd<K>(n_tiles, records, backgrounds, width, height, tile_w, tile_h,     \
                              tile_offsets, tile_order, pair_ids, render_colors,                \
                              render_alphas, last_ids, zero_rows, n_zero_rows, st, nullptr, 0.f, nullptr, planar);
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

extern "C" int gsr_rasterize_fwd(int C, int CH, const float *records, const float *backgrounds,
                                 int width, int height, int tile_w, int tile_h,
                                 const int32_t *tile_offsets, const int32_t *tile_order,
                                 const int32_t *pair_ids, float *render_colors,
                                 float *render_alphas, int32_t *last_ids, float *zero_rows,
                                 int64_t n_zero_rows, void *stream) {
  return rasterize_fwd_impl(C, CH, records, backgrounds, width, height, tile_w, tile_h, tile_offsets, tile_order,
                            pair_ids, render_colors, render_alphas, last_ids, zero_rows, n_zero_rows, 0, stream);
}
// The same with render_colors laid out [C,CH,H,W] (planes) instead of [C,H,W,CH]: what the fused L1 + SSIM loss
// kernels (gsr_ssim_l1_fwd / _bwd, strides per image) read without the 3x line traffic of channel-interleaved memory.
extern "C" int gsr_rasterize_fwd_planar(int C, int CH, const float *records, const float *backgrounds,
                                        int width, int height, int tile_w, int tile_h,
                                        const int32_t *tile_offsets, const int32_t *tile_order,
                                        const int32_t *pair_ids, float *render_colors,
                                        float *render_alphas, int32_t *last_ids, float *zero_rows,
                                        int64_t n_zero_rows, void *stream) {
  return rasterize_fwd_impl(C, CH, records, backgrounds, width, height, tile_w, tile_h, tile_offsets, tile_order,
                            pair_ids, render_colors, render_alphas, last_ids, zero_rows, n_zero_rows, 1, stream);
}

// The compositing forward of a training step whose loss is the plain L1 (runner.py:506, ssim_lambda = 0), three
// channels: `grad_out` [C,H,W,3] receives d mean|render - target| / d render = sign(render - target) / (C H W 3) (what
// gsr_rasterize_bwd takes as v_render_colors under a root gradient of 1), mean_out[0] the loss; the render itself is
// not written. l1_partials: n_tiles device doubles (scratch).
extern "C" int gsr_rasterize_fwd_l1(int C, const float *records, const float *backgrounds, int width,
                                    int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                                    const int32_t *tile_order, const int32_t *pair_ids,
                                    const float *target, float *grad_out, float *render_alphas,
                                    int32_t *last_ids, float *zero_rows, int64_t n_zero_rows,
                                    double *l1_partials, float *mean_out, void *stream) {
  GSR_REQUIRE(C >= 0 && width > 0 && height > 0, "rasterize_fwd_l1: bad sizes");
  GSR_REQUIRE(tile_w == gsr::ceil_div(width, GSR_TILE) && tile_h == gsr::ceil_div(height, GSR_TILE),
              "rasterize_fwd_l1: tile grid %dx%d does not match %dx%d image", tile_w, tile_h, width, height);
  if (C == 0) return GSR_OK;
  GSR_REQUIRE(tile_offsets && target && grad_out && render_alphas && last_ids && l1_partials && mean_out,
              "rasterize_fwd_l1: null pointer");
  GSR_REQUIRE(((uintptr_t)records & 15) == 0, "rasterize_fwd_l1: records must be 16-byte aligned");
  GSR_REQUIRE(!zero_rows || (n_zero_rows >= 0 && ((uintptr_t)zero_rows & 15) == 0),
              "rasterize_fwd_l1: zero_rows must be 16-byte aligned");
  const int n_tiles = C * tile_w * tile_h;
  const double n = (double)C * height * width * 3.0;
  hipStream_t st = (hipStream_t)stream;
  const int rc = gsr::launch_fwd<3>(n_tiles, records, backgrounds, width, height, tile_w, tile_h, tile_offsets, tile_order,
                                    pair_ids, grad_out, render_alphas, last_ids, zero_rows, n_zero_rows, st, target,
                                    (float)(1.0 / n), l1_partials);
  if (rc != GSR_OK) return rc;
  hipLaunchKernelGGL(gsr::l1_tiles_finalize_kernel, dim3(1), dim3(1024), 0, st, n_tiles, l1_partials, 1.0 / n, mean_out);
  GSR_CHECK_LAUNCH("rasterize_fwd_l1 (finalize)");
  return GSR_OK;
}

#ifdef GSR_RASTER_TIMELINE
// buf: [n_tiles, 8] uint64 on the device (or NULL to switch off)
extern "C" int gsr_debug_set_fwd_timeline(void *buf) {
  unsigned long long *p = (unsigned long long *)buf;
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_raster_timeline), &p, sizeof(p), 0));
  return GSR_OK;
}
#endif
