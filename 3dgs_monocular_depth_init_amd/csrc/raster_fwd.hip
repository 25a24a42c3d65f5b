// raster_fwd.hip -- A6: front-to-back alpha compositing.
//
// Replaces gsplat's rasterize_to_pixels forward reached from
// gs_init_compare/runner.py:341. CDNA4 shape: ONE wave64 per 16x16 tile, every
// lane owns a 2x2 pixel quad (4 independent blend chains per lane -> ILP, 4x
// fewer LDS broadcast reads per pixel-Gaussian pair than one pixel per thread,
// no cross-wave barriers, wave-uniform early exit). The tile's depth-sorted
// list is streamed through LDS in batches of 64 Gaussians, double-buffered:
// the gather of batch k+1 is in flight while batch k is composited.
#include "common.h"
#include "gs_math.h"

namespace gsr {

template <int CH>
struct GaussRec {          // what one lane gathers for one Gaussian
  float4 a;                // mx, my, conic a, conic b
  float4 b;                // conic c, opacity, col0, col1
  float c[(CH > 2) ? (CH - 2) : 1];  // remaining colour channels
};

template <int CH>
__device__ __forceinline__ void gather_gauss(int g, int N, const float *__restrict__ means2d,
                                             const float *__restrict__ conics,
                                             const float *__restrict__ colors, int color_stride,
                                             const float *__restrict__ opacities,
                                             int opac_per_camera, GaussRec<CH> &r) {
  float2 m = *reinterpret_cast<const float2 *>(means2d + (int64_t)g * 2);
  const float *cn = conics + (int64_t)g * 3;
  const float *cl = colors + (int64_t)g * color_stride;
  float op = opacities[opac_per_camera ? g : (g % N)];
  r.a = make_float4(m.x, m.y, cn[0], cn[1]);
  float c0 = cl[0];
  float c1 = (CH > 1) ? cl[1] : 0.f;
  r.b = make_float4(cn[2], op, c0, c1);
#pragma unroll
  for (int k = 2; k < CH; ++k) r.c[k - 2] = cl[k];
}

template <int CH>
__global__ void __launch_bounds__(64)
raster_fwd_kernel(int n_tiles, int N, const float *__restrict__ means2d,
                  const float *__restrict__ conics, const float *__restrict__ colors,
                  int color_stride, const float *__restrict__ opacities, int opac_per_camera,
                  const float *__restrict__ backgrounds, int width, int height, int tile_w,
                  int tile_h, const int32_t *__restrict__ tile_offsets,
                  const int32_t *__restrict__ flatten_ids, float *__restrict__ render_colors,
                  float *__restrict__ render_alphas, int32_t *__restrict__ last_ids) {
  constexpr int NC = (CH > 2) ? (CH - 2) : 1;
  __shared__ float4 sA[2][64];
  __shared__ float4 sB[2][64];
  __shared__ float sC[2][64][NC];

  const int tile = xcd_remap(blockIdx.x, n_tiles);
  if (tile >= n_tiles) return;
  const int tiles_per_cam = tile_w * tile_h;
  const int cam = tile / tiles_per_cam;
  const int tin = tile - cam * tiles_per_cam;
  const int ty = tin / tile_w, tx = tin - ty * tile_w;
  const int lane = threadIdx.x;
  const int qx = lane & 7, qy = lane >> 3;
  const int x0 = tx * GSR_TILE + 2 * qx, y0 = ty * GSR_TILE + 2 * qy;

  float px[4], py[4], T[4], acc[4][CH];
  int last[4];
  unsigned done = 0;  // bit p: pixel p finished (or outside the image)
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int x = x0 + (p & 1), y = y0 + (p >> 1);
    px[p] = (float)x + 0.5f;
    py[p] = (float)y + 0.5f;
    T[p] = 1.0f;
    last[p] = -1;
    if (x >= width || y >= height) done |= 1u << p;
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[p][k] = 0.f;
  }
  const unsigned outside = done;

  const int s = tile_offsets[tile], e = tile_offsets[tile + 1];
  GaussRec<CH> rec;
  if (s + lane < e)
    gather_gauss<CH>(flatten_ids[s + lane], N, means2d, conics, colors, color_stride, opacities,
                     opac_per_camera, rec);
  int buf = 0;
  for (int base = s; base < e; base += 64) {
    if (__all(done == 0xfu)) break;  // wave-uniform: every pixel of the tile is saturated
    const int n = min(64, e - base);
    if (lane < n) {
      sA[buf][lane] = rec.a;
      sB[buf][lane] = rec.b;
#pragma unroll
      for (int k = 0; k < NC; ++k) sC[buf][lane][k] = rec.c[k];
    }
    __syncthreads();
    // prefetch the next batch while this one is composited
    const int nb = base + 64;
    if (nb + lane < e)
      gather_gauss<CH>(flatten_ids[nb + lane], N, means2d, conics, colors, color_stride,
                       opacities, opac_per_camera, rec);
    {
      for (int j = 0; j < n; ++j) {
        const float4 A = sA[buf][j];
        const float4 B = sB[buf][j];
        float col[CH];
        col[0] = B.z;
        if (CH > 1) col[1] = B.w;
#pragma unroll
        for (int k = 2; k < CH; ++k) col[k] = sC[buf][j][k - 2];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float dx = A.x - px[p], dy = A.y - py[p];
          const float sigma = 0.5f * (A.z * dx * dx + B.x * dy * dy) + A.w * dx * dy;
          const float alpha = fminf(gs::ALPHA_MAX, B.y * __expf(-sigma));
          const bool ok = (sigma >= 0.f) && (alpha >= gs::ALPHA_THRESHOLD) && !((done >> p) & 1u);
          if (ok) {
            const float nT = T[p] * (1.0f - alpha);
            if (nT <= gs::T_THRESHOLD) {
              done |= 1u << p;
            } else {
              const float w = alpha * T[p];
#pragma unroll
              for (int k = 0; k < CH; ++k) acc[p][k] += col[k] * w;
              T[p] = nT;
              last[p] = base + j;
            }
          }
        }
      }
    }
    buf ^= 1;
  }

#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if ((outside >> p) & 1u) continue;
    const int x = x0 + (p & 1), y = y0 + (p >> 1);
    const int64_t pix = ((int64_t)cam * height + y) * width + x;
    float *out = render_colors + pix * CH;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      float v = acc[p][k];
      if (backgrounds) v += T[p] * backgrounds[cam * CH + k];
      out[k] = v;
    }
    render_alphas[pix] = 1.0f - T[p];
    last_ids[pix] = last[p];
  }
}

template <int CH>
static int launch_fwd(int n_tiles, int N, const float *means2d, const float *conics,
                      const float *colors, int color_stride, const float *opacities,
                      int opac_per_camera, const float *backgrounds, int width, int height,
                      int tile_w, int tile_h, const int32_t *tile_offsets,
                      const int32_t *flatten_ids, float *render_colors, float *render_alphas,
                      int32_t *last_ids, hipStream_t stream) {
  hipLaunchKernelGGL(raster_fwd_kernel<CH>, dim3(xcd_grid(n_tiles)), dim3(64), 0, stream, n_tiles,
                     N, means2d, conics, colors, color_stride, opacities, opac_per_camera,
                     backgrounds, width, height, tile_w, tile_h, tile_offsets, flatten_ids,
                     render_colors, render_alphas, last_ids);
  GSR_CHECK_LAUNCH("rasterize_fwd");
  return GSR_OK;
}

}  // namespace gsr

extern "C" int gsr_rasterize_fwd(int C, int N, int CH, const float *means2d, const float *conics,
                                 const float *colors, int color_stride, const float *opacities,
                                 int opac_per_camera, const float *backgrounds, int width,
                                 int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                                 const int32_t *flatten_ids, float *render_colors,
                                 float *render_alphas, int32_t *last_ids, void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "rasterize_fwd: bad sizes");
  GSR_REQUIRE(tile_w == gsr::ceil_div(width, GSR_TILE) && tile_h == gsr::ceil_div(height, GSR_TILE),
              "rasterize_fwd: tile grid %dx%d does not match %dx%d image", tile_w, tile_h, width,
              height);
  GSR_REQUIRE(CH >= 1 && CH <= 5 && color_stride >= CH, "rasterize_fwd: CH=%d stride=%d", CH,
              color_stride);
  if (C == 0) return GSR_OK;
  GSR_REQUIRE(tile_offsets && render_colors && render_alphas && last_ids,
              "rasterize_fwd: null pointer");
  GSR_REQUIRE(N == 0 || (means2d && conics && colors && opacities),
              "rasterize_fwd: null Gaussian arrays");
  int n_tiles = C * tile_w * tile_h;
  hipStream_t st = (hipStream_t)stream;
#define GSR_FWD_CASE(K)                                                                      \
  case K:                                                                                    \
    return gsr::launch_fwd<K>(n_tiles, N, means2d, conics, colors, color_stride, opacities,  \
                              opac_per_camera, backgrounds, width, height, tile_w, tile_h,   \
                              tile_offsets, flatten_ids, render_colors, render_alphas,       \
                              last_ids, st);
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
