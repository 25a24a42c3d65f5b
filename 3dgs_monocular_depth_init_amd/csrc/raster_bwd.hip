// raster_bwd.hip -- A7: per-Gaussian gradients of the compositing.
//
// Replaces gsplat's rasterize_to_pixels backward that loss.backward()
// (gs_init_compare/runner.py:547) triggers. Same shape as the forward: one
// wave64 per 16x16 tile, 2x2 pixels per lane, tile list replayed back to front
// in LDS batches of 64. Per Gaussian the lane sums its 4 pixels in registers,
// the wave reduces on DPP (no LDS), and the batch's totals are parked in LDS
// as [64][16] rows; each row is then flushed with ONE 64-byte-aligned group
// of float atomics into grad_rows[g][16] (16 adjacent lanes = one memory-side
// atomic request), instead of 9-11 scattered dword atomics per Gaussian.
#include "common.h"
#include "gs_math.h"

namespace gsr {

template <int CH, bool ABSGRAD>
__global__ void __launch_bounds__(64)
raster_bwd_kernel(int n_tiles, int N, const float *__restrict__ means2d,
                  const float *__restrict__ conics, const float *__restrict__ colors,
                  int color_stride, const float *__restrict__ opacities, int opac_per_camera,
                  const float *__restrict__ backgrounds, int width, int height, int tile_w,
                  int tile_h, const int32_t *__restrict__ tile_offsets,
                  const int32_t *__restrict__ flatten_ids,
                  const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
                  const float *__restrict__ v_render_colors,
                  const float *__restrict__ v_render_alphas, float *__restrict__ grad_rows) {
  constexpr int NC = (CH > 2) ? (CH - 2) : 1;
  __shared__ float4 sA[2][64];
  __shared__ float4 sB[2][64];
  __shared__ float sC[2][64][NC];
  __shared__ int sId[2][64];
  __shared__ float sG[64][GSR_GRAD_ROW];  // reduced gradients of the batch
  __shared__ int sTouched[64];

  const int tile = xcd_remap(blockIdx.x, n_tiles);
  if (tile >= n_tiles) return;
  const int tiles_per_cam = tile_w * tile_h;
  const int cam = tile / tiles_per_cam;
  const int tin = tile - cam * tiles_per_cam;
  const int ty = tin / tile_w, tx = tin - ty * tile_w;
  const int lane = threadIdx.x;
  const int qx = lane & 7, qy = lane >> 3;
  const int x0 = tx * GSR_TILE + 2 * qx, y0 = ty * GSR_TILE + 2 * qy;

  const int s = tile_offsets[tile], e = tile_offsets[tile + 1];
  if (e <= s) return;

  float px[4], py[4], T[4], Tfin[4], buf_c[4][CH], vout[4][CH], valpha[4];
  int last[4];
  int max_last = -1;
  float bgdot[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = x0 + (p & 1), y = y0 + (p >> 1);
    px[p] = (float)x + 0.5f;
    py[p] = (float)y + 0.5f;
    const bool inside = (x < width) && (y < height);
    last[p] = -1;
    Tfin[p] = 1.f;
    valpha[p] = 0.f;
    bgdot[p] = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      vout[p][k] = 0.f;
      buf_c[p][k] = 0.f;
    }
    if (inside) {
      const int64_t pix = ((int64_t)cam * height + y) * width + x;
      last[p] = last_ids[pix];
      Tfin[p] = 1.0f - render_alphas[pix];
      valpha[p] = v_render_alphas[pix];
#pragma unroll
      for (int k = 0; k < CH; ++k) vout[p][k] = v_render_colors[pix * CH + k];
      if (backgrounds) {
#pragma unroll
        for (int k = 0; k < CH; ++k) bgdot[p] += backgrounds[cam * CH + k] * vout[p][k];
      }
    }
    T[p] = Tfin[p];
    max_last = max(max_last, last[p]);
  }
  max_last = wave_max_i32(max_last);
  if (max_last < s) return;  // nothing was blended into this tile
  const int start = min(max_last, e - 1);

  // lane l of a batch holds Gaussian (batch_end - l): j = 0 is the LAST one.
  float4 rA, rB;
  float rC[NC];
  int rId = 0;
  auto gather = [&](int idx) {
    const int g = flatten_ids[idx];
    rId = g;
    const float2 m = *reinterpret_cast<const float2 *>(means2d + (int64_t)g * 2);
    const float *cn = conics + (int64_t)g * 3;
    const float *cl = colors + (int64_t)g * color_stride;
    const float op = opacities[opac_per_camera ? g : (g % N)];
    rA = make_float4(m.x, m.y, cn[0], cn[1]);
    rB = make_float4(cn[2], op, cl[0], (CH > 1) ? cl[1] : 0.f);
#pragma unroll
    for (int k = 2; k < CH; ++k) rC[k - 2] = cl[k];
  };
  if (start - lane >= s) gather(start - lane);

  int buf = 0;
  for (int batch_end = start; batch_end >= s; batch_end -= 64) {
    const int n = min(64, batch_end - s + 1);
    if (lane < n) {
      sA[buf][lane] = rA;
      sB[buf][lane] = rB;
#pragma unroll
      for (int k = 0; k < NC; ++k) sC[buf][lane][k] = rC[k];
      sId[buf][lane] = rId;
    }
    sTouched[lane] = 0;
    __syncthreads();
    const int nb = batch_end - 64;
    if (nb - lane >= s) gather(nb - lane);

    for (int j = 0; j < n; ++j) {
      const int idx = batch_end - j;
      const float4 A = sA[buf][j];
      const float4 B = sB[buf][j];
      float col[CH];
      col[0] = B.z;
      if (CH > 1) col[1] = B.w;
#pragma unroll
      for (int k = 2; k < CH; ++k) col[k] = sC[buf][j][k - 2];
      const float opac = B.y;

      float dx[4], dy[4], vis[4], alpha[4];
      bool valid[4];
      bool any_valid = false;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        dx[p] = A.x - px[p];
        dy[p] = A.y - py[p];
        const float sigma = 0.5f * (A.z * dx[p] * dx[p] + B.x * dy[p] * dy[p]) + A.w * dx[p] * dy[p];
        vis[p] = __expf(-sigma);
        alpha[p] = fminf(gs::ALPHA_MAX, opac * vis[p]);
        valid[p] = (idx <= last[p]) && (sigma >= 0.f) && (alpha[p] >= gs::ALPHA_THRESHOLD);
        any_valid |= valid[p];
      }
      if (!__any(any_valid)) continue;  // wave-uniform skip

      float g_xy[2] = {0.f, 0.f}, g_con[3] = {0.f, 0.f, 0.f}, g_op = 0.f, g_col[CH];
      float g_abs[2] = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < CH; ++k) g_col[k] = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (valid[p]) {
          const float ra = 1.0f / (1.0f - alpha[p]);
          T[p] *= ra;
          const float fac = alpha[p] * T[p];
          float v_alpha = 0.f;
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            g_col[k] += fac * vout[p][k];
            v_alpha += (col[k] * T[p] - buf_c[p][k] * ra) * vout[p][k];
          }
          v_alpha += Tfin[p] * ra * valpha[p];
          if (backgrounds) v_alpha -= Tfin[p] * ra * bgdot[p];
          if (opac * vis[p] <= gs::ALPHA_MAX) {
            const float v_sigma = -opac * vis[p] * v_alpha;
            g_con[0] += 0.5f * v_sigma * dx[p] * dx[p];
            g_con[1] += v_sigma * dx[p] * dy[p];
            g_con[2] += 0.5f * v_sigma * dy[p] * dy[p];
            const float vx = v_sigma * (A.z * dx[p] + A.w * dy[p]);
            const float vy = v_sigma * (A.w * dx[p] + B.x * dy[p]);
            g_xy[0] += vx;
            g_xy[1] += vy;
            if (ABSGRAD) {
              g_abs[0] += fabsf(vx);
              g_abs[1] += fabsf(vy);
            }
            g_op += vis[p] * v_alpha;
          }
#pragma unroll
          for (int k = 0; k < CH; ++k) buf_c[p][k] += col[k] * fac;
        }
      }
      // wave reduction on DPP; totals are wave-uniform after readlane
      const float r_x = wave_sum(g_xy[0]), r_y = wave_sum(g_xy[1]);
      const float r_a = wave_sum(g_con[0]), r_b = wave_sum(g_con[1]), r_c = wave_sum(g_con[2]);
      const float r_o = wave_sum(g_op);
      float r_col[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) r_col[k] = wave_sum(g_col[k]);
      float r_ax = 0.f, r_ay = 0.f;
      if (ABSGRAD) {
        r_ax = wave_sum(g_abs[0]);
        r_ay = wave_sum(g_abs[1]);
      }
      // lane f writes field f of row j
      float val = 0.f;
      val = (lane == GSR_GR_MEAN2D) ? r_x : val;
      val = (lane == GSR_GR_MEAN2D + 1) ? r_y : val;
      val = (lane == GSR_GR_CONIC) ? r_a : val;
      val = (lane == GSR_GR_CONIC + 1) ? r_b : val;
      val = (lane == GSR_GR_CONIC + 2) ? r_c : val;
      val = (lane == GSR_GR_OPAC) ? r_o : val;
#pragma unroll
      for (int k = 0; k < CH; ++k) val = (lane == GSR_GR_COLOR + k) ? r_col[k] : val;
      if (ABSGRAD) {
        val = (lane == GSR_GR_ABS) ? r_ax : val;
        val = (lane == GSR_GR_ABS + 1) ? r_ay : val;
      }
      if (lane < GSR_GRAD_ROW) sG[j][lane] = val;
      if (lane == 0) sTouched[j] = 1;
    }
    __syncthreads();
    // flush: 4 Gaussians per wave instruction, 16 lanes = one 64-byte row
    const int f = lane & 15;
    const bool field_used = (f < GSR_GR_COLOR + CH) || (ABSGRAD && (f == GSR_GR_ABS || f == GSR_GR_ABS + 1));
    for (int j0 = 0; j0 < n; j0 += 4) {
      const int j = j0 + (lane >> 4);
      if (j < n && field_used && sTouched[j]) {
        const int g = sId[buf][j];
        atomicAdd(grad_rows + (int64_t)g * GSR_GRAD_ROW + f, sG[j][f]);
      }
    }
    __syncthreads();
    buf ^= 1;
  }
}

template <int CH>
static int launch_bwd(int n_tiles, int N, const float *means2d, const float *conics,
                      const float *colors, int color_stride, const float *opacities,
                      int opac_per_camera, const float *backgrounds, int width, int height,
                      int tile_w, int tile_h, const int32_t *tile_offsets,
                      const int32_t *flatten_ids, const float *render_alphas,
                      const int32_t *last_ids, const float *v_render_colors,
                      const float *v_render_alphas, int absgrad, float *grad_rows,
                      hipStream_t stream) {
  if (absgrad)
    hipLaunchKernelGGL((raster_bwd_kernel<CH, true>), dim3(xcd_grid(n_tiles)), dim3(64), 0, stream,
                       n_tiles, N, means2d, conics, colors, color_stride, opacities,
                       opac_per_camera, backgrounds, width, height, tile_w, tile_h, tile_offsets,
                       flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas,
                       grad_rows);
  else
    hipLaunchKernelGGL((raster_bwd_kernel<CH, false>), dim3(xcd_grid(n_tiles)), dim3(64), 0, stream,
                       n_tiles, N, means2d, conics, colors, color_stride, opacities,
                       opac_per_camera, backgrounds, width, height, tile_w, tile_h, tile_offsets,
                       flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas,
                       grad_rows);
  GSR_CHECK_LAUNCH("rasterize_bwd");
  return GSR_OK;
}

}  // namespace gsr

extern "C" int gsr_rasterize_bwd(int C, int N, int CH, const float *means2d, const float *conics,
                                 const float *colors, int color_stride, const float *opacities,
                                 int opac_per_camera, const float *backgrounds, int width,
                                 int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                                 const int32_t *flatten_ids, const float *render_alphas,
                                 const int32_t *last_ids, const float *v_render_colors,
                                 const float *v_render_alphas, int absgrad, float *grad_rows,
                                 void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "rasterize_bwd: bad sizes");
  GSR_REQUIRE(tile_w == gsr::ceil_div(width, GSR_TILE) && tile_h == gsr::ceil_div(height, GSR_TILE),
              "rasterize_bwd: tile grid does not match image");
  GSR_REQUIRE(CH >= 1 && CH <= 5 && color_stride >= CH, "rasterize_bwd: CH=%d stride=%d", CH,
              color_stride);
  if (C == 0 || N == 0) return GSR_OK;
  GSR_REQUIRE(means2d && conics && colors && opacities && tile_offsets &&
                  render_alphas && last_ids && v_render_colors && v_render_alphas && grad_rows,
              "rasterize_bwd: null pointer");
  int n_tiles = C * tile_w * tile_h;
  hipStream_t st = (hipStream_t)stream;
#define GSR_BWD_CASE(K)                                                                       \
  case K:                                                                                     \
    return gsr::launch_bwd<K>(n_tiles, N, means2d, conics, colors, color_stride, opacities,   \
                              opac_per_camera, backgrounds, width, height, tile_w, tile_h,    \
                              tile_offsets, flatten_ids, render_alphas, last_ids,             \
                              v_render_colors, v_render_alphas, absgrad, grad_rows, st);
  switch (CH) {
    GSR_BWD_CASE(1)
    GSR_BWD_CASE(2)
    GSR_BWD_CASE(3)
    GSR_BWD_CASE(4)
    GSR_BWD_CASE(5)
  }
#undef GSR_BWD_CASE
  return GSR_EINVAL;
}
