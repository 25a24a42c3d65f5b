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
#include <type_traits>

#include "raster_common.h"

namespace gsr {

// Sum over the 16 lanes of each DPP row; every lane of a row ends with the row sum.
__device__ __forceinline__ float row_sum16(float v) {
  v = dpp_add<0xb1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4e>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x124>(v);   // row_ror:4
  v = dpp_add<0x128>(v);   // row_ror:8
  return v;
}

template <int CH, bool ABSGRAD>
__global__ void __launch_bounds__(64, (CH <= 3 && !ABSGRAD) ? 5 : 4)
raster_bwd_kernel(int n_tiles, const float *__restrict__ records,
                  const float *__restrict__ backgrounds, int width, int height, int tile_w,
                  int tile_h, const int32_t *__restrict__ tile_offsets,
                  const int32_t *__restrict__ tile_order,
                  const int32_t *__restrict__ flatten_ids,
                  const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
                  const float *__restrict__ v_render_colors,
                  const float *__restrict__ v_render_alphas, float *__restrict__ grad_rows) {
  __shared__ float4 sA[1][64];   // single buffer: the batch-end barriers already order reuse
  __shared__ float4 sB[1][64];
  // CH <= 3: {col2, quadrant mask}; CH 4,5: {col2, col3, col4, mask} (LDS per wave
  // decides how many waves fit a CU: 9.7 KB -> 16 waves)
  using CT = typename std::conditional<(CH <= 3), float2, float4>::type;
  __shared__ CT sC[1][64];
  __shared__ int sId[1][64];
  __shared__ __attribute__((aligned(16))) float sG[64][GSR_GRAD_ROW];  // batch gradient rows

  if ((int)blockIdx.x >= n_tiles) return;
  const int tile = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
  const int tiles_per_cam = tile_w * tile_h;
  const int cam = tile / tiles_per_cam;
  const int tin = tile - cam * tiles_per_cam;
  const int ty = tin / tile_w, tx = tin - ty * tile_w;
  const int lane = threadIdx.x;
  const int lx = lane & 7, ly = lane >> 3;
  const int tx0 = tx * GSR_TILE, ty0 = ty * GSR_TILE;

  const int s = tile_offsets[tile], e = tile_offsets[tile + 1];
  if (e <= s) return;
  // which field of a gradient row this lane stores after tree_reduce8 (-1: none).
  // tree values: 0,1 = first moments (slots MEAN2D), 2..4 = conic, 5..7 = colours 0..2
  int tree_field = -1;
  if ((lane & 7) == 0) {
    const int ti = tree8_index(lane);
    tree_field = (ti < 2) ? GSR_GR_MEAN2D + ti
                 : (ti < 5) ? GSR_GR_CONIC + (ti - 2)
                 : ((ti - 5) < CH ? GSR_GR_COLOR + (ti - 5) : -1);
  }

  // Kq = T_final * (v_alpha_out - <background, v_out>): the per-pixel constant of v_alpha
  float px[4], py[2], T[4], Kq[4], buf_c[4][CH], vout[4][CH];
  int last[4], qmax[4];
  py[0] = (float)(ty0 + ly) + 0.5f;
  py[1] = py[0] + 8.0f;
  int max_last = -1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = tx0 + 8 * (q & 1) + lx, y = ty0 + 8 * (q >> 1) + ly;
    px[q] = (float)x + 0.5f;
    last[q] = -1;
    T[q] = 1.f;
    Kq[q] = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      vout[q][k] = 0.f;
      buf_c[q][k] = 0.f;
    }
    if (x < width && y < height) {
      const int64_t pix = ((int64_t)cam * height + y) * width + x;
      last[q] = last_ids[pix];
      T[q] = 1.0f - render_alphas[pix];
      float kk = v_render_alphas[pix];
#pragma unroll
      for (int k = 0; k < CH; ++k) vout[q][k] = v_render_colors[pix * CH + k];
      if (backgrounds) {
#pragma unroll
        for (int k = 0; k < CH; ++k) kk -= backgrounds[cam * CH + k] * vout[q][k];
      }
      Kq[q] = T[q] * kk;
    }
    qmax[q] = __builtin_amdgcn_readfirstlane(wave_max_i32(last[q]));
    max_last = max(max_last, qmax[q]);
  }
  if (max_last < s) return;  // nothing was blended into this tile (wave-uniform)
  const int start = min(max_last, e - 1);

  // lane l of a batch stages Gaussian (batch_end - l): j = 0 is the LAST one.
  TileRec<CH> rec;
  int rId = 0;
  if (start - lane >= s) {
    rId = flatten_ids[start - lane];
    stage_gauss<CH>(rId, records, (float)tx0, (float)ty0, rec);
  }

  constexpr int buf = 0;
  for (int batch_end = start; batch_end >= s; batch_end -= 64) {
    const int n = min(64, batch_end - s + 1);
    if (lane < n) {
      sA[buf][lane] = rec.a;
      sB[buf][lane] = rec.b;
      if constexpr (CH <= 3) sC[buf][lane] = make_float2(rec.c.x, rec.c.w);
      else sC[buf][lane] = rec.c;
      sId[buf][lane] = rId;
    }
    sG[lane][15] = 0.f;   // "row touched" flag of Gaussian `lane` of this batch
    __syncthreads();
    const int nb = batch_end - 64;
    if (nb - lane >= s) {
      rId = flatten_ids[nb - lane];
      stage_gauss<CH>(rId, records, (float)tx0, (float)ty0, rec);
    }

    for (int j = 0; j < n; ++j) {
      const float4 Ac = sA[buf][j], Bc = sB[buf][j];
      float4 Cc;
      if constexpr (CH <= 3) {
        const float2 c2 = sC[buf][j];
        Cc = make_float4(c2.x, 0.f, 0.f, c2.y);
      } else {
        Cc = sC[buf][j];
      }
      const int idx = batch_end - j;
      unsigned qm = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(Cc.w));
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (idx > qmax[q]) qm &= ~(1u << q);   // scalar: no pixel of q blended this far down
      if (qm == 0) continue;
      float col[CH];
      col[0] = Bc.z;
      if (CH > 1) col[1] = Bc.w;
      if (CH > 2) col[2] = Cc.x;
      if (CH > 3) col[3] = Cc.y;
      if (CH > 4) col[4] = Cc.z;
      const float opac = Bc.y;
      // conic in natural units for the gradient formulas: a = 2*ha/log2e etc.
      const float ca = Ac.z * (2.0f / LOG2E), cb = Ac.w * (1.0f / LOG2E), cc = Bc.x * (2.0f / LOG2E);

      float g_xy[2] = {0.f, 0.f}, g_con[3] = {0.f, 0.f, 0.f}, g_op = 0.f, g_col[CH];
      float g_abs[2] = {0.f, 0.f};
      bool any_valid = false;
#pragma unroll
      for (int k = 0; k < CH; ++k) g_col[k] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (qm & (1u << q)) {   // scalar branch
          const float dx = Ac.x - px[q], dy = Ac.y - py[q >> 1];
          const float sg = sigma_l2(Ac.z, Ac.w, Bc.x, dx, dy);
          const float vis = __builtin_amdgcn_exp2f(-sg);
          const float alpha = fminf(gs::ALPHA_MAX, opac * vis);
          const bool valid = (idx <= last[q]) && (sg >= 0.f) && (alpha >= gs::ALPHA_THRESHOLD);
          any_valid |= valid;
          const float a = valid ? alpha : 0.f;
          const float ra = __builtin_amdgcn_rcpf(1.0f - a);
          T[q] *= ra;
          const float fac = a * T[q];
          float v_alpha = Kq[q] * ra;
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            g_col[k] = fmaf(fac, vout[q][k], g_col[k]);
            v_alpha = fmaf(col[k] * T[q] - buf_c[q][k] * ra, vout[q][k], v_alpha);
            buf_c[q][k] = fmaf(col[k], fac, buf_c[q][k]);
          }
          const float ov = opac * vis;
          const float va = (valid && ov <= gs::ALPHA_MAX) ? v_alpha : 0.f;
          const float v_sigma = -ov * va;
          const float tdx = v_sigma * dx, tdy = v_sigma * dy;
          g_con[0] = fmaf(0.5f * tdx, dx, g_con[0]);
          g_con[1] = fmaf(tdx, dy, g_con[1]);
          g_con[2] = fmaf(0.5f * tdy, dy, g_con[2]);
          // v_xy = conic * (sum v_sigma*dx, sum v_sigma*dy): only the two first
          // moments are summed per pixel, the 2x2 product is applied once after
          // the reduction (absgrad needs the per-pixel value).
          g_xy[0] += tdx;
          g_xy[1] += tdy;
          if (ABSGRAD) {
            g_abs[0] += fabsf(fmaf(ca, tdx, cb * tdy));
            g_abs[1] += fabsf(fmaf(cb, tdx, cc * tdy));
          }
          g_op = fmaf(vis, va, g_op);
        }
      }
      if (!__any(any_valid)) continue;  // wave-uniform skip
      // 8 of the sums go through the lane-swap halving tree (18 VALU for all 8),
      // the rest through plain wave sums; writer lanes store straight into the
      // batch's 64-byte LDS row.
      {
        const float tv[8] = {g_xy[0], g_xy[1], g_con[0], g_con[1], g_con[2], g_col[0],
                             (CH > 1) ? g_col[1] : 0.f, (CH > 2) ? g_col[2] : 0.f};
        const float u = tree_reduce8(tv, lane);
        const float r_o = wave_sum(g_op);
        float *row = &sG[j][0];
        if (tree_field >= 0) row[tree_field] = u;      // 8 writer lanes
        if (lane == 0) {
          row[GSR_GR_OPAC] = r_o;
          row[15] = 1.0f;                              // "row touched" flag
        }
        if (CH > 3) {
          const float r3 = wave_sum(g_col[3]);
          if (lane == 0) row[GSR_GR_COLOR + 3] = r3;
        }
        if (CH > 4) {
          const float r4 = wave_sum(g_col[4]);
          if (lane == 0) row[GSR_GR_COLOR + 4] = r4;
        }
        if (ABSGRAD) {
          const float r_ax = wave_sum(g_abs[0]), r_ay = wave_sum(g_abs[1]);
          if (lane == 0) {
            row[GSR_GR_ABS] = r_ax;
            row[GSR_GR_ABS + 1] = r_ay;
          }
        }
      }
    }
    __syncthreads();
    // flush: 4 Gaussians per wave instruction, 16 lanes = one 64-byte row
    const int f = lane & 15;
    const bool field_used =
        (f < GSR_GR_COLOR + CH) || (ABSGRAD && (f == GSR_GR_ABS || f == GSR_GR_ABS + 1));
    for (int j0 = 0; j0 < n; j0 += 4) {
      const int j = j0 + (lane >> 4);
      if (j < n && field_used && sG[j][15] != 0.f) {
        const int g = sId[buf][j];
        float val = sG[j][f];
        if (f < 2) {   // v_xy = conic * (first moments): row holds (m_x, m_y)
          const float4 Aj = sA[buf][j];
          const float4 Bj = sB[buf][j];
          const float ca = Aj.z * (2.0f / LOG2E), cb = Aj.w * (1.0f / LOG2E),
                      cc = Bj.x * (2.0f / LOG2E);
          const float mx_ = sG[j][0], my_ = sG[j][1];
          val = (f == 0) ? fmaf(ca, mx_, cb * my_) : fmaf(cb, mx_, cc * my_);
        }
        atomicAdd(grad_rows + (int64_t)g * GSR_GRAD_ROW + f, val);
      }
    }
    __syncthreads();
  }
}

// Test hook for the lane-swap tree (the semantics of v_permlane{16,32}_swap are
// checked on the GPU by tests/test_gpu_rasterization.py::test_tree_reduce8).
__global__ void debug_tree_reduce8_kernel(const float *__restrict__ in, float *__restrict__ out,
                                          int *__restrict__ idx_out) {
  const int lane = threadIdx.x;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = in[k * 64 + lane];
  out[lane] = tree_reduce8(v, lane);
  idx_out[lane] = tree8_index(lane);
  out[64 + lane] = wave_sum(v[0]);
}

template <int CH>
static int launch_bwd(int n_tiles, const float *records, const float *backgrounds, int width,
                      int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                      const int32_t *tile_order, const int32_t *flatten_ids,
                      const float *render_alphas, const int32_t *last_ids,
                      const float *v_render_colors, const float *v_render_alphas, int absgrad,
                      float *grad_rows, hipStream_t stream) {
  if (absgrad)
    hipLaunchKernelGGL((raster_bwd_kernel<CH, true>), dim3(n_tiles), dim3(64), 0, stream, n_tiles,
                       records, backgrounds, width, height, tile_w, tile_h, tile_offsets,
                       tile_order, flatten_ids, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, grad_rows);
  else
    hipLaunchKernelGGL((raster_bwd_kernel<CH, false>), dim3(n_tiles), dim3(64), 0, stream, n_tiles,
                       records, backgrounds, width, height, tile_w, tile_h, tile_offsets,
                       tile_order, flatten_ids, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, grad_rows);
  GSR_CHECK_LAUNCH("rasterize_bwd");
  return GSR_OK;
}

}  // namespace gsr

extern "C" int gsr_debug_tree_reduce8(const float *in, float *out, int32_t *idx_out,
                                      void *stream) {
  GSR_REQUIRE(in && out && idx_out, "debug_tree_reduce8: null pointer");
  hipLaunchKernelGGL(gsr::debug_tree_reduce8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     in, out, idx_out);
  GSR_CHECK_LAUNCH("debug_tree_reduce8");
  return GSR_OK;
}

extern "C" int gsr_rasterize_bwd(int C, int CH, const float *records, const float *backgrounds,
                                 int width, int height, int tile_w, int tile_h,
                                 const int32_t *tile_offsets, const int32_t *tile_order,
                                 const int32_t *flatten_ids, const float *render_alphas,
                                 const int32_t *last_ids, const float *v_render_colors,
                                 const float *v_render_alphas, int absgrad, float *grad_rows,
                                 void *stream) {
  GSR_REQUIRE(C >= 0 && width > 0 && height > 0, "rasterize_bwd: bad sizes");
  GSR_REQUIRE(tile_w == gsr::ceil_div(width, GSR_TILE) && tile_h == gsr::ceil_div(height, GSR_TILE),
              "rasterize_bwd: tile grid does not match image");
  GSR_REQUIRE(CH >= 1 && CH <= 5, "rasterize_bwd: CH=%d", CH);
  if (C == 0) return GSR_OK;
  GSR_REQUIRE(tile_offsets && render_alphas && last_ids && v_render_colors && v_render_alphas &&
                  grad_rows,
              "rasterize_bwd: null pointer");
  int n_tiles = C * tile_w * tile_h;
  hipStream_t st = (hipStream_t)stream;
#define GSR_BWD_CASE(K)                                                                         \
  case K:                                                                                       \
    return gsr::launch_bwd<K>(n_tiles, records, backgrounds, width, height, tile_w, tile_h,     \
                              tile_offsets, tile_order, flatten_ids, render_alphas, last_ids,   \
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
