// init_depth.hip -- monocular-depth initialisation kernels.
//
// Replaces, under /root/reference/gs_init_compare/:
//   depth_prediction/points_from_depth.py:111-180   SfM reprojection + validity   (B1)
//   depth_alignment/alignment/lstsqrs.py:9-54       scale/shift least squares     (B2)
//   depth_alignment/alignment/ransacs.py:100-189    RANSAC/MSAC hypothesis scoring(B3)
//   depth_subsampling/static_subsampler.py:8-22,
//   depth_subsampling/adaptive_subsampling.py:48-122  sampling masks              (B5/B6)
//   depth_subsampling/num_sfm_points_mask.py:38-64  patch-density mask            (B7)
//   depth_prediction/points_from_depth.py:192-212   depth-gradient magnitude      (B8)
//   depth_prediction/points_from_depth.py:270-312   mask + compaction + unproject (B9)
//
// All of it is HBM-trivial integer/byte work at 1080p (about 11 MB per image):
// the point of the kernels is to replace the reference's Python loops (2 500
// RANSAC iterations x ~10 launches, a 720-patch double loop, a 33 MB
// cartesian_prod per mask) by a handful of launches per image.
#include "common.h"

namespace gsr {

// ---------------------------------------------------------------- B1
__global__ void __launch_bounds__(256)
project_sfm_kernel(int M, const float *__restrict__ pts, const float *__restrict__ P, int W, int H,
                   const uint8_t *__restrict__ pred_mask, int64_t *__restrict__ coords,
                   float *__restrict__ depth_out, uint8_t *__restrict__ inbounds,
                   uint8_t *__restrict__ valid) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float X = pts[i * 3], Y = pts[i * 3 + 1], Z = pts[i * 3 + 2];
  float c[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    c[r] = P[r * 4 + 0] * X + P[r * 4 + 1] * Y + P[r * 4 + 2] * Z + P[r * 4 + 3];
  const float u = rintf(__fdiv_rn(c[0], c[2]));   // torch.round: half to even
  const float v = rintf(__fdiv_rn(c[1], c[2]));
  // non-finite or huge values cannot be in bounds
  const bool finite = (fabsf(u) < 1e9f) && (fabsf(v) < 1e9f);
  int64_t x = finite ? (int64_t)u : -1, y = finite ? (int64_t)v : -1;
  const bool inb = finite && x >= 0 && x < W && y >= 0 && y < H && c[2] >= 0.f;
  inbounds[i] = inb;
  if (!inb) x = y = 0;   // points_from_depth.py:131-135: zeroed so the mask can be indexed
  coords[i] = x;
  coords[M + i] = y;
  depth_out[i] = c[2];
  valid[i] = inb && pred_mask[y * W + x];
}

// ---------------------------------------------------------------- B2 / B3
// sums[t][5] = {sum d^2, sum d, count, sum d*g, sum g} over a subset of the M
// correspondences. mode 0: all; mode 1: the S sample indices of hypothesis t;
// mode 2: inliers of hypothesis t ((s*d+t-g)^2 < thr). One block per t.
__global__ void __launch_bounds__(256)
lsq_sums_kernel(int T, int M, int mode, const float *__restrict__ d, const float *__restrict__ g,
                const int64_t *__restrict__ sample_idx, int S, const float *__restrict__ hyp,
                float thr, double *__restrict__ sums) {
  const int t = blockIdx.x;
  double a[5] = {0, 0, 0, 0, 0};
  if (mode == 1) {
    for (int k = threadIdx.x; k < S; k += blockDim.x) {
      const int64_t j = sample_idx[(int64_t)t * S + k];
      const double dd = d[j], gg = g[j];
      a[0] += dd * dd; a[1] += dd; a[2] += 1.0; a[3] += dd * gg; a[4] += gg;
    }
  } else {
    float hs = 0.f, ht = 0.f;
    if (mode == 2) { hs = hyp[t * 2]; ht = hyp[t * 2 + 1]; }
    for (int j = threadIdx.x; j < M; j += blockDim.x) {
      const float df = d[j], gf = g[j];
      if (mode == 2) {
        const float r = __fsub_rn(__fadd_rn(__fmul_rn(hs, df), ht), gf);
        if (!(__fmul_rn(r, r) < thr)) continue;
      }
      const double dd = df, gg = gf;
      a[0] += dd * dd; a[1] += dd; a[2] += 1.0; a[3] += dd * gg; a[4] += gg;
    }
  }
  __shared__ double red[4][5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double s = wave_sum_f64(a[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    sums[(int64_t)t * 5 + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
  }
}

// h = pinv([[Sdd, Sd],[Sd, n]]) @ [Sdg, Sg]  (lstsqrs.py:22-25), in fp64 with
// torch.linalg.pinv's default cut-off (singular values <= 2*eps_f32*max dropped).
__global__ void __launch_bounds__(256)
solve_scale_shift_kernel(int T, const double *__restrict__ sums, float *__restrict__ hyp) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const double a = sums[t * 5 + 0], b = sums[t * 5 + 1], c = sums[t * 5 + 2];
  const double r0 = sums[t * 5 + 3], r1 = sums[t * 5 + 4];
  // eigen-decomposition of the symmetric PSD 2x2
  const double tr = a + c, df = a - c;
  const double disc = sqrt(df * df + 4.0 * b * b);
  const double l1 = 0.5 * (tr + disc), l2 = 0.5 * (tr - disc);
  double v1x, v1y;
  if (fabs(b) > 0.0) { v1x = l1 - c; v1y = b; }
  else if (a >= c) { v1x = 1.0; v1y = 0.0; }
  else { v1x = 0.0; v1y = 1.0; }
  const double nrm = sqrt(v1x * v1x + v1y * v1y);
  if (nrm > 0.0) { v1x /= nrm; v1y /= nrm; }
  const double v2x = -v1y, v2y = v1x;
  const double cut = 2.0 * 1.1920928955078125e-07 * fmax(fabs(l1), fabs(l2));
  const double i1 = (fabs(l1) > cut) ? 1.0 / l1 : 0.0;
  const double i2 = (fabs(l2) > cut) ? 1.0 / l2 : 0.0;
  const double p1 = v1x * r0 + v1y * r1, p2 = v2x * r0 + v2y * r1;
  hyp[t * 2 + 0] = (float)(i1 * p1 * v1x + i2 * p2 * v2x);
  hyp[t * 2 + 1] = (float)(i1 * p1 * v1y + i2 * p2 * v2y);
}

// Score T hypotheses against all M correspondences: one block per hypothesis.
__global__ void __launch_bounds__(256)
ransac_score_kernel(int T, int M, const float *__restrict__ hyp, const float *__restrict__ d,
                    const float *__restrict__ g, float thr, int32_t *__restrict__ out_outliers,
                    float *__restrict__ out_msac, int32_t *__restrict__ out_inliers) {
  const int t = blockIdx.x;
  const float hs = hyp[t * 2], ht = hyp[t * 2 + 1];
  int n_out = 0, n_in = 0;
  double msac = 0.0;
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    // (h0*depth + h1 - gt)**2 in fp32, op by op (ransacs.py:94-97)
    const float r = __fsub_rn(__fadd_rn(__fmul_rn(hs, d[j]), ht), g[j]);
    const float r2 = __fmul_rn(r, r);
    n_out += (r2 >= thr) ? 1 : 0;
    n_in += (r2 < thr) ? 1 : 0;
    msac += (double)fminf(r2, thr);
  }
  __shared__ int red_o[4], red_i[4];
  __shared__ double red_m[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  n_out = wave_sum_i32(n_out);
  n_in = wave_sum_i32(n_in);
  msac = wave_sum_f64(msac);
  if (lane == 0) { red_o[wave] = n_out; red_i[wave] = n_in; red_m[wave] = msac; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out_outliers[t] = red_o[0] + red_o[1] + red_o[2] + red_o[3];
    out_inliers[t] = red_i[0] + red_i[1] + red_i[2] + red_i[3];
    out_msac[t] = (float)(red_m[0] + red_m[1] + red_m[2] + red_m[3]);
  }
}

__global__ void __launch_bounds__(256)
gather_depth_kernel(int M, const float *__restrict__ depth_map, int W,
                    const int64_t *__restrict__ coords, float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) out[i] = depth_map[coords[M + i] * W + coords[i]];
}

// aligned = depth * scale + shift, two rounded fp32 ops like torch (lstsqrs.py:52)
__global__ void __launch_bounds__(256)
affine_depth_kernel(int64_t n, const float *__restrict__ depth, const float *__restrict__ hs,
                    float *__restrict__ out) {
  const float s = hs[0], t = hs[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = __fadd_rn(__fmul_rn(depth[i], s), t);
}

// ---------------------------------------------------------------- B5 / B6
__global__ void __launch_bounds__(256)
subsample_mask_kernel(int H, int W, int mode, int static_k, const float *__restrict__ depth,
                      const uint8_t *__restrict__ valid, const float *__restrict__ range, int fmin,
                      int fmax, uint8_t *__restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  const bool ok = valid[i] != 0;
  int f = static_k;
  if (mode == 1) {
    // adaptive_subsampling.py:89-98, 102-117, op by op in fp32
    const float lo = range[0], hi = range[1];
    float m = __fdiv_rn(__fsub_rn(depth[i], lo), __fsub_rn(hi, lo));
    m = fminf(fmaxf(m, 0.f), 1.f);
    if (!ok) m = 0.5f;
    m = __fsub_rn(1.0f, m);
    float ff = __fadd_rn(__fmul_rn((float)(fmax - fmin), m), (float)fmin);
    ff = fminf(fmaxf(ff, (float)fmin), (float)fmax);
    f = (int)ff;                       // .to(int): truncation
    if (f == 0) f = 1;
  }
  keep[i] = ok && (y % f == 0) && (x % f == 0);
}

// ---------------------------------------------------------------- B7
__global__ void __launch_bounds__(256)
patch_hist_kernel(int M, const int64_t *__restrict__ coords, int ph, int pw, int gh, int gw,
                  int32_t *__restrict__ counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int64_t x = coords[i], y = coords[M + i];
  if (x < 0 || y < 0) return;
  const int64_t pj = x / pw, pi = y / ph;
  if (pi < gh && pj < gw) atomicAdd(&counts[pi * gw + pj], 1);
}
__global__ void __launch_bounds__(256)
patch_mask_kernel(int H, int W, int ph, int pw, int gh, int gw, int threshold,
                  const int32_t *__restrict__ counts, uint8_t *__restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  const int pi = y / ph, pj = x / pw;
  bool m = true;
  if (pi < gh && pj < gw) m = !(counts[pi * gw + pj] > threshold);
  mask[i] = m;
}

// ---------------------------------------------------------------- B8
__global__ void __launch_bounds__(256)
depth_grad_kernel(int H, int W, const float *__restrict__ depth, float *__restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  float gsum = 0.f;
  if (x > 0) gsum = __fadd_rn(gsum, fabsf(__fsub_rn(depth[i], depth[i - 1])));
  if (y > 0) gsum = __fadd_rn(gsum, fabsf(__fsub_rn(depth[i], depth[i - W])));
  grad[i] = gsum;
}

// ---------------------------------------------------------------- B9
constexpr int UNPROJ_BLOCK = 1024;   // pixels per compaction block (256 threads x 4)

__device__ __forceinline__ bool keep_pixel(int64_t i, int64_t n, const float *depth,
                                           const uint8_t *valid, const uint8_t *subsample,
                                           const uint8_t *extra) {
  if (i >= n) return false;
  bool k = valid[i] && subsample[i] && (depth[i] >= 0.f);
  if (extra) k = k && extra[i];
  return k;
}

__global__ void __launch_bounds__(256)
unproject_count_kernel(int64_t n, const float *__restrict__ depth,
                       const uint8_t *__restrict__ valid, const uint8_t *__restrict__ subsample,
                       const uint8_t *__restrict__ extra, int32_t *__restrict__ block_counts) {
  const int64_t base = (int64_t)blockIdx.x * UNPROJ_BLOCK + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) c += keep_pixel(base + k, n, depth, valid, subsample, extra) ? 1 : 0;
  __shared__ int red[4];
  c = wave_sum_i32(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256)
unproject_emit_kernel(int H, int W, const float *__restrict__ depth,
                      const uint8_t *__restrict__ valid, const uint8_t *__restrict__ subsample,
                      const uint8_t *__restrict__ extra, const float *__restrict__ rgb,
                      const float *__restrict__ Kinv, const float *__restrict__ c2w,
                      const int32_t *__restrict__ block_offsets, float *__restrict__ pts,
                      float *__restrict__ rgb_out, uint8_t *__restrict__ final_mask) {
  const int64_t n = (int64_t)H * W;
  const int64_t base = (int64_t)blockIdx.x * UNPROJ_BLOCK + threadIdx.x * 4;
  bool k[4];
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    k[j] = keep_pixel(base + j, n, depth, valid, subsample, extra);
    c += k[j] ? 1 : 0;
    if (final_mask && base + j < n) final_mask[base + j] = k[j];
  }
  // exclusive scan of c over the 256 threads (pixel order is thread order)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  __shared__ int wtot[4];
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int pos = block_offsets[blockIdx.x] + incl - c;
  for (int w = 0; w < wave; ++w) pos += wtot[w];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!k[j]) continue;
    const int64_t i = base + j;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const float z = depth[i];
    // points_from_depth.py:299-300, 302-310
    const float cx = __fmul_rn(__fadd_rn((float)x, 0.5f), z);
    const float cy = __fmul_rn(__fadd_rn((float)y, 0.5f), z);
    float dcam[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      dcam[r] = fmaf(Kinv[r * 3 + 2], z, fmaf(Kinv[r * 3 + 1], cy, Kinv[r * 3] * cx));
#pragma unroll
    for (int r = 0; r < 3; ++r)
      pts[(int64_t)pos * 3 + r] =
          fmaf(c2w[r * 4 + 2], dcam[2], fmaf(c2w[r * 4 + 1], dcam[1], c2w[r * 4] * dcam[0])) +
          c2w[r * 4 + 3];
    if (rgb_out) {
      rgb_out[(int64_t)pos * 3 + 0] = rgb[i * 3 + 0];
      rgb_out[(int64_t)pos * 3 + 1] = rgb[i * 3 + 1];
      rgb_out[(int64_t)pos * 3 + 2] = rgb[i * 3 + 2];
    }
    ++pos;
  }
}

// ---------------------------------------------------------------- B10
// Metric3D pre-processing (predictors/metric3d.py:42-83) fused into one pass:
// float RGB [H,W,3] in [0,1] -> uint8 (truncation) -> channel flip ([:, :, ::-1]) ->
// keep-ratio bilinear resize to (rh, rw) (half-pixel centres, result rounded to
// uint8 like cv2.INTER_LINEAR) -> constant border (the mean colour, saturated to
// uint8) up to (out_h, out_w) -> (x - mean) / std -> planar [3, out_h, out_w].
__global__ void __launch_bounds__(256)
m3d_preprocess_kernel(int H, int W, const float *__restrict__ img, int rh, int rw, int pad_top,
                      int pad_left, int out_h, int out_w, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)out_h * out_w) return;
  const int oy = (int)(i / out_w), ox = (int)(i - (int64_t)oy * out_w);
  const float mean[3] = {123.675f, 116.28f, 103.53f}, stdv[3] = {58.395f, 57.12f, 57.375f};
  const float border[3] = {124.f, 116.f, 104.f};   // cv2 saturate_cast<uchar> of the pad value
  const int ry = oy - pad_top, rx = ox - pad_left;
  float px[3];
  if (ry < 0 || rx < 0 || ry >= rh || rx >= rw) {
    px[0] = border[0]; px[1] = border[1]; px[2] = border[2];
  } else {
    const float sy = fmaxf(((float)ry + 0.5f) * ((float)H / (float)rh) - 0.5f, 0.f);
    const float sx = fmaxf(((float)rx + 0.5f) * ((float)W / (float)rw) - 0.5f, 0.f);
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sc = 2 - c;   // [:, :, ::-1]
      auto q = [&](int y, int x) { return floorf(img[((int64_t)y * W + x) * 3 + sc] * 255.0f); };
      const float top = q(y0, x0) + (q(y0, x1) - q(y0, x0)) * fx;
      const float bot = q(y1, x0) + (q(y1, x1) - q(y1, x0)) * fx;
      px[c] = fminf(fmaxf(rintf(top + (bot - top) * fy), 0.f), 255.f);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[(int64_t)c * out_h * out_w + i] = (px[c] - mean[c]) / stdv[c];
}

// Metric3D post-processing (metric3d.py:96-131) fused: un-pad, bilinear upsample
// (F.interpolate, align_corners=False) to (H, W), * scale, clamp [lo, hi].
__global__ void __launch_bounds__(256)
m3d_postprocess_kernel(int in_h, int in_w, const float *__restrict__ in, int pad_top, int pad_bot,
                       int pad_left, int pad_right, int H, int W, float scale, float lo, float hi,
                       int do_clamp, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  const int ch = in_h - pad_top - pad_bot, cw = in_w - pad_left - pad_right;
  const float sy = fmaxf(((float)y + 0.5f) * ((float)ch / (float)H) - 0.5f, 0.f);
  const float sx = fmaxf(((float)x + 0.5f) * ((float)cw / (float)W) - 0.5f, 0.f);
  const int y0 = min((int)sy, ch - 1), x0 = min((int)sx, cw - 1);
  const int y1 = min(y0 + 1, ch - 1), x1 = min(x0 + 1, cw - 1);
  const float fy = sy - (float)y0, fx = sx - (float)x0;
  auto q = [&](int yy, int xx) { return in[(int64_t)(yy + pad_top) * in_w + (xx + pad_left)]; };
  const float v = (1.f - fy) * ((1.f - fx) * q(y0, x0) + fx * q(y0, x1)) +
                  fy * ((1.f - fx) * q(y1, x0) + fx * q(y1, x1));
  float r = v * scale;
  if (do_clamp) r = fminf(fmaxf(r, lo), hi);
  out[i] = r;
}

}  // namespace gsr

#define ST ((hipStream_t)stream)

extern "C" int gsr_project_sfm(int M, const float *pts, const float *P, int W, int H,
                               const uint8_t *pred_mask, int64_t *coords, float *depth_out,
                               uint8_t *inbounds, uint8_t *valid, void *stream) {
  GSR_REQUIRE(M >= 0 && W > 0 && H > 0, "project_sfm: bad sizes");
  if (M == 0) return GSR_OK;
  GSR_REQUIRE(pts && P && pred_mask && coords && depth_out && inbounds && valid,
              "project_sfm: null pointer");
  hipLaunchKernelGGL(gsr::project_sfm_kernel, dim3(gsr::ceil_div(M, 256)), dim3(256), 0, ST, M,
                     pts, P, W, H, pred_mask, coords, depth_out, inbounds, valid);
  GSR_CHECK_LAUNCH("project_sfm");
  return GSR_OK;
}

extern "C" int gsr_gather_depth(int M, const float *depth_map, int depth_w, const int64_t *coords,
                                float *out, void *stream) {
  GSR_REQUIRE(M >= 0 && depth_w > 0, "gather_depth: bad sizes");
  if (M == 0) return GSR_OK;
  GSR_REQUIRE(depth_map && coords && out, "gather_depth: null pointer");
  hipLaunchKernelGGL(gsr::gather_depth_kernel, dim3(gsr::ceil_div(M, 256)), dim3(256), 0, ST, M,
                     depth_map, depth_w, coords, out);
  GSR_CHECK_LAUNCH("gather_depth");
  return GSR_OK;
}

extern "C" int gsr_lsq_sums(int T, int M, int mode, const float *d, const float *g,
                            const int64_t *sample_idx, int S, const float *hyp, float thr,
                            double *sums_out, void *stream) {
  GSR_REQUIRE(T >= 0 && M >= 0 && mode >= 0 && mode <= 2, "lsq_sums: bad arguments");
  if (T == 0) return GSR_OK;
  GSR_REQUIRE(d && g && sums_out, "lsq_sums: null pointer");
  GSR_REQUIRE(mode != 1 || (sample_idx && S > 0), "lsq_sums: mode 1 needs sample_idx");
  GSR_REQUIRE(mode != 2 || hyp, "lsq_sums: mode 2 needs hypotheses");
  hipLaunchKernelGGL(gsr::lsq_sums_kernel, dim3(T), dim3(256), 0, ST, T, M, mode, d, g,
                     sample_idx, S, hyp, thr, sums_out);
  GSR_CHECK_LAUNCH("lsq_sums");
  return GSR_OK;
}

extern "C" int gsr_solve_scale_shift(int T, const double *sums, float *hyp, void *stream) {
  GSR_REQUIRE(T >= 0, "solve_scale_shift: bad T");
  if (T == 0) return GSR_OK;
  GSR_REQUIRE(sums && hyp, "solve_scale_shift: null pointer");
  hipLaunchKernelGGL(gsr::solve_scale_shift_kernel, dim3(gsr::ceil_div(T, 256)), dim3(256), 0, ST,
                     T, sums, hyp);
  GSR_CHECK_LAUNCH("solve_scale_shift");
  return GSR_OK;
}

extern "C" int gsr_ransac_score(int T, int M, const float *hyp, const float *d, const float *g,
                                float thr, int32_t *out_ransac, float *out_msac,
                                int32_t *out_inliers, void *stream) {
  GSR_REQUIRE(T >= 0 && M >= 0, "ransac_score: bad sizes");
  if (T == 0) return GSR_OK;
  GSR_REQUIRE(hyp && d && g && out_ransac && out_msac && out_inliers, "ransac_score: null pointer");
  hipLaunchKernelGGL(gsr::ransac_score_kernel, dim3(T), dim3(256), 0, ST, T, M, hyp, d, g, thr,
                     out_ransac, out_msac, out_inliers);
  GSR_CHECK_LAUNCH("ransac_score");
  return GSR_OK;
}

extern "C" int gsr_affine_depth(int64_t n, const float *depth, const float *hs, float *out,
                                void *stream) {
  GSR_REQUIRE(n >= 0, "affine_depth: bad n");
  if (n == 0) return GSR_OK;
  GSR_REQUIRE(depth && hs && out, "affine_depth: null pointer");
  int blocks = (int)(gsr::ceil_div64(n, 256) < 8192 ? gsr::ceil_div64(n, 256) : 8192);
  hipLaunchKernelGGL(gsr::affine_depth_kernel, dim3(blocks), dim3(256), 0, ST, n, depth, hs, out);
  GSR_CHECK_LAUNCH("affine_depth");
  return GSR_OK;
}

extern "C" int gsr_subsample_mask(int H, int W, int mode, int static_k, const float *depth,
                                  const uint8_t *valid, const float *range, int fmin, int fmax,
                                  uint8_t *keep, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && (mode == 0 || mode == 1), "subsample_mask: bad arguments");
  GSR_REQUIRE(valid && keep, "subsample_mask: null pointer");
  GSR_REQUIRE(mode == 1 || static_k > 0, "subsample_mask: static factor must be > 0");
  GSR_REQUIRE(mode == 0 || (depth && range), "subsample_mask: adaptive needs depth + range");
  int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(gsr::subsample_mask_kernel, dim3((unsigned)gsr::ceil_div64(n, 256)),
                     dim3(256), 0, ST, H, W, mode, static_k, depth, valid, range, fmin, fmax, keep);
  GSR_CHECK_LAUNCH("subsample_mask");
  return GSR_OK;
}

extern "C" int gsr_sfm_patch_mask(int H, int W, int M, const int64_t *coords, int ph, int pw,
                                  int gh, int gw, int threshold, int32_t *patch_counts,
                                  uint8_t *mask, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && M >= 0 && ph > 0 && pw > 0 && gh > 0 && gw > 0,
              "sfm_patch_mask: bad arguments");
  GSR_REQUIRE(patch_counts && mask && (M == 0 || coords), "sfm_patch_mask: null pointer");
  GSR_CHECK_HIP(hipMemsetAsync(patch_counts, 0, sizeof(int32_t) * gh * gw, ST));
  if (M > 0) {
    hipLaunchKernelGGL(gsr::patch_hist_kernel, dim3(gsr::ceil_div(M, 256)), dim3(256), 0, ST, M,
                       coords, ph, pw, gh, gw, patch_counts);
    GSR_CHECK_LAUNCH("patch_hist");
  }
  int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(gsr::patch_mask_kernel, dim3((unsigned)gsr::ceil_div64(n, 256)), dim3(256), 0,
                     ST, H, W, ph, pw, gh, gw, threshold, patch_counts, mask);
  GSR_CHECK_LAUNCH("patch_mask");
  return GSR_OK;
}

extern "C" int gsr_depth_grad(int H, int W, const float *depth, float *grad, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && depth && grad, "depth_grad: bad arguments");
  int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(gsr::depth_grad_kernel, dim3((unsigned)gsr::ceil_div64(n, 256)), dim3(256), 0,
                     ST, H, W, depth, grad);
  GSR_CHECK_LAUNCH("depth_grad");
  return GSR_OK;
}

extern "C" int gsr_unproject_num_blocks(int H, int W) {
  return (int)gsr::ceil_div64((int64_t)H * W, gsr::UNPROJ_BLOCK);
}

extern "C" int gsr_unproject_count(int H, int W, const float *depth, const uint8_t *valid,
                                   const uint8_t *subsample, const uint8_t *extra,
                                   int32_t *block_counts, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && depth && valid && subsample && block_counts,
              "unproject_count: bad arguments");
  hipLaunchKernelGGL(gsr::unproject_count_kernel, dim3(gsr_unproject_num_blocks(H, W)), dim3(256),
                     0, ST, (int64_t)H * W, depth, valid, subsample, extra, block_counts);
  GSR_CHECK_LAUNCH("unproject_count");
  return GSR_OK;
}

extern "C" int gsr_unproject_emit(int H, int W, const float *depth, const uint8_t *valid,
                                  const uint8_t *subsample, const uint8_t *extra, const float *rgb,
                                  const float *Kinv, const float *c2w,
                                  const int32_t *block_offsets, float *pts, float *rgb_out,
                                  uint8_t *final_mask, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && depth && valid && subsample && Kinv && c2w && block_offsets,
              "unproject_emit: bad arguments");
  GSR_REQUIRE(!rgb_out || rgb, "unproject_emit: rgb_out without rgb");
  hipLaunchKernelGGL(gsr::unproject_emit_kernel, dim3(gsr_unproject_num_blocks(H, W)), dim3(256),
                     0, ST, H, W, depth, valid, subsample, extra, rgb, Kinv, c2w, block_offsets,
                     pts, rgb_out, final_mask);
  GSR_CHECK_LAUNCH("unproject_emit");
  return GSR_OK;
}

extern "C" int gsr_m3d_preprocess(int H, int W, const float *img, int rh, int rw, int pad_top,
                                  int pad_left, int out_h, int out_w, float *out, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && rh > 0 && rw > 0 && out_h >= rh + pad_top && out_w >= rw + pad_left &&
                  pad_top >= 0 && pad_left >= 0,
              "m3d_preprocess: bad sizes");
  GSR_REQUIRE(img && out, "m3d_preprocess: null pointer");
  int64_t n = (int64_t)out_h * out_w;
  hipLaunchKernelGGL(gsr::m3d_preprocess_kernel, dim3((unsigned)gsr::ceil_div64(n, 256)),
                     dim3(256), 0, ST, H, W, img, rh, rw, pad_top, pad_left, out_h, out_w, out);
  GSR_CHECK_LAUNCH("m3d_preprocess");
  return GSR_OK;
}

extern "C" int gsr_m3d_postprocess(int in_h, int in_w, const float *in, int pad_top, int pad_bot,
                                   int pad_left, int pad_right, int H, int W, float scale,
                                   float lo, float hi, int do_clamp, float *out, void *stream) {
  GSR_REQUIRE(in_h > pad_top + pad_bot && in_w > pad_left + pad_right && H > 0 && W > 0 &&
                  pad_top >= 0 && pad_bot >= 0 && pad_left >= 0 && pad_right >= 0,
              "m3d_postprocess: bad sizes");
  GSR_REQUIRE(in && out, "m3d_postprocess: null pointer");
  int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(gsr::m3d_postprocess_kernel, dim3((unsigned)gsr::ceil_div64(n, 256)),
                     dim3(256), 0, ST, in_h, in_w, in, pad_top, pad_bot, pad_left, pad_right, H, W,
                     scale, lo, hi, do_clamp, out);
  GSR_CHECK_LAUNCH("m3d_postprocess");
  return GSR_OK;
}

// ---- F4 (tail): piecewise-linear interpolation over a Delaunay triangulation ------------------
// Replaces scipy.interpolate.LinearNDInterpolator(dt, values)(X, Y) on the full pixel grid
// (depth_alignment/alignment/interp.py:77-110: ~2 M queries on the CPU in the reference). The
// triangulation itself (a few thousand SfM points) stays a host call to scipy, as in the
// reference; here one thread block per triangle walks the integer pixels of its bounding box
// and writes the barycentric interpolation (fp64, like scipy) for those inside or on the edge.
// Pixels on a shared edge are written by both neighbours with values equal to rounding.
namespace gsr {
__global__ void __launch_bounds__(64)
tri_interp_kernel(int H, int W, int n_tri, const double *__restrict__ xy, const int32_t *__restrict__ tris,
                  const double *__restrict__ values, float *__restrict__ out) {
  const int t = blockIdx.x;
  if (t >= n_tri) return;
  const int i0 = tris[t * 3], i1 = tris[t * 3 + 1], i2 = tris[t * 3 + 2];
  const double x0 = xy[i0 * 2], y0 = xy[i0 * 2 + 1], x1 = xy[i1 * 2], y1 = xy[i1 * 2 + 1];
  const double x2 = xy[i2 * 2], y2 = xy[i2 * 2 + 1];
  const double det = (y1 - y2) * (x0 - x2) + (x2 - x1) * (y0 - y2);
  if (det == 0.0) return;                                  // degenerate (collinear) simplex
  const double v0 = values[i0], v1 = values[i1], v2 = values[i2];
  const int bx0 = max(0, (int)ceil(fmin(x0, fmin(x1, x2)))), bx1 = min(W - 1, (int)floor(fmax(x0, fmax(x1, x2))));
  const int by0 = max(0, (int)ceil(fmin(y0, fmin(y1, y2)))), by1 = min(H - 1, (int)floor(fmax(y0, fmax(y1, y2))));
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  if (bw <= 0 || bh <= 0) return;
  const double eps = 1e-10;
  for (int k = threadIdx.x; k < bw * bh; k += blockDim.x) {
    const int px = bx0 + k % bw, py = by0 + k / bw;
    const double l0 = ((y1 - y2) * (px - x2) + (x2 - x1) * (py - y2)) / det;
    const double l1 = ((y2 - y0) * (px - x2) + (x0 - x2) * (py - y2)) / det;
    const double l2 = 1.0 - l0 - l1;
    if (l0 >= -eps && l1 >= -eps && l2 >= -eps) out[(int64_t)py * W + px] = (float)(l0 * v0 + l1 * v1 + l2 * v2);
  }
}
}  // namespace gsr

extern "C" int gsr_tri_interp(int H, int W, int n_tri, const double *xy, const int32_t *tris,
                              const double *values, float *out, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && n_tri >= 0, "tri_interp: bad sizes");
  if (n_tri == 0) return GSR_OK;
  GSR_REQUIRE(xy && tris && values && out, "tri_interp: null pointer");
  hipLaunchKernelGGL(gsr::tri_interp_kernel, dim3((unsigned)n_tri), dim3(64), 0, (hipStream_t)stream, H, W,
                     n_tri, xy, tris, values, out);
  GSR_CHECK_LAUNCH("tri_interp");
  return GSR_OK;
}

// ---- F4 tail: region margin mask (depth_alignment/segmentation/region_margin.py:21-35) --------
// The reference box-blurs the label map (replicate padding, k = 2m+1) in fp32, snaps values that
// are `isclose` to an integer, and keeps the pixels whose blurred label equals their own: a pixel
// is "interior" when its k x k window averages to its own label. Here the window SUM is exact
// integer arithmetic (separable: row sums, then column sums, both with clamped coordinates =
// replicate padding); the mean is the correctly rounded fp32 of S / k^2 and goes through the same
// snap rule (|x - round(x)| <= 1e-8 + 1e-5 |round(x)|, torch.isclose's defaults), so that labels
// large enough for the tolerance to exceed 1/k^2 behave as in the reference.
namespace gsr {
__global__ void __launch_bounds__(256)
region_rowsum_kernel(int H, int W, int m, const int32_t *__restrict__ labels, int64_t *__restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  const int32_t *row = labels + (int64_t)y * W;
  int64_t s = 0;
  for (int d = -m; d <= m; ++d) s += row[min(max(x + d, 0), W - 1)];
  rows[i] = s;
}

__global__ void __launch_bounds__(256)
region_margin_kernel(int H, int W, int m, const int32_t *__restrict__ labels,
                     const int64_t *__restrict__ rows, uint8_t *__restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  int64_t s = 0;
  for (int d = -m; d <= m; ++d) s += rows[(int64_t)min(max(y + d, 0), H - 1) * W + x];
  const int k = 2 * m + 1;
  const float mean = (float)((double)s / (double)((int64_t)k * k));
  const float nearest = rintf(mean);                                  // torch.round: half to even
  const bool close = fabsf(mean - nearest) <= 1e-8f + 1e-5f * fabsf(nearest);
  mask[i] = ((close ? nearest : mean) == (float)labels[i]) ? 1 : 0;
}
}  // namespace gsr

extern "C" int gsr_region_margin_mask(int H, int W, int half_width, const int32_t *labels,
                                      int64_t *row_sums, uint8_t *mask, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && half_width >= 0, "region_margin_mask: bad sizes");
  GSR_REQUIRE(labels && row_sums && mask, "region_margin_mask: null pointer");
  const unsigned nb = (unsigned)gsr::ceil_div64((int64_t)H * W, 256);
  hipLaunchKernelGGL(gsr::region_rowsum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, H, W, half_width,
                     labels, row_sums);
  hipLaunchKernelGGL(gsr::region_margin_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, H, W, half_width,
                     labels, row_sums, mask);
  GSR_CHECK_LAUNCH("region_margin_mask");
  return GSR_OK;
}
