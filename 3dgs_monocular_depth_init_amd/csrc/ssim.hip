// ssim.hip -- F1: fused L1 + SSIM training loss (forward and backward).
//
// Replaces, in the step body of gs_init_compare/runner.py:506-510,
//   l1loss   = F.l1_loss(colors, pixels)
//   ssimloss = 1 - fused_ssim(colors.permute(0,3,1,2), pixels.permute(0,3,1,2), padding="valid")
//   loss     = l1loss * (1 - ssim_lambda) + ssimloss * ssim_lambda
// (`fused_ssim` = rahul-goel/fused-ssim @30fb258, a third-party CUDA op pinned at
// setup.py:14: 11x11 Gaussian window, sigma 1.5, C1 = 0.01^2, C2 = 0.03^2, zero
// padding for the window, "valid" = mean over the map cropped by 5 pixels).
//
// HBM-bound: two passes over two images. Each workgroup owns a 32x32 output
// tile of one (image, channel) plane; the 42x42 input halo tile is staged in
// LDS and the 11-tap window is applied separably (rows, then columns) from LDS.
// Images are addressed with element strides, so the rasterizer's NHWC output is
// consumed in place (no permute().contiguous() copies).
#include "common.h"

namespace gsr {

constexpr int SSIM_R = 5;                 // window radius (11 taps)
constexpr int SSIM_T = 32;                // output tile edge
constexpr int SSIM_H = SSIM_T + 2 * SSIM_R;   // halo tile edge (42)
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

// exp(-(i-5)^2 / (2*1.5^2)) normalised to sum 1
__constant__ float SSIM_G[11] = {0.001028380123898387f, 0.0075987582094967365f,
                                 0.036000773310661316f, 0.10936068743467331f,
                                 0.21300552785396576f,  0.26601171493530273f,
                                 0.21300552785396576f,  0.10936068743467331f,
                                 0.036000773310661316f, 0.0075987582094967365f,
                                 0.001028380123898387f};

struct ImgView {          // element strides of an [N, CH, H, W] logical image
  int64_t sn, sc, sh, sw;
};

__device__ __forceinline__ float ld(const float *p, const ImgView &v, int n, int c, int y, int x,
                                    int H, int W) {
  if (x < 0 || y < 0 || x >= W || y >= H) return 0.f;
  return p[n * v.sn + c * v.sc + y * v.sh + x * v.sw];
}

// Forward: per-pixel SSIM (+ L1) summed into sums[0] (ssim over the counted
// region) and sums[1] (sum |x-y| over all pixels); optionally the three
// derivative maps for the backward.
__global__ void __launch_bounds__(256)
ssim_fwd_kernel(int N, int CH, int H, int W, const float *__restrict__ img1, ImgView v1,
                const float *__restrict__ img2, ImgView v2, int valid_only,
                double *__restrict__ ws, float *__restrict__ dm_mu1,
                float *__restrict__ dm_s1, float *__restrict__ dm_s12) {
  __shared__ float sX[SSIM_H][SSIM_H + 1];
  __shared__ float sY[SSIM_H][SSIM_H + 1];
  __shared__ float sHz[5][SSIM_H][SSIM_T + 1];   // row-filtered: x, y, xx, yy, xy
  __shared__ double red[2][4];
  const int tid = threadIdx.x;
  const int tiles_x = (W + SSIM_T - 1) / SSIM_T;
  const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
  const int plane = blockIdx.y, n = plane / CH, c = plane % CH;
  const int x0 = bx * SSIM_T - SSIM_R, y0 = by * SSIM_T - SSIM_R;

  for (int i = tid; i < SSIM_H * SSIM_H; i += 256) {
    const int ly = i / SSIM_H, lx = i % SSIM_H;
    sX[ly][lx] = ld(img1, v1, n, c, y0 + ly, x0 + lx, H, W);
    sY[ly][lx] = ld(img2, v2, n, c, y0 + ly, x0 + lx, H, W);
  }
  __syncthreads();
  for (int i = tid; i < SSIM_H * SSIM_T; i += 256) {
    const int ly = i / SSIM_T, lx = i % SSIM_T;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = SSIM_G[k], x = sX[ly][lx + k], y = sY[ly][lx + k];
      a = fmaf(g, x, a);
      b = fmaf(g, y, b);
      aa = fmaf(g, x * x, aa);
      bb = fmaf(g, y * y, bb);
      ab = fmaf(g, x * y, ab);
    }
    sHz[0][ly][lx] = a;
    sHz[1][ly][lx] = b;
    sHz[2][ly][lx] = aa;
    sHz[3][ly][lx] = bb;
    sHz[4][ly][lx] = ab;
  }
  __syncthreads();
  double acc_ssim = 0.0, acc_l1 = 0.0;
  for (int i = tid; i < SSIM_T * SSIM_T; i += 256) {
    const int ly = i / SSIM_T, lx = i % SSIM_T;
    const int gx = bx * SSIM_T + lx, gy = by * SSIM_T + ly;
    if (gx >= W || gy >= H) continue;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = SSIM_G[k];
      mu1 = fmaf(g, sHz[0][ly + k][lx], mu1);
      mu2 = fmaf(g, sHz[1][ly + k][lx], mu2);
      e11 = fmaf(g, sHz[2][ly + k][lx], e11);
      e22 = fmaf(g, sHz[3][ly + k][lx], e22);
      e12 = fmaf(g, sHz[4][ly + k][lx], e12);
    }
    const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
    const float A = mu1 * mu1 + mu2 * mu2 + SSIM_C1, B = s1 + s2 + SSIM_C2;
    const float Cc = 2.f * mu1 * mu2 + SSIM_C1, D = 2.f * s12 + SSIM_C2;
    const float iAB = 1.0f / (A * B);
    const float m = Cc * D * iAB;
    const bool counted = !valid_only || (gx >= SSIM_R && gx < W - SSIM_R && gy >= SSIM_R && gy < H - SSIM_R);
    if (counted) acc_ssim += (double)m;
    acc_l1 += (double)fabsf(sX[ly + SSIM_R][lx + SSIM_R] - sY[ly + SSIM_R][lx + SSIM_R]);
    if (dm_mu1) {
      const int64_t o = ((int64_t)plane * H + gy) * W + gx;
      const float w = counted ? 1.f : 0.f;   // cropped pixels get no gradient
      dm_mu1[o] = w * (2.f * mu2 * D * iAB - 2.f * mu2 * Cc * iAB - 2.f * mu1 * Cc * D * iAB / A +
                       2.f * mu1 * Cc * D * iAB / B);
      dm_s1[o] = w * (-Cc * D * iAB / B);
      dm_s12[o] = w * (2.f * Cc * iAB);
    }
  }
  acc_ssim = wave_sum_f64(acc_ssim);
  acc_l1 = wave_sum_f64(acc_l1);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = acc_ssim;
    red[1][tid >> 6] = acc_l1;
  }
  __syncthreads();
  // One pair of partial sums per workgroup, plain stores: thousands of workgroups adding
  // into the same two addresses serialise at the memory side (measured: most of the kernel).
  if (tid == 0) {
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    ws[2 * wg + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ws[2 * wg + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// out[0] = mean SSIM, out[1] = mean |x-y|, out[2] = the reference's loss
// (1-lambda)*L1 + lambda*(1-SSIM) (runner.py:506-510) from the per-workgroup partials.
__global__ void __launch_bounds__(1024)
ssim_finalize_kernel(int n_wg, const double *__restrict__ ws, float *__restrict__ out,
                     double inv_n_ssim, double inv_n_l1, float ssim_lambda) {
  __shared__ double red[2][16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n_wg; i += 1024) {
    a += ws[2 * i];
    b += ws[2 * i + 1];
  }
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
    for (int w = 0; w < 16; ++w) {
      sa += red[0][w];
      sb += red[1][w];
    }
    const double ssim = sa * inv_n_ssim, l1 = sb * inv_n_l1;
    out[0] = (float)ssim;
    out[1] = (float)l1;
    out[2] = (float)(l1 * (1.0 - (double)ssim_lambda) + (1.0 - ssim) * (double)ssim_lambda);
  }
}

// Backward: grad_img1 = w_l1 * sign(x - y) + w_ssim * ( G*dm_mu1 + 2x (G*dm_s1) + y (G*dm_s12) )
// where w_ssim already carries d loss / d mean-ssim divided by the counted pixels.
__global__ void __launch_bounds__(256)
ssim_bwd_kernel(int N, int CH, int H, int W, const float *__restrict__ img1, ImgView v1,
                const float *__restrict__ img2, ImgView v2, const float *__restrict__ dm_mu1,
                const float *__restrict__ dm_s1, const float *__restrict__ dm_s12,
                const float *__restrict__ weights /* device [2]: w_ssim, w_l1, or NULL */,
                const float *__restrict__ upstream /* device scalar (with weights NULL) */,
                float scale_ssim, float scale_l1,
                float *__restrict__ grad, ImgView vg) {
  __shared__ float sM[3][SSIM_H][SSIM_H + 1];
  __shared__ float sHz[3][SSIM_H][SSIM_T + 1];
  const int tid = threadIdx.x;
  const int tiles_x = (W + SSIM_T - 1) / SSIM_T;
  const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
  const int plane = blockIdx.y, n = plane / CH, c = plane % CH;
  const int x0 = bx * SSIM_T - SSIM_R, y0 = by * SSIM_T - SSIM_R;
  const float up = upstream ? upstream[0] : 1.0f;
  const float w_ssim = weights ? weights[0] : up * scale_ssim;
  const float w_l1 = weights ? weights[1] : up * scale_l1;
  for (int i = tid; i < SSIM_H * SSIM_H; i += 256) {
    const int ly = i / SSIM_H, lx = i % SSIM_H;
    const int gx = x0 + lx, gy = y0 + ly;
    float a = 0.f, b = 0.f, d = 0.f;
    if (gx >= 0 && gy >= 0 && gx < W && gy < H) {
      const int64_t o = ((int64_t)plane * H + gy) * W + gx;
      a = dm_mu1[o];
      b = dm_s1[o];
      d = dm_s12[o];
    }
    sM[0][ly][lx] = a;
    sM[1][ly][lx] = b;
    sM[2][ly][lx] = d;
  }
  __syncthreads();
  for (int i = tid; i < SSIM_H * SSIM_T; i += 256) {
    const int ly = i / SSIM_T, lx = i % SSIM_T;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = SSIM_G[k];
      a = fmaf(g, sM[0][ly][lx + k], a);
      b = fmaf(g, sM[1][ly][lx + k], b);
      d = fmaf(g, sM[2][ly][lx + k], d);
    }
    sHz[0][ly][lx] = a;
    sHz[1][ly][lx] = b;
    sHz[2][ly][lx] = d;
  }
  __syncthreads();
  for (int i = tid; i < SSIM_T * SSIM_T; i += 256) {
    const int ly = i / SSIM_T, lx = i % SSIM_T;
    const int gx = bx * SSIM_T + lx, gy = by * SSIM_T + ly;
    if (gx >= W || gy >= H) continue;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = SSIM_G[k];
      a = fmaf(g, sHz[0][ly + k][lx], a);
      b = fmaf(g, sHz[1][ly + k][lx], b);
      d = fmaf(g, sHz[2][ly + k][lx], d);
    }
    const float x = img1[n * v1.sn + c * v1.sc + gy * v1.sh + gx * v1.sw];
    const float y = img2[n * v2.sn + c * v2.sc + gy * v2.sh + gx * v2.sw];
    const float df = x - y;
    const float sgn = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
    grad[n * vg.sn + c * vg.sc + gy * vg.sh + gx * vg.sw] =
        w_l1 * sgn + w_ssim * (a + 2.f * x * b + y * d);
  }
}

// ---------------------------------------------------------------------------------------------------
// Round 4: sliding-window kernels (the launches gsr_ssim_l1_fwd / _bwd now make).
//
// The tile kernels above stage a 42 x 42 halo tile and BOTH separable passes through LDS (42 KB per
// workgroup: three workgroups per CU, three barriers each); at 1080p they took 0.121 + 0.083 ms where the
// images and maps they move are worth ~0.05 ms of HBM time. Here ONE WAVE owns a strip of 64 columns x
// SS_ROWS rows of one (image, channel) plane, lane = column, and walks it row by row:
//   * the row's 74 input values (64 + the 2 x 5 halo) go through a per-wave LDS row buffer -- the only
//     LDS traffic: 22 (forward) / 33 (backward) conflict-free broadcast-free reads per pixel;
//   * the horizontally filtered values of the last 11 rows live in REGISTERS (a window per quantity; the
//     row loop is unrolled by 11 so that every slot index is static), the vertical filter is 11 FMAs per
//     quantity on registers;
//   * no workgroup barrier, no shared tile: a wave's LDS operations execute in order, the row buffers are
//     double buffered, the next row's global loads are issued one row ahead.
// The 11-tap window is symmetric: taps k and 10 - k share a weight (6 multiplies per 11 taps).
constexpr int SS_COLS = 64;                        // one column per lane
#ifndef GSR_SS_ROWS
#define GSR_SS_ROWS 32
#endif
#ifndef GSR_SS_ROWS_FWD
#define GSR_SS_ROWS_FWD 24          // 1080 / 24 = 45 strips: 4 050 forward waves = 4 per SIMD (32 rows: 3 060 = 3 per SIMD;
                                    // forward 0.060 -> 0.056 ms; the backward, with fewer registers per wave, loses at 24)
#endif
constexpr int SS_ROWS = GSR_SS_ROWS;               // output rows per strip of the backward (+ 10 halo rows walked)
constexpr int SS_ROWS_F = GSR_SS_ROWS_FWD;         // ... of the forward
constexpr int SS_W = SS_COLS + 2 * SSIM_R;         // 74 inputs per row
constexpr int SS_WAVES = 4;                        // strips per 256-thread workgroup (independent)

// Order one lane's LDS stores before the other lanes' loads of the same row (hardware executes a wave's LDS
// operations in order; this keeps the COMPILER from moving a load of column lane + k above the store of
// column lane, which alias analysis of a single thread would allow).
__device__ __forceinline__ void ss_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float sym11(const float v[11]) {     // sum_k G[k] v[k], G symmetric
  float a = SSIM_G[5] * v[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) a = fmaf(SSIM_G[k], v[k] + v[10 - k], a);
  return a;
}

__global__ void __launch_bounds__(64 * SS_WAVES)
ssim_fwd_sw_kernel(int N, int CH, int H, int W, const float *__restrict__ img1, ImgView v1,
                   const float *__restrict__ img2, ImgView v2, int valid_only,
                   double *__restrict__ ws, float *__restrict__ dm_mu1,
                   float *__restrict__ dm_s1, float *__restrict__ dm_s12) {
  __shared__ float sRow[SS_WAVES][2][2][SS_W + 2];   // [wave][buffer][image][column]
  __shared__ double red[2][SS_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tiles_x = (W + SS_COLS - 1) / SS_COLS, strips = (H + SS_ROWS_F - 1) / SS_ROWS_F;
  const int wid = blockIdx.x * SS_WAVES + wave;
  const int plane = blockIdx.y, n = plane / CH, c = plane % CH;
  // per-lane partial sums in fp32 (<= 42 terms of magnitude <= 1 each), fp64 across lanes and workgroups
  float acc_ssim = 0.f, acc_l1 = 0.f;
  if (wid < tiles_x * strips) {
    const int bx = wid % tiles_x, by = wid / tiles_x;
    const int x0 = bx * SS_COLS, y0 = by * SS_ROWS_F;
    const int rows_out = min(SS_ROWS_F, H - y0);
    const int gx = x0 + lane;                               // this lane's output column
    const int gxa = x0 - SSIM_R + lane, gxb = x0 + SS_COLS - SSIM_R + lane;   // columns it stages (b: lanes 0..9)
    const float *p1 = img1 + n * v1.sn + c * v1.sc, *p2 = img2 + n * v2.sn + c * v2.sc;
    // running pointers of the row being fetched (one 64-bit add per row instead of four 64-bit multiply-adds)
    const float *r1a = p1 + (int64_t)(y0 - SSIM_R) * v1.sh + (int64_t)gxa * v1.sw, *r1b = r1a + (int64_t)SS_COLS * v1.sw;
    const float *r2a = p2 + (int64_t)(y0 - SSIM_R) * v2.sh + (int64_t)gxa * v2.sw, *r2b = r2a + (int64_t)SS_COLS * v2.sw;
    const bool col_a = gxa >= 0 && gxa < W, col_b = lane < 2 * SSIM_R && gxb < W;
    auto fetch = [&](int j, float &xa, float &ya, float &xb, float &yb) {   // input row j of the strip (called for j = 0, 1, 2, ...)
      const int gy = y0 - SSIM_R + j;
      xa = ya = xb = yb = 0.f;
      if (gy >= 0 && gy < H) {
        if (col_a) {
          xa = *r1a;
          ya = *r2a;
        }
        if (col_b) {
          xb = *r1b;
          yb = *r2b;
        }
      }
      r1a += v1.sh;
      r1b += v1.sh;
      r2a += v2.sh;
      r2b += v2.sh;
    };
    const int64_t o_first = ((int64_t)plane * H + y0) * W + gx;   // this lane's element of the strip's first output row
    // horizontally filtered rows: x, y, xx + yy, xy. (The two second moments are only ever used as their sum: B = s1 + s2
    // + C2 = E[xx] + E[yy] - mu1^2 - mu2^2 + C2 -- four filtered quantities instead of five.)
    float wm1[11], wm2[11], wss[11], w12[11];
    float nxa, nya, nxb, nyb;
    fetch(0, nxa, nya, nxb, nyb);
    const int rows_in = rows_out + 2 * SSIM_R;
    for (int jb = 0; jb < rows_in; jb += 11) {
#pragma unroll
      for (int sl = 0; sl < 11; ++sl) {
        const int j = jb + sl;
        if (j < rows_in) {                                  // (wave-uniform)
          float(*buf)[SS_W + 2] = sRow[wave][j & 1];
          buf[0][lane] = nxa;
          buf[1][lane] = nya;
          if (lane < 2 * SSIM_R) {
            buf[0][SS_COLS + lane] = nxb;
            buf[1][SS_COLS + lane] = nyb;
          }
          ss_wave_sync();
          if (j + 1 < rows_in) fetch(j + 1, nxa, nya, nxb, nyb);   // in flight during this row's arithmetic
          float xs[11], ys[11], t[11];
#pragma unroll
          for (int k = 0; k < 11; ++k) {
            xs[k] = buf[0][lane + k];
            ys[k] = buf[1][lane + k];
          }
          // the L1 term of this row's own pixel (input row j is output row j - 5; the lane's column is tap 5)
          if (j >= SSIM_R && j < rows_out + SSIM_R && gx < W) acc_l1 += fabsf(xs[SSIM_R] - ys[SSIM_R]);
          wm1[sl] = sym11(xs);
          wm2[sl] = sym11(ys);
#pragma unroll
          for (int k = 0; k < 11; ++k) t[k] = fmaf(xs[k], xs[k], ys[k] * ys[k]);
          wss[sl] = sym11(t);
#pragma unroll
          for (int k = 0; k < 11; ++k) t[k] = xs[k] * ys[k];
          w12[sl] = sym11(t);
          if (j >= 2 * SSIM_R) {                            // output row i = j - 10: window rows j-10 .. j
            const int i = j - 2 * SSIM_R, gy = y0 + i;
            const int64_t o_row = o_first + (int64_t)i * W;
            float o1[11], o2[11], oss[11], o12[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) {                  // slot of input row (j - 10 + k)
              const int q = (sl + 1 + k) % 11;
              o1[k] = wm1[q];
              o2[k] = wm2[q];
              oss[k] = wss[q];
              o12[k] = w12[q];
            }
            const float mu1 = sym11(o1), mu2 = sym11(o2), ess = sym11(oss), e12 = sym11(o12);
            if (gx < W) {
              const float mm = mu1 * mu1 + mu2 * mu2, s12 = e12 - mu1 * mu2;
              const float A = mm + SSIM_C1, B = (ess - mm) + SSIM_C2;
              const float Cc = 2.f * mu1 * mu2 + SSIM_C1, D = 2.f * s12 + SSIM_C2;
              // (two v_rcp_f32 instead of four IEEE divisions: 44 of this loop's 278 VALU instructions per row went
              // into the division sequences; A >= C1 and B >= C2 are far from the denormal range)
              const float iA = __builtin_amdgcn_rcpf(A), iB = __builtin_amdgcn_rcpf(B);
              const float iAB = iA * iB;
              const float m = Cc * D * iAB;
              const bool counted = !valid_only || (gx >= SSIM_R && gx < W - SSIM_R && gy >= SSIM_R && gy < H - SSIM_R);
              if (counted) acc_ssim += m;
              if (dm_mu1) {
                const int64_t o = o_row;
                const float w = counted ? 1.f : 0.f;   // cropped pixels get no gradient
                dm_mu1[o] = w * (2.f * mu2 * D * iAB - 2.f * mu2 * Cc * iAB - 2.f * mu1 * m * iA +
                                 2.f * mu1 * m * iB);
                dm_s1[o] = w * (-m * iB);
                dm_s12[o] = w * (2.f * Cc * iAB);
              }
            }
          }
        }
      }
    }
  }
  const double w_ssim = wave_sum_f64((double)acc_ssim), w_l1 = wave_sum_f64((double)acc_l1);
  if (lane == 0) {
    red[0][wave] = w_ssim;
    red[1][wave] = w_l1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {      // one pair of partial sums per workgroup, plain stores (see ssim_fwd_kernel)
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int w = 0; w < SS_WAVES; ++w) {
      a += red[0][w];
      b += red[1][w];
    }
    ws[2 * wg + 0] = a;
    ws[2 * wg + 1] = b;
  }
}

__global__ void __launch_bounds__(64 * SS_WAVES)
ssim_bwd_sw_kernel(int N, int CH, int H, int W, const float *__restrict__ img1, ImgView v1,
                   const float *__restrict__ img2, ImgView v2, const float *__restrict__ dm_mu1,
                   const float *__restrict__ dm_s1, const float *__restrict__ dm_s12,
                   const float *__restrict__ weights, const float *__restrict__ upstream,
                   float scale_ssim, float scale_l1, float *__restrict__ grad, ImgView vg) {
  __shared__ float sRow[SS_WAVES][2][3][SS_W + 2];   // [wave][buffer][map][column]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tiles_x = (W + SS_COLS - 1) / SS_COLS, strips = (H + SS_ROWS - 1) / SS_ROWS;
  const int wid = blockIdx.x * SS_WAVES + wave;
  if (wid >= tiles_x * strips) return;
  const int plane = blockIdx.y, n = plane / CH, c = plane % CH;
  const int bx = wid % tiles_x, by = wid / tiles_x;
  const int x0 = bx * SS_COLS, y0 = by * SS_ROWS;
  const int rows_out = min(SS_ROWS, H - y0);
  const int gx = x0 + lane;
  const int gxa = x0 - SSIM_R + lane, gxb = x0 + SS_COLS - SSIM_R + lane;
  const float up = upstream ? upstream[0] : 1.0f;
  const float w_ssim = weights ? weights[0] : up * scale_ssim;
  const float w_l1 = weights ? weights[1] : up * scale_l1;
  // running offsets / pointers (one add per row each instead of 64-bit multiply-adds per access)
  const bool col_a = gxa >= 0 && gxa < W, col_b = lane < 2 * SSIM_R && gxb < W;
  int64_t ro = (int64_t)plane * H * W + (int64_t)(y0 - SSIM_R) * W;      // map row being fetched
  auto fetch = [&](int j, float a[2], float b[2], float d[2]) {           // (called for j = 0, 1, 2, ...)
    const int gy = y0 - SSIM_R + j;
    a[0] = a[1] = b[0] = b[1] = d[0] = d[1] = 0.f;
    if (gy >= 0 && gy < H) {
      if (col_a) {
        a[0] = dm_mu1[ro + gxa];
        b[0] = dm_s1[ro + gxa];
        d[0] = dm_s12[ro + gxa];
      }
      if (col_b) {
        a[1] = dm_mu1[ro + gxb];
        b[1] = dm_s1[ro + gxb];
        d[1] = dm_s12[ro + gxb];
      }
    }
    ro += W;
  };
  // the images' own pixel of the NEXT output row travels with the next map row (it was loaded at its use: a global
  // round trip per row in the wave's dependent chain)
  const float *px = img1 + n * v1.sn + c * v1.sc + (int64_t)y0 * v1.sh + (int64_t)gx * v1.sw;
  const float *py = img2 + n * v2.sn + c * v2.sc + (int64_t)y0 * v2.sh + (int64_t)gx * v2.sw;
  float *pg = grad + n * vg.sn + c * vg.sc + (int64_t)y0 * vg.sh + (int64_t)gx * vg.sw;
  float x_next = 0.f, y_next = 0.f;
  float wa[11], wb[11], wd[11];
  float na[2], nb[2], nd[2];
  fetch(0, na, nb, nd);
  const int rows_in = rows_out + 2 * SSIM_R;
  for (int jb = 0; jb < rows_in; jb += 11) {
#pragma unroll
    for (int sl = 0; sl < 11; ++sl) {
      const int j = jb + sl;
      if (j < rows_in) {
        float(*buf)[SS_W + 2] = sRow[wave][j & 1];
        buf[0][lane] = na[0];
        buf[1][lane] = nb[0];
        buf[2][lane] = nd[0];
        if (lane < 2 * SSIM_R) {
          buf[0][SS_COLS + lane] = na[1];
          buf[1][SS_COLS + lane] = nb[1];
          buf[2][SS_COLS + lane] = nd[1];
        }
        ss_wave_sync();
        if (j + 1 < rows_in) fetch(j + 1, na, nb, nd);
        const float x = x_next, y = y_next;                  // pixel of output row j - 10
        if (j + 1 >= 2 * SSIM_R && j + 1 < rows_in && gx < W) {   // output row j + 1 - 10 exists
          x_next = *px;
          y_next = *py;
          px += v1.sh;
          py += v2.sh;
        }
        float t[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) t[k] = buf[0][lane + k];
        wa[sl] = sym11(t);
#pragma unroll
        for (int k = 0; k < 11; ++k) t[k] = buf[1][lane + k];
        wb[sl] = sym11(t);
#pragma unroll
        for (int k = 0; k < 11; ++k) t[k] = buf[2][lane + k];
        wd[sl] = sym11(t);
        if (j >= 2 * SSIM_R) {
          float oa[11], ob[11], od[11];
#pragma unroll
          for (int k = 0; k < 11; ++k) {
            const int q = (sl + 1 + k) % 11;
            oa[k] = wa[q];
            ob[k] = wb[q];
            od[k] = wd[q];
          }
          const float a = sym11(oa), b = sym11(ob), d = sym11(od);
          if (gx < W) {
            const float df = x - y;
            const float sgn = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
            *pg = w_l1 * sgn + w_ssim * (a + 2.f * x * b + y * d);
            pg += vg.sh;
          }
        }
      }
    }
  }
}

// Plain L1 (runner.py:506 with ssim_lambda = 0): contiguous buffers, 16 B per lane.
__global__ void __launch_bounds__(256)
l1_fwd_kernel(int64_t n, const float *__restrict__ a, const float *__restrict__ b,
              double *__restrict__ ws, float *__restrict__ mean_out,
              float *__restrict__ unit_grad, float scale) {
  // fp32 partial per thread (a few dozen terms of magnitude <= 1), fp64 across threads.
  // Few, fat workgroups: the kernel ends with one same-address fp64 atomic per workgroup
  // and those serialise at the memory side (2048 of them cost more than the 50 MB read),
  // so each thread keeps 4 independent 16-byte load pairs in flight instead.
  float part = 0.f;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  const float4 *a4 = reinterpret_cast<const float4 *>(a);
  const float4 *b4 = reinterpret_cast<const float4 *>(b);
  // unit_grad (optional): d mean|a-b| / d a = sign(a-b)/n, written while the operands are in
  // registers anyway -- under the usual root gradient of 1 the backward then needs no launch
  float4 *g4 = reinterpret_cast<float4 *>(unit_grad);
  auto sg = [scale](float d) { return d > 0.f ? scale : (d < 0.f ? -scale : 0.f); };
  auto l1 = [&](int64_t idx, const float4 &x, const float4 &y) {
    const float dx = x.x - y.x, dy = x.y - y.y, dz = x.z - y.z, dw = x.w - y.w;
    if (unit_grad) g4[idx] = make_float4(sg(dx), sg(dy), sg(dz), sg(dw));
    return (fabsf(dx) + fabsf(dy)) + (fabsf(dz) + fabsf(dw));
  };
  int64_t i = tid;
  for (; i + 3 * nthreads < n4; i += 4 * nthreads) {
    const float4 x0 = a4[i], x1 = a4[i + nthreads], x2 = a4[i + 2 * nthreads], x3 = a4[i + 3 * nthreads];
    const float4 y0 = b4[i], y1 = b4[i + nthreads], y2 = b4[i + 2 * nthreads], y3 = b4[i + 3 * nthreads];
    part += (l1(i, x0, y0) + l1(i + nthreads, x1, y1)) +
            (l1(i + 2 * nthreads, x2, y2) + l1(i + 3 * nthreads, x3, y3));
  }
  for (; i < n4; i += nthreads) part += l1(i, a4[i], b4[i]);
  if (tid == 0)
    for (int64_t k = n4 << 2; k < n; ++k) {
      const float d = a[k] - b[k];
      if (unit_grad) unit_grad[k] = sg(d);
      part += fabsf(d);
    }
  __shared__ double red[4];
  double acc = wave_sum_f64((double)part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  // one partial per workgroup; l1_finalize_kernel (next launch: the boundary makes the partials
  // visible) sums them. Until round 3 the workgroups added into ONE fp64 word and took a ticket
  // from another: 1024 same-address atomics that serialise at the memory side (~12 ns each, half of
  // the kernel's 27 us).
  if (threadIdx.x == 0) ws[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(512)
l1_finalize_kernel(int n_partials, const double *__restrict__ ws, double inv_n, float *__restrict__ mean_out) {
  __shared__ double red[8];
  double acc = (int)threadIdx.x < n_partials ? ws[threadIdx.x] : 0.0;
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 8; ++w) t += red[w];
    mean_out[0] = (float)(t * inv_n);
  }
}
__global__ void __launch_bounds__(256)
l1_bwd_kernel(int64_t n, const float *__restrict__ a, const float *__restrict__ b,
              const float *__restrict__ upstream, float scale, float *__restrict__ grad) {
  const float w = (upstream ? upstream[0] : 1.0f) * scale;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  auto sg = [w](float d) { return d > 0.f ? w : (d < 0.f ? -w : 0.f); };
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 x = *reinterpret_cast<const float4 *>(a + i);
      const float4 y = *reinterpret_cast<const float4 *>(b + i);
      *reinterpret_cast<float4 *>(grad + i) =
          make_float4(sg(x.x - y.x), sg(x.y - y.y), sg(x.z - y.z), sg(x.w - y.w));
    } else {
      for (int64_t k = i; k < n; ++k) grad[k] = sg(a[k] - b[k]);
    }
  }
}

}  // namespace gsr

extern "C" int gsr_l1_fwd(int64_t n, const float *a, const float *b, double *workspace,
                          float *mean_out, float *unit_grad, void *stream) {
  GSR_REQUIRE(n > 0 && a && b && workspace && mean_out, "l1_fwd: bad arguments");
  GSR_REQUIRE(((uintptr_t)a | (uintptr_t)b | (uintptr_t)unit_grad) % 16 == 0,
              "l1_fwd: buffers must be 16-byte aligned");
  int blocks = (int)(gsr::ceil_div64(n, 4096) < GSR_L1_WS_DOUBLES ? gsr::ceil_div64(n, 4096) : GSR_L1_WS_DOUBLES);
  hipLaunchKernelGGL(gsr::l1_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, a, b,
                     workspace, mean_out, unit_grad, (float)(1.0 / (double)n));
  GSR_CHECK_LAUNCH("l1_fwd");
  hipLaunchKernelGGL(gsr::l1_finalize_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, blocks,
                     workspace, 1.0 / (double)n, mean_out);
  GSR_CHECK_LAUNCH("l1_finalize");
  return GSR_OK;
}

extern "C" int gsr_l1_bwd(int64_t n, const float *a, const float *b, const float *upstream,
                          float scale, float *grad, void *stream) {
  GSR_REQUIRE(n >= 0 && a && b && grad, "l1_bwd: bad arguments");
  GSR_REQUIRE(((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad) % 16 == 0,
              "l1_bwd: buffers must be 16-byte aligned");
  if (n == 0) return GSR_OK;
  int blocks = (int)(gsr::ceil_div64(n, 1024) < 2048 ? gsr::ceil_div64(n, 1024) : 2048);
  hipLaunchKernelGGL(gsr::l1_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, a, b,
                     upstream, scale, grad);
  GSR_CHECK_LAUNCH("l1_bwd");
  return GSR_OK;
}

// img strides are in ELEMENTS for a logical [N, CH, H, W] image (pass the
// strides of an NHWC tensor permuted to NCHW to consume it in place).
// out[3] (device float) = {mean SSIM over the counted region, mean |img1-img2| over all
// pixels, (1-lambda)*L1 + lambda*(1-SSIM)}; workspace = 2 device doubles per workgroup
// (gsr_ssim_workspace_doubles), any content. dm_* [N,CH,H,W] may be NULL (no backward).
extern "C" int64_t gsr_ssim_workspace_doubles(int N, int CH, int H, int W) {
  return 2 * (int64_t)gsr::ceil_div(W, gsr::SSIM_T) * gsr::ceil_div(H, gsr::SSIM_T) * N * CH;
}
extern "C" int gsr_ssim_l1_fwd(int N, int CH, int H, int W, const float *img1,
                               const int64_t *strides1, const float *img2,
                               const int64_t *strides2, int valid_only, double *workspace,
                               float *out, float ssim_lambda, float *dm_mu1, float *dm_s1,
                               float *dm_s12, void *stream) {
  GSR_REQUIRE(N > 0 && CH > 0 && H > 0 && W > 0, "ssim_l1_fwd: bad sizes");
  GSR_REQUIRE(img1 && img2 && strides1 && strides2 && workspace && out, "ssim_l1_fwd: null pointer");
  GSR_REQUIRE((dm_mu1 == nullptr) == (dm_s1 == nullptr) && (dm_s1 == nullptr) == (dm_s12 == nullptr),
              "ssim_l1_fwd: pass all three derivative maps or none");
  GSR_REQUIRE((int64_t)N * CH < 65536, "ssim_l1_fwd: too many planes");
  const int hh = valid_only ? H - 10 : H, ww = valid_only ? W - 10 : W;
  const double n_ssim = (double)N * CH * (hh > 0 ? hh : 0) * (ww > 0 ? ww : 0);
  const double n_l1 = (double)N * CH * H * W;
  gsr::ImgView v1{strides1[0], strides1[1], strides1[2], strides1[3]};
  gsr::ImgView v2{strides2[0], strides2[1], strides2[2], strides2[3]};
#ifdef GSR_SSIM_TILE_KERNELS
  dim3 grid(gsr::ceil_div(W, gsr::SSIM_T) * gsr::ceil_div(H, gsr::SSIM_T), N * CH);
  hipLaunchKernelGGL(gsr::ssim_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, N, CH, H, W,
                     img1, v1, img2, v2, valid_only, workspace, dm_mu1, dm_s1, dm_s12);
#else
  // (a strip of 64 x 24 pixels per wave, four waves per workgroup: never more partial sums than the 32 x 32 tiles the
  // workspace is sized for)
  dim3 grid(gsr::ceil_div(gsr::ceil_div(W, gsr::SS_COLS) * gsr::ceil_div(H, gsr::SS_ROWS_F), gsr::SS_WAVES), N * CH);
  hipLaunchKernelGGL(gsr::ssim_fwd_sw_kernel, grid, dim3(64 * gsr::SS_WAVES), 0, (hipStream_t)stream, N, CH, H, W,
                     img1, v1, img2, v2, valid_only, workspace, dm_mu1, dm_s1, dm_s12);
#endif
  GSR_CHECK_LAUNCH("ssim_l1_fwd");
  hipLaunchKernelGGL(gsr::ssim_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                     (int)(grid.x * grid.y), workspace, out, 1.0 / (n_ssim > 0 ? n_ssim : 1.0),
                     1.0 / n_l1, ssim_lambda);
  GSR_CHECK_LAUNCH("ssim_l1_fwd");
  return GSR_OK;
}

// grad (strides stridesg) = w_l1*sign(img1-img2) + w_ssim*dSSIM/dimg1 with either
// (w_ssim, w_l1) = weights[0..1] (device float[2]) or, weights NULL, = upstream[0] *
// (scale_ssim, scale_l1): the upstream gradient is a device scalar and never visits the host.
extern "C" int gsr_ssim_l1_bwd(int N, int CH, int H, int W, const float *img1,
                               const int64_t *strides1, const float *img2,
                               const int64_t *strides2, const float *dm_mu1, const float *dm_s1,
                               const float *dm_s12, const float *weights, const float *upstream,
                               float scale_ssim, float scale_l1, float *grad,
                               const int64_t *stridesg, void *stream) {
  GSR_REQUIRE(N > 0 && CH > 0 && H > 0 && W > 0, "ssim_l1_bwd: bad sizes");
  GSR_REQUIRE(img1 && img2 && strides1 && strides2 && dm_mu1 && dm_s1 && dm_s12 &&
                  grad && stridesg,
              "ssim_l1_bwd: null pointer");
  GSR_REQUIRE((int64_t)N * CH < 65536, "ssim_l1_bwd: too many planes");
  gsr::ImgView v1{strides1[0], strides1[1], strides1[2], strides1[3]};
  gsr::ImgView v2{strides2[0], strides2[1], strides2[2], strides2[3]};
  gsr::ImgView vg{stridesg[0], stridesg[1], stridesg[2], stridesg[3]};
#ifdef GSR_SSIM_TILE_KERNELS
  dim3 grid(gsr::ceil_div(W, gsr::SSIM_T) * gsr::ceil_div(H, gsr::SSIM_T), N * CH);
  hipLaunchKernelGGL(gsr::ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, N, CH, H, W,
                     img1, v1, img2, v2, dm_mu1, dm_s1, dm_s12, weights, upstream, scale_ssim,
                     scale_l1, grad, vg);
#else
  dim3 grid(gsr::ceil_div(gsr::ceil_div(W, gsr::SS_COLS) * gsr::ceil_div(H, gsr::SS_ROWS), gsr::SS_WAVES), N * CH);
  hipLaunchKernelGGL(gsr::ssim_bwd_sw_kernel, grid, dim3(64 * gsr::SS_WAVES), 0, (hipStream_t)stream, N, CH, H, W,
                     img1, v1, img2, v2, dm_mu1, dm_s1, dm_s12, weights, upstream, scale_ssim,
                     scale_l1, grad, vg);
#endif
  GSR_CHECK_LAUNCH("ssim_l1_bwd");
  return GSR_OK;
}
