// rbf.hip -- radial-basis-function interpolation of the per-point scale factors:
// the reference's `rbf_interpolation` (gs_init_compare/depth_alignment/alignment/interp.py:30-72) builds a
// torchrbf.RBFInterpolator (a port of scipy.interpolate.RBFInterpolator; neither is a dependency here) over
// at most `max_rbf_points` = 5000 sites, evaluates it on a grid 256 pixels wide and upsamples bilinearly.
// Restated algorithm and its pin: oracle/rbf_oracle.py. Everything is float64: the thin-plate system is
// ill-conditioned, and at 5003 unknowns the dense solve is a fraction of a second on one GPU.
//
//   gsr_rbf_fit       assemble [[K + sI, P], [P^T, 0]] and the right-hand side, LU with partial pivoting
//                     (one pivot/scale launch + one rank-1 update launch per column: an init-time path),
//                     forward / back substitution in one single-workgroup launch
//   gsr_rbf_eval_grid f(x) = sum_i c_i phi(|x - y_i|) + polynomial, on the reference's query grid
//   gsr_bilinear_ac_t F.interpolate(bilinear, align_corners=True) of the [qw, qh] grid to [W, H], transposed
#include <hip/hip_runtime.h>

#include "common.h"

namespace gsr {
namespace rbf {

// the scale-invariant kernels of scipy / torchrbf (the only ones usable without `epsilon`, which the
// reference's call never passes): minimum polynomial degree 0, 1, 1, 2
enum Kernel { K_LINEAR = 0, K_TPS = 1, K_CUBIC = 2, K_QUINTIC = 3, K_COUNT = 4 };
constexpr int MAX_MONOMIALS = 6;

__device__ __forceinline__ double phi(double r, int kernel) {
  if (kernel == K_LINEAR) return -r;
  if (kernel == K_CUBIC) return r * r * r;
  if (kernel == K_QUINTIC) return -(r * r) * (r * r) * r;
  return r == 0.0 ? 0.0 : r * r * log(r);
}
__host__ __device__ __forceinline__ int n_monomials(int kernel) {
  return kernel == K_LINEAR ? 1 : (kernel == K_QUINTIC ? 6 : 3);
}
// monomial m of the normalised point (u, v), scipy's order (_monomial_powers): 1, u, v, u^2, u v, v^2
__device__ __forceinline__ double monomial(int m, double u, double v) {
  switch (m) {
    case 0: return 1.0;
    case 1: return u;
    case 2: return v;
    case 3: return u * u;
    case 4: return u * v;
    default: return v * v;
  }
}

// min / max of the sites -> shift, scale (scipy: (max + min) / 2, (max - min) / 2, 1 where that is 0)
__global__ void __launch_bounds__(1024)
shift_scale_kernel(int P, const float *__restrict__ y, double *__restrict__ ss) {
  __shared__ float smin[2][16], smax[2][16];
  float mn[2] = {3.0e38f, 3.0e38f}, mx[2] = {-3.0e38f, -3.0e38f};
  for (int i = threadIdx.x; i < P; i += 1024)
    for (int k = 0; k < 2; ++k) {
      mn[k] = fminf(mn[k], y[2 * i + k]);
      mx[k] = fmaxf(mx[k], y[2 * i + k]);
    }
  for (int k = 0; k < 2; ++k) {
    for (int o = 32; o; o >>= 1) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], o, 64));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      smin[k][threadIdx.x >> 6] = mn[k];
      smax[k][threadIdx.x >> 6] = mx[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int k = threadIdx.x;
    float a = smin[k][0], b = smax[k][0];
    for (int w = 1; w < 16; ++w) {
      a = fminf(a, smin[k][w]);
      b = fmaxf(b, smax[k][w]);
    }
    const double sc = ((double)b - (double)a) / 2.0;
    ss[k] = ((double)b + (double)a) / 2.0;
    ss[2 + k] = sc == 0.0 ? 1.0 : sc;
  }
}

// lhs [n, n] row-major, n = P + R; rhs [n]
__global__ void __launch_bounds__(256)
assemble_kernel(int P, int n, const float *__restrict__ y, const float *__restrict__ d, double smoothing,
                int kernel, const double *__restrict__ ss, double *__restrict__ A, double *__restrict__ b) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * n) return;
  const int i = (int)(idx / n), j = (int)(idx - (int64_t)i * n);
  double v = 0.0;
  auto mono = [&](int site, int m) -> double {
    return monomial(m, ((double)y[2 * site] - ss[0]) / ss[2], ((double)y[2 * site + 1] - ss[1]) / ss[3]);
  };
  if (i < P && j < P) {
    const double dx = (double)y[2 * i] - (double)y[2 * j], dy = (double)y[2 * i + 1] - (double)y[2 * j + 1];
    v = phi(sqrt(dx * dx + dy * dy), kernel) + (i == j ? smoothing : 0.0);
  } else if (i < P) {
    v = mono(i, j - P);
  } else if (j < P) {
    v = mono(j, i - P);
  }
  A[idx] = v;
  if (j == 0) b[i] = i < P ? (double)d[i] : 0.0;
}

// column k: pivot = first row of the largest |A[i][k]|, i >= k (LAPACK idamax), rows k and pivot swapped
// over their whole length (and the right-hand side), multipliers A[i][k] /= A[k][k]
__global__ void __launch_bounds__(1024)
lu_pivot_kernel(int n, int k, double *__restrict__ A, double *__restrict__ b, int *__restrict__ singular) {
  __shared__ double sv[16];
  __shared__ int si[16], piv_s;
  const int tid = threadIdx.x;
  double best = -1.0;
  int bi = n;
  for (int i = k + tid; i < n; i += 1024) {
    const double a = fabs(A[(int64_t)i * n + k]);
    if (a > best) {                      // strided ascending i per thread: the first maximum stays
      best = a;
      bi = i;
    }
  }
  for (int o = 32; o; o >>= 1) {
    const double ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if ((tid & 63) == 0) {
    sv[tid >> 6] = best;
    si[tid >> 6] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (sv[w] > sv[0] || (sv[w] == sv[0] && si[w] < si[0])) {
        sv[0] = sv[w];
        si[0] = si[w];
      }
    piv_s = si[0];
    if (!(sv[0] > 0.0)) *singular = 1;
  }
  __syncthreads();
  const int p = piv_s;
  if (p != k && p < n) {
    for (int j = tid; j < n; j += 1024) {
      const double t = A[(int64_t)k * n + j];
      A[(int64_t)k * n + j] = A[(int64_t)p * n + j];
      A[(int64_t)p * n + j] = t;
    }
    if (tid == 0) {
      const double t = b[k];
      b[k] = b[p];
      b[p] = t;
    }
  }
  __syncthreads();
  const double inv = 1.0 / A[(int64_t)k * n + k];
  for (int i = k + 1 + tid; i < n; i += 1024) A[(int64_t)i * n + k] *= inv;
}

// trailing update A[i][j] -= A[i][k] * A[k][j], i, j > k
__global__ void __launch_bounds__(256)
lu_update_kernel(int n, int k, double *__restrict__ A) {
  const int j = k + 1 + blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = k + 1 + (blockIdx.y * 4 + (threadIdx.x >> 6)) * 8;
  if (j >= n) return;
  const double u = A[(int64_t)k * n + j];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int i = i0 + r;
    if (i < n) A[(int64_t)i * n + j] -= A[(int64_t)i * n + k] * u;
  }
}

// L y = b (unit lower), U x = y, column by column, one workgroup (the right-hand side was permuted with the rows)
__global__ void __launch_bounds__(1024)
lu_solve_kernel(int n, const double *__restrict__ A, double *__restrict__ b) {
  const int tid = threadIdx.x;
  for (int k = 0; k < n; ++k) {
    const double bk = b[k];
    for (int i = k + 1 + tid; i < n; i += 1024) b[i] -= A[(int64_t)i * n + k] * bk;
    __syncthreads();
  }
  for (int k = n - 1; k >= 0; --k) {
    if (tid == 0) b[k] /= A[(int64_t)k * n + k];
    __syncthreads();
    const double bk = b[k];
    for (int i = tid; i < k; i += 1024) b[i] -= A[(int64_t)i * n + k] * bk;
    __syncthreads();
  }
}

// query grid of the reference: x-major, x_a = a / (qw - 1), y_b = b / (qh - 1) in float32 (torch.linspace)
__global__ void __launch_bounds__(256)
eval_grid_kernel(int P, const float *__restrict__ y, const double *__restrict__ coeffs, const double *__restrict__ ss,
                 int kernel, int qw, int qh, float *__restrict__ out) {
  __shared__ double sy[256][2], sc[256];
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int a = q / qh, bq = q - a * qh;
  // torch.linspace(0, 1, n): start + i * step for the first half, end - (n - 1 - i) * step for the second
  auto lin = [](int i, int n) -> float {
    if (n <= 1) return 0.f;
    const float step = 1.0f / (float)(n - 1);
    return i < n / 2 ? (float)i * step : 1.0f - (float)(n - 1 - i) * step;
  };
  const double x0 = (double)lin(min(a, qw - 1), qw), x1 = (double)lin(bq, qh);
  double acc = 0.0;
  for (int base = 0; base < P; base += 256) {
    const int i = base + threadIdx.x;
    if (i < P) {
      sy[threadIdx.x][0] = (double)y[2 * i];
      sy[threadIdx.x][1] = (double)y[2 * i + 1];
      sc[threadIdx.x] = coeffs[i];
    }
    __syncthreads();
    const int m = min(256, P - base);
    for (int t = 0; t < m; ++t) {
      const double dx = x0 - sy[t][0], dy = x1 - sy[t][1];
      acc += sc[t] * phi(sqrt(dx * dx + dy * dy), kernel);
    }
    __syncthreads();
  }
  if (q >= qw * qh) return;
  {
    const double u = (x0 - ss[0]) / ss[2], v = (x1 - ss[1]) / ss[3];
    const int R = n_monomials(kernel);
    for (int m = 0; m < R; ++m) acc += coeffs[P + m] * monomial(m, u, v);
  }
  out[q] = (float)acc;
}

// out[h][w] = bilinear(align_corners) sample of src [qw, qh] at (w (qw-1)/(W-1), h (qh-1)/(H-1))
__global__ void __launch_bounds__(256)
bilinear_ac_t_kernel(int qw, int qh, const float *__restrict__ src, int W, int H, float *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)W * H) return;
  const int h = (int)(idx / W), w = (int)(idx - (int64_t)h * W);
  const float sw = W > 1 ? (float)(qw - 1) / (float)(W - 1) : 0.f, sh = H > 1 ? (float)(qh - 1) / (float)(H - 1) : 0.f;
  const float fw = sw * (float)w, fh = sh * (float)h;
  const int w0 = min((int)fw, qw - 1), h0 = min((int)fh, qh - 1);
  const int w1 = min(w0 + 1, qw - 1), h1 = min(h0 + 1, qh - 1);
  const float lw = fw - (float)w0, lh = fh - (float)h0;
  // src is [qw, qh]: first index along the image's x. ATen's order: rows of the output's first dimension
  // (here x) blended first along the second (y).
  const float v0 = src[(int64_t)w0 * qh + h0] * (1.f - lh) + src[(int64_t)w0 * qh + h1] * lh;
  const float v1 = src[(int64_t)w1 * qh + h0] * (1.f - lh) + src[(int64_t)w1 * qh + h1] * lh;
  out[idx] = v0 * (1.f - lw) + v1 * lw;
}

}  // namespace rbf
}  // namespace gsr

using namespace gsr::rbf;

extern "C" int64_t gsr_rbf_workspace_bytes(int P) {
  const int64_t n = (int64_t)P + MAX_MONOMIALS;
  return (n * n + n + 8) * (int64_t)sizeof(double) + 256;
}

extern "C" int gsr_rbf_fit(int P, const float *sites_xy, const float *values, double smoothing, int kernel,
                           void *workspace, int64_t workspace_bytes, double *coeffs, double *shift_scale,
                           void *stream) {
  GSR_REQUIRE(P >= 1 && kernel >= 0 && kernel < K_COUNT && smoothing >= 0.0, "rbf_fit: P=%d kernel=%d", P, kernel);
  GSR_REQUIRE(sites_xy && values && workspace && coeffs && shift_scale, "rbf_fit: null pointer");
  const int R = n_monomials(kernel);
  GSR_REQUIRE(P >= R, "rbf_fit: %d sites cannot determine a polynomial of %d terms", P, R);
  const int n = P + R;
  GSR_REQUIRE(workspace_bytes >= gsr_rbf_workspace_bytes(P) && ((uintptr_t)workspace & 7) == 0, "rbf_fit: workspace");
  hipStream_t st = (hipStream_t)stream;
  double *A = (double *)workspace;
  double *b = A + (int64_t)n * n;
  int *singular = (int *)(b + n + 4);
  GSR_CHECK_HIP(hipMemsetAsync(singular, 0, sizeof(int), st));
  hipLaunchKernelGGL(shift_scale_kernel, dim3(1), dim3(1024), 0, st, P, sites_xy, shift_scale);
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)gsr::ceil_div64((int64_t)n * n, 256)), dim3(256), 0, st, P, n,
                     sites_xy, values, smoothing, kernel, shift_scale, A, b);
  for (int k = 0; k < n; ++k) {
    hipLaunchKernelGGL(lu_pivot_kernel, dim3(1), dim3(1024), 0, st, n, k, A, b, singular);
    const int rest = n - k - 1;
    if (rest > 0)
      hipLaunchKernelGGL(lu_update_kernel, dim3((unsigned)gsr::ceil_div(rest, 64), (unsigned)gsr::ceil_div(rest, 32)),
                         dim3(256), 0, st, n, k, A);
  }
  hipLaunchKernelGGL(lu_solve_kernel, dim3(1), dim3(1024), 0, st, n, A, b);
  GSR_CHECK_HIP(hipMemcpyAsync(coeffs, b, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  GSR_CHECK_LAUNCH("rbf_fit");
  int h_singular = 0;
  GSR_CHECK_HIP(hipMemcpyAsync(&h_singular, singular, sizeof(int), hipMemcpyDeviceToHost, st));
  GSR_CHECK_HIP(hipStreamSynchronize(st));
  GSR_REQUIRE(!h_singular, "rbf_fit: singular system (coincident sites with smoothing 0?)");
  return GSR_OK;
}

extern "C" int gsr_rbf_eval_grid(int P, const float *sites_xy, const double *coeffs, const double *shift_scale,
                                 int kernel, int qw, int qh, float *out, void *stream) {
  GSR_REQUIRE(P >= 1 && kernel >= 0 && kernel < K_COUNT && qw >= 1 && qh >= 1, "rbf_eval_grid: bad sizes");
  GSR_REQUIRE(sites_xy && coeffs && shift_scale && out, "rbf_eval_grid: null pointer");
  hipLaunchKernelGGL(eval_grid_kernel, dim3((unsigned)gsr::ceil_div(qw * qh, 256)), dim3(256), 0, (hipStream_t)stream,
                     P, sites_xy, coeffs, shift_scale, kernel, qw, qh, out);
  GSR_CHECK_LAUNCH("rbf_eval_grid");
  return GSR_OK;
}

extern "C" int gsr_bilinear_ac_t(int qw, int qh, const float *src, int W, int H, float *out, void *stream) {
  GSR_REQUIRE(qw >= 1 && qh >= 1 && W >= 1 && H >= 1, "bilinear_ac_t: bad sizes");
  GSR_REQUIRE(src && out, "bilinear_ac_t: null pointer");
  hipLaunchKernelGGL(bilinear_ac_t_kernel, dim3((unsigned)gsr::ceil_div64((int64_t)W * H, 256)), dim3(256), 0,
                     (hipStream_t)stream, qw, qh, src, W, H, out);
  GSR_CHECK_LAUNCH("bilinear_ac_t");
  return GSR_OK;
}
