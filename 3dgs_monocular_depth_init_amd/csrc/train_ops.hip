// train_ops.hip -- the step either side of the rasterizer:
// one fused multi-tensor Adam launch over all Gaussian parameters, replacing
// the six per-parameter torch.optim.Adam steps of
// gs_init_compare/runner.py:129-137, 676-679 (7 x 236 MB of HBM traffic at
// 1M Gaussians: read p,g,m,v, write p,m,v -- streamed once, 16 B per lane).
#include "adam_math.h"
#include "common.h"
#include "gs_math.h"

namespace gsr {

constexpr int ADAM_MAX_TENSORS = 8;
constexpr int ADAM_ELEMS_PER_BLOCK = 256 * 4 * 4;   // 256 threads x float4 x 4

struct AdamArgs {
  float *p[ADAM_MAX_TENSORS];
  const float *g[ADAM_MAX_TENSORS];
  float *m[ADAM_MAX_TENSORS];
  float *v[ADAM_MAX_TENSORS];
  int64_t numel[ADAM_MAX_TENSORS];
  int32_t block_start[ADAM_MAX_TENSORS + 1];
  float step_size[ADAM_MAX_TENSORS];      // lr / (1 - beta1^t)
  float bc2_sqrt[ADAM_MAX_TENSORS];       // sqrt(1 - beta2^t)
  float beta1, beta2, eps;
  float omb1, omb2;                        // 1-beta1, 1-beta2 rounded from fp64 (as torch does)
  int n;
};

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
  int t = 0;
  const int b = blockIdx.x;
#pragma unroll
  for (int i = 1; i < ADAM_MAX_TENSORS; ++i)
    if (i < a.n && b >= a.block_start[i]) t = i;
  const int64_t base = (int64_t)(b - a.block_start[t]) * ADAM_ELEMS_PER_BLOCK;
  const int64_t n = a.numel[t];
  float *__restrict__ p = a.p[t];
  const float *__restrict__ g = a.g[t];
  float *__restrict__ m = a.m[t];
  float *__restrict__ v = a.v[t];
  const float ss = a.step_size[t], bc2 = a.bc2_sqrt[t];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + ((int64_t)r * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
      // gradients and moments are touched once per step: NONTEMPORAL 16-byte accesses keep them from
      // pushing the parameters out of L2 / Infinity Cache before the next forward reads them
      // (measured: this kernel 0.300 -> 0.253 ms, the next projection forward 0.114 -> 0.083 ms)
      typedef float f4v __attribute__((ext_vector_type(4)));
      float4 pp = *reinterpret_cast<float4 *>(p + i);
      const f4v gv = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(g + i));
      const f4v mv = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(m + i));
      const f4v vv_ = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(v + i));
      const float4 gg = make_float4(gv.x, gv.y, gv.z, gv.w);
      float4 mm = make_float4(mv.x, mv.y, mv.z, mv.w), vv = make_float4(vv_.x, vv_.y, vv_.z, vv_.w);
      adam_one(pp.x, gg.x, mm.x, vv.x, a.omb1, a.beta2, a.omb2, a.eps, ss, bc2);
      adam_one(pp.y, gg.y, mm.y, vv.y, a.omb1, a.beta2, a.omb2, a.eps, ss, bc2);
      adam_one(pp.z, gg.z, mm.z, vv.z, a.omb1, a.beta2, a.omb2, a.eps, ss, bc2);
      adam_one(pp.w, gg.w, mm.w, vv.w, a.omb1, a.beta2, a.omb2, a.eps, ss, bc2);
      *reinterpret_cast<float4 *>(p + i) = pp;
      {
        f4v mo = {mm.x, mm.y, mm.z, mm.w}, vo = {vv.x, vv.y, vv.z, vv.w};
        __builtin_nontemporal_store(mo, reinterpret_cast<f4v *>(m + i));
        __builtin_nontemporal_store(vo, reinterpret_cast<f4v *>(v + i));
      }
    } else {
      for (int64_t k = i; k < n && k < i + 4; ++k) {
        float pp = p[k], mm = m[k], vv = v[k];
        adam_one(pp, g[k], mm, vv, a.omb1, a.beta2, a.omb2, a.eps, ss, bc2);
        p[k] = pp;
        m[k] = mm;
        v[k] = vv;
      }
    }
  }
}

// Batched 4x4 inverse (adjugate / determinant, fp64 inside): replaces the
// rocSOLVER LU pipeline torch.linalg.inv launches (~14 tiny kernels, ~50 us)
// for `viewmats = inv(camtoworlds)` (runner.py:347) and for the camera
// positions the SH view directions need.
__global__ void inverse4x4_kernel(int C, const float *__restrict__ in, float *__restrict__ out,
                                  float *__restrict__ in_translation,
                                  float *__restrict__ out_translation) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double m[16], inv[16];
  for (int k = 0; k < 16; ++k) m[k] = in[c * 16 + k];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const double idet = 1.0 / det;
  for (int k = 0; k < 16; ++k) out[c * 16 + k] = (float)(inv[k] * idet);
  if (in_translation) {
    in_translation[c * 3 + 0] = in[c * 16 + 3];
    in_translation[c * 3 + 1] = in[c * 16 + 7];
    in_translation[c * 3 + 2] = in[c * 16 + 11];
  }
  if (out_translation) {
    out_translation[c * 3 + 0] = (float)(inv[3] * idet);
    out_translation[c * 3 + 1] = (float)(inv[7] * idet);
    out_translation[c * 3 + 2] = (float)(inv[11] * idet);
  }
}

// ---- MCMC strategy (reference call site runner.py:649-658) ----
// Relocation ("3D Gaussian Splatting as Markov Chain Monte Carlo", eq. 9): a Gaussian
// that is to be represented by N copies gets opacity o' = 1 - (1 - o)^(1/N) and its
// scales multiplied by o / sum_{i=1..N} sum_{k=0..i-1} C(i-1,k) (-1)^k o'^(k+1)/sqrt(k+1).
__global__ void __launch_bounds__(256)
relocation_kernel(int n, const float *__restrict__ opacities, const float *__restrict__ scales,
                  const int32_t *__restrict__ ratios, const float *__restrict__ binoms, int n_max,
                  float *__restrict__ new_opacities, float *__restrict__ new_scales) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int N = min(max(ratios[i], 1), n_max);
  const float o = opacities[i];
  const float no = 1.0f - powf(1.0f - o, 1.0f / (float)N);
  float denom = 0.f;
  for (int a = 1; a <= N; ++a) {
    float pw = no, sgn = 1.f;                 // o'^(k+1), (-1)^k
    for (int k = 0; k <= a - 1; ++k) {
      denom += binoms[(a - 1) * n_max + k] * (sgn * rsqrtf((float)(k + 1)) * pw);
      pw *= no;
      sgn = -sgn;
    }
  }
  const float coeff = o / denom;
  new_opacities[i] = no;
  new_scales[i * 3 + 0] = coeff * scales[i * 3 + 0];
  new_scales[i * 3 + 1] = coeff * scales[i * 3 + 1];
  new_scales[i * 3 + 2] = coeff * scales[i * 3 + 2];
}

// Position noise of the MCMC strategy, fused: means += Sigma * (noise * s(1 - sigmoid(op)) *
// scaler), Sigma = R diag(exp(scales))^2 R^T, s(x) = 1/(1 + exp(-100 (x - 0.995))).
// Replaces sigmoid, exp, quat->covariance, the gate, an einsum and the add (seven
// passes over N) by one 68-byte-per-Gaussian pass.
__global__ void __launch_bounds__(256)
inject_noise_kernel(int N, float *__restrict__ means, const float *__restrict__ quats,
                    const float *__restrict__ log_scales, const float *__restrict__ logit_opac,
                    const float *__restrict__ noise, float scaler) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float op = 1.0f / (1.0f + expf(-logit_opac[i]));
  const float gate = 1.0f / (1.0f + expf(-100.0f * ((1.0f - op) - 0.995f))) * scaler;
  const float q[4] = {quats[i * 4], quats[i * 4 + 1], quats[i * 4 + 2], quats[i * 4 + 3]};
  const float sc[3] = {expf(log_scales[i * 3]), expf(log_scales[i * 3 + 1]), expf(log_scales[i * 3 + 2])};
  const gs::Mat3 cov = gs::quat_scale_to_covar(q, sc);
  const float nz[3] = {noise[i * 3] * gate, noise[i * 3 + 1] * gate, noise[i * 3 + 2] * gate};
#pragma unroll
  for (int r = 0; r < 3; ++r)
    means[i * 3 + r] += cov.m[r][0] * nz[0] + cov.m[r][1] * nz[1] + cov.m[r][2] * nz[2];
}

// DefaultStrategy statistics (gsplat's DefaultStrategy._update_state, driven from
// runner.py:639-647), one launch and no host sync instead of clone / scale / nonzero /
// index_add / maximum: for every (camera, Gaussian) pair with both radii > 0,
//   grad2d[i] += || (g.x * sx, g.y * sy) ||,  count[i] += 1,
//   radii_state[i] = max(radii_state[i], max(rx, ry) / max_wh)      (optional)
// g = means2d gradient of the pair, read with a row stride (it lives in the 64-byte rows).
__global__ void __launch_bounds__(256)
strategy_accumulate_kernel(int C, int N, const float *__restrict__ grad, int grad_stride,
                           const int32_t *__restrict__ radii, float sx, float sy,
                           float *__restrict__ grad2d, float *__restrict__ count,
                           float *__restrict__ radii_state, float inv_max_wh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float acc = 0.f, cnt = 0.f, rmax = 0.f;
  for (int c = 0; c < C; ++c) {
    const int64_t g = (int64_t)c * N + i;
    const int rx = radii[g * 2], ry = radii[g * 2 + 1];
    if (rx <= 0 || ry <= 0) continue;
    const float gx = grad[g * grad_stride] * sx, gy = grad[g * grad_stride + 1] * sy;
    acc += sqrtf(gx * gx + gy * gy);
    cnt += 1.f;
    rmax = fmaxf(rmax, (float)max(rx, ry) * inv_max_wh);
  }
  if (cnt > 0.f) {
    grad2d[i] += acc;
    count[i] += cnt;
    if (radii_state) radii_state[i] = fmaxf(radii_state[i], rmax);
  }
}

}  // namespace gsr

// ---------------------------------------------------------------------------------------------
// F2: one-pass densification (DefaultStrategy refine step: duplicate -> split -> prune,
// gsplat.strategy.ops as driven by gs_init_compare/runner.py:639-647). The three operations of a
// refine step each rebuild all six parameter tensors and their twelve Adam moment tensors when
// written as tensor ops; here the per-Gaussian decisions are one launch (refine_decide), their
// output positions one scan, and every tensor is rebuilt by ONE multi-tensor gather launch.
// Output order = what the three operations produce in sequence: surviving unsplit originals in
// order, surviving duplicates in order, surviving split children (sample 0 block, sample 1 block).
// flags rows (int32 [5,N]): original kept, duplicate kept, split children kept, is split (draws noise
// whether or not its children survive), is duplicated (for the counts the strategy reports).
namespace gsr {
struct RefineParams {
  float grow_grad2d, grow_scale3d, grow_scale2d /* < 0: off */, prune_opa, prune_scale3d /* < 0: off */,
      prune_scale2d /* < 0: off */;
  int revised_opacity;
};

__global__ void __launch_bounds__(256)
refine_decide_kernel(int N, const float *__restrict__ log_scales, const float *__restrict__ logit_opac,
                     const float *__restrict__ grad2d, const float *__restrict__ count,
                     const float *__restrict__ radii_state, RefineParams rp, int32_t *__restrict__ flags4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float s0 = expf(log_scales[i * 3]), s1 = expf(log_scales[i * 3 + 1]), s2 = expf(log_scales[i * 3 + 2]);
  const float smax = fmaxf(fmaxf(s0, s1), s2);
  const float g = grad2d[i] / fmaxf(count[i], 1.0f);
  const bool grad_high = g > rp.grow_grad2d;
  const bool small = smax <= rp.grow_scale3d;
  const float rad = radii_state ? radii_state[i] : 0.f;
  const bool dup = grad_high && small;
  bool split = grad_high && !small;
  if (rp.grow_scale2d >= 0.f) split = split || (rad > rp.grow_scale2d);
  // prune test on the values the surviving copy would carry
  auto pruned = [&](float opac_logit, float scale_max) {
    bool pr = (1.0f / (1.0f + expf(-opac_logit))) < rp.prune_opa;
    if (rp.prune_scale3d >= 0.f) {
      bool big = scale_max > rp.prune_scale3d;
      if (rp.prune_scale2d >= 0.f) big = big || (rad > rp.prune_scale2d);
      pr = pr || big;
    }
    return pr;
  };
  const float o = logit_opac[i];
  const bool own_pruned = pruned(o, smax);
  // children: scales log(exp(s) / 1.6) (re-exponentiated as the prune test reads them), opacity
  // optionally 1 - sqrt(1 - sigmoid(o)) stored as a logit
  const float c0 = expf(logf(s0 / 1.6f)), c1 = expf(logf(s1 / 1.6f)), c2 = expf(logf(s2 / 1.6f));
  float oc = o;
  if (rp.revised_opacity) {
    const float no = 1.0f - sqrtf(1.0f - 1.0f / (1.0f + expf(-o)));
    oc = logf(no / (1.0f - no));
  }
  const bool child_pruned = pruned(oc, fmaxf(fmaxf(c0, c1), c2));
  flags4[i] = (!split && !own_pruned) ? 1 : 0;
  flags4[N + i] = (dup && !own_pruned) ? 1 : 0;
  flags4[2 * N + i] = (split && !child_pruned) ? 1 : 0;
  flags4[3 * N + i] = split ? 1 : 0;
  flags4[4 * N + i] = dup ? 1 : 0;
}

// incl4: inclusive scans of the four flag rows. src[r] = source row of output row r,
// kind[r] = 0 original (keeps its Adam moments), 1 duplicate, 2 / 3 split child with sample 0 / 1.
__global__ void __launch_bounds__(256)
refine_plan_kernel(int N, const int32_t *__restrict__ flags4, const int32_t *__restrict__ incl4, int n0, int n1,
                   int n2, int32_t *__restrict__ src, uint8_t *__restrict__ kind) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (flags4[i]) {
    const int r = incl4[i] - 1;
    src[r] = i;
    kind[r] = 0;
  }
  if (flags4[N + i]) {
    const int r = n0 + incl4[N + i] - 1;
    src[r] = i;
    kind[r] = 1;
  }
  if (flags4[2 * N + i]) {
    const int e = incl4[2 * N + i] - 1;
    src[n0 + n1 + e] = i;
    kind[n0 + n1 + e] = 2;
    src[n0 + n1 + n2 + e] = i;
    kind[n0 + n1 + n2 + e] = 3;
  }
}

constexpr int REFINE_MAX_TENSORS = 24;
struct GatherArgs {
  const float *src[REFINE_MAX_TENSORS];
  float *dst[REFINE_MAX_TENSORS];
  int row_len[REFINE_MAX_TENSORS];
  int zero_new[REFINE_MAX_TENSORS];   // 1: rows of kind != 0 are zero (Adam moments of new Gaussians)
};

// blockIdx.y = tensor, blockIdx.x = a block of 256 output rows whose source rows and kinds are
// staged in LDS once; the block then walks the 256 * L output elements of its rows linearly
// (coalesced writes; reads are whole source rows). L is a compile-time constant for the row
// lengths of the Gaussian parameters (1, 3, 4, 45) so that the element -> (row, column) split is a
// multiply-shift, not a division.
template <int LC>
__device__ __forceinline__ void gather_rows(int rows, int L, const int32_t *sSrc, const uint8_t *sKind, bool zn,
                                            const float *__restrict__ s, float *__restrict__ d) {
  const int Lr = LC > 0 ? LC : L;
  const int total = rows * Lr;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int r = e / Lr, c = e - r * Lr;
    d[e] = (zn && sKind[r] != 0) ? 0.f : s[(int64_t)sSrc[r] * Lr + c];
  }
}

__global__ void __launch_bounds__(256)
refine_gather_kernel(int M, const int32_t *__restrict__ src_row, const uint8_t *__restrict__ kind, GatherArgs a) {
  __shared__ int32_t sSrc[256];
  __shared__ uint8_t sKind[256];
  const int t = blockIdx.y;
  const int L = a.row_len[t];
  const bool zn = a.zero_new[t] != 0;
  const int r0 = blockIdx.x * 256;
  const int rows = min(256, M - r0);
  if (rows <= 0) return;
  if ((int)threadIdx.x < rows) {
    sSrc[threadIdx.x] = src_row[r0 + threadIdx.x];
    sKind[threadIdx.x] = kind[r0 + threadIdx.x];
  }
  __syncthreads();
  const float *s = a.src[t];
  float *d = a.dst[t] + (int64_t)r0 * L;
  switch (L) {
    case 1: gather_rows<1>(rows, L, sSrc, sKind, zn, s, d); break;
    case 3: gather_rows<3>(rows, L, sSrc, sKind, zn, s, d); break;
    case 4: gather_rows<4>(rows, L, sSrc, sKind, zn, s, d); break;
    case 45: gather_rows<45>(rows, L, sSrc, sKind, zn, s, d); break;
    default: gather_rows<0>(rows, L, sSrc, sKind, zn, s, d); break;
  }
}
}  // namespace gsr

extern "C" int gsr_refine_decide(int N, const float *log_scales, const float *logit_opacities,
                                 const float *grad2d, const float *count, const float *radii_state,
                                 float grow_grad2d, float grow_scale3d, float grow_scale2d, float prune_opa,
                                 float prune_scale3d, float prune_scale2d, int revised_opacity,
                                 int32_t *flags4, void *stream) {
  GSR_REQUIRE(N >= 0, "refine_decide: bad N");
  if (N == 0) return GSR_OK;
  GSR_REQUIRE(log_scales && logit_opacities && grad2d && count && flags4, "refine_decide: null pointer");
  GSR_REQUIRE((grow_scale2d < 0.f && prune_scale2d < 0.f) || radii_state,
              "refine_decide: screen-space thresholds need the radii statistic");
  gsr::RefineParams rp{grow_grad2d, grow_scale3d, grow_scale2d, prune_opa, prune_scale3d, prune_scale2d,
                       revised_opacity};
  hipLaunchKernelGGL(gsr::refine_decide_kernel, dim3(gsr::ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, N,
                     log_scales, logit_opacities, grad2d, count, radii_state, rp, flags4);
  GSR_CHECK_LAUNCH("refine_decide");
  return GSR_OK;
}

extern "C" int gsr_refine_plan(int N, const int32_t *flags4, const int32_t *incl4, int n0, int n1, int n2,
                               int32_t *src_row, uint8_t *kind, void *stream) {
  GSR_REQUIRE(N >= 0 && n0 >= 0 && n1 >= 0 && n2 >= 0, "refine_plan: bad sizes");
  if (N == 0) return GSR_OK;
  GSR_REQUIRE(flags4 && incl4 && src_row && kind, "refine_plan: null pointer");
  hipLaunchKernelGGL(gsr::refine_plan_kernel, dim3(gsr::ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, N,
                     flags4, incl4, n0, n1, n2, src_row, kind);
  GSR_CHECK_LAUNCH("refine_plan");
  return GSR_OK;
}

// src / dst: HOST arrays of n device pointers (fp32 tensors [*, row_len[i]] / [M, row_len[i]]);
// row_len, zero_new: HOST arrays of n ints. n <= 24.
extern "C" int gsr_refine_gather(int n, int M, const int32_t *src_row, const uint8_t *kind,
                                 const void *const *src, void *const *dst, const int32_t *row_len,
                                 const int32_t *zero_new, void *stream) {
  GSR_REQUIRE(n >= 0 && n <= gsr::REFINE_MAX_TENSORS && M >= 0, "refine_gather: n=%d (max %d), M=%d", n,
              gsr::REFINE_MAX_TENSORS, M);
  if (n == 0 || M == 0) return GSR_OK;
  GSR_REQUIRE(src_row && kind && src && dst && row_len && zero_new, "refine_gather: null array");
  gsr::GatherArgs a;
  int max_len = 1;
  for (int i = 0; i < n; ++i) {
    GSR_REQUIRE(src[i] && dst[i] && row_len[i] > 0, "refine_gather: tensor %d", i);
    a.src[i] = (const float *)src[i];
    a.dst[i] = (float *)dst[i];
    a.row_len[i] = row_len[i];
    a.zero_new[i] = zero_new[i];
    max_len = row_len[i] > max_len ? row_len[i] : max_len;
  }
  (void)max_len;
  hipLaunchKernelGGL(gsr::refine_gather_kernel, dim3((unsigned)gsr::ceil_div(M, 256), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, M, src_row, kind, a);
  GSR_CHECK_LAUNCH("refine_gather");
  return GSR_OK;
}

extern "C" int gsr_strategy_accumulate(int C, int N, const float *grad, int grad_stride,
                                       const int32_t *radii, float sx, float sy, float *grad2d,
                                       float *count, float *radii_state, float max_wh,
                                       void *stream) {
  GSR_REQUIRE(C >= 0 && N >= 0 && grad_stride >= 2 && max_wh > 0.f, "strategy_accumulate: bad sizes");
  if ((int64_t)C * N == 0) return GSR_OK;
  GSR_REQUIRE(grad && radii && grad2d && count, "strategy_accumulate: null pointer");
  hipLaunchKernelGGL(gsr::strategy_accumulate_kernel, dim3(gsr::ceil_div(N, 256)), dim3(256), 0,
                     (hipStream_t)stream, C, N, grad, grad_stride, radii, sx, sy, grad2d, count,
                     radii_state, 1.0f / max_wh);
  GSR_CHECK_LAUNCH("strategy_accumulate");
  return GSR_OK;
}

extern "C" int gsr_relocation(int n, const float *opacities, const float *scales,
                              const int32_t *ratios, const float *binoms, int n_max,
                              float *new_opacities, float *new_scales, void *stream) {
  GSR_REQUIRE(n >= 0 && n_max > 0, "relocation: bad sizes");
  if (n == 0) return GSR_OK;
  GSR_REQUIRE(opacities && scales && ratios && binoms && new_opacities && new_scales,
              "relocation: null pointer");
  hipLaunchKernelGGL(gsr::relocation_kernel, dim3(gsr::ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, opacities, scales, ratios, binoms, n_max,
                     new_opacities, new_scales);
  GSR_CHECK_LAUNCH("relocation");
  return GSR_OK;
}

namespace gsr {
// gsplat.strategy.ops.reset_opa (DefaultStrategy every reset_every steps, runner.py:639-647): logit
// opacities clamped from above, the opacity optimizer's two moments cleared -- one pass, in place.
__global__ void __launch_bounds__(256)
reset_opacity_kernel(int64_t n, float *__restrict__ logit_opac, float *__restrict__ exp_avg,
                     float *__restrict__ exp_avg_sq, float max_logit) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  logit_opac[i] = fminf(logit_opac[i], max_logit);
  if (exp_avg) exp_avg[i] = 0.f;
  if (exp_avg_sq) exp_avg_sq[i] = 0.f;
}
}  // namespace gsr

extern "C" int gsr_reset_opacity(int64_t n, float *logit_opacities, float *exp_avg, float *exp_avg_sq,
                                 float max_logit, void *stream) {
  GSR_REQUIRE(n >= 0, "reset_opacity: bad n");
  if (n == 0) return GSR_OK;
  GSR_REQUIRE(logit_opacities, "reset_opacity: null pointer");
  hipLaunchKernelGGL(gsr::reset_opacity_kernel, dim3((unsigned)gsr::ceil_div64(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, logit_opacities, exp_avg, exp_avg_sq, max_logit);
  GSR_CHECK_LAUNCH("reset_opacity");
  return GSR_OK;
}

namespace gsr {
// torch.optim.SparseAdam (the reference's optimizer under cfg.sparse_grad, runner.py:130, 661-679) on the
// rows a step actually rendered: row r of every tensor is updated iff visible[r] != 0, every other row --
// parameter AND moments -- is left alone. visible[r] = the NUMBER of cameras of the batch that render row r: the
// reference builds sparse_coo_tensor(gaussian_ids, grad[gaussian_ids]) over the (camera, Gaussian) pairs, and
// SparseAdam's coalesce() sums the duplicates, i.e. a row seen by k cameras steps on k times its dense gradient
// (runner.py:661-672; with one camera k = 1). Arithmetic as torch/optim/_functional.py sparse_adam:
//   m += (g - m) (1 - b1);  v += (g^2 - v) (1 - b2);  p -= step_size * m / (sqrt(v) + eps),
//   step_size = lr sqrt(1 - b2^t) / (1 - b1^t)  (eps is NOT divided by the bias correction, unlike Adam).
constexpr int SPARSE_ADAM_MAX_TENSORS = 8;
struct SparseAdamArgs {
  float *p[SPARSE_ADAM_MAX_TENSORS];
  const float *g[SPARSE_ADAM_MAX_TENSORS];
  float *m[SPARSE_ADAM_MAX_TENSORS], *v[SPARSE_ADAM_MAX_TENSORS];
  int row_len[SPARSE_ADAM_MAX_TENSORS];
  float step_size[SPARSE_ADAM_MAX_TENSORS];
  int64_t first[SPARSE_ADAM_MAX_TENSORS + 1];   // element ranges of the tensors in the flat index space
  int n;
  float omb1, omb2, eps;
};
__global__ void __launch_bounds__(256)
sparse_adam_kernel(SparseAdamArgs a, const uint8_t *__restrict__ visible) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.first[a.n]) return;
  int t = 0;
  while (i >= a.first[t + 1]) ++t;
  const int64_t e = i - a.first[t];
  const int seen = visible[e / a.row_len[t]];
  if (!seen) return;
  const float g = a.g[t][e] * (float)seen;
  float m = a.m[t][e], v = a.v[t][e];
  m = m + (g - m) * a.omb1;
  v = v + (g * g - v) * a.omb2;
  a.m[t][e] = m;
  a.v[t][e] = v;
  a.p[t][e] = a.p[t][e] + (-a.step_size[t]) * (m / (sqrtf(v) + a.eps));
}
}  // namespace gsr

extern "C" int gsr_sparse_adam_step(int n, int64_t rows, const uint8_t *visible, void *const *params,
                                    const void *const *grads, void *const *exp_avg, void *const *exp_avg_sq,
                                    const int32_t *row_len, const float *step_size, double beta1, double beta2,
                                    double eps, void *stream) {
  GSR_REQUIRE(n >= 0 && n <= gsr::SPARSE_ADAM_MAX_TENSORS && rows >= 0, "sparse_adam_step: n=%d rows=%lld", n, (long long)rows);
  if (n == 0 || rows == 0) return GSR_OK;
  GSR_REQUIRE(visible && params && grads && exp_avg && exp_avg_sq && row_len && step_size, "sparse_adam_step: null array");
  gsr::SparseAdamArgs a;
  a.first[0] = 0;
  for (int t = 0; t < n; ++t) {
    GSR_REQUIRE(params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t] && row_len[t] > 0, "sparse_adam_step: tensor %d", t);
    a.p[t] = (float *)params[t];
    a.g[t] = (const float *)grads[t];
    a.m[t] = (float *)exp_avg[t];
    a.v[t] = (float *)exp_avg_sq[t];
    a.row_len[t] = row_len[t];
    a.step_size[t] = step_size[t];
    a.first[t + 1] = a.first[t] + rows * row_len[t];
  }
  a.n = n;
  a.omb1 = (float)(1.0 - beta1);
  a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  const int64_t blocks = gsr::ceil_div64(a.first[n], 256);
  GSR_REQUIRE(blocks < 2147483647LL, "sparse_adam_step: too many elements");
  hipLaunchKernelGGL(gsr::sparse_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, visible);
  GSR_CHECK_LAUNCH("sparse_adam_step");
  return GSR_OK;
}

extern "C" int gsr_inject_noise(int N, float *means, const float *quats, const float *log_scales,
                                const float *logit_opacities, const float *noise, float scaler,
                                void *stream) {
  GSR_REQUIRE(N >= 0, "inject_noise: bad N");
  if (N == 0) return GSR_OK;
  GSR_REQUIRE(means && quats && log_scales && logit_opacities && noise, "inject_noise: null pointer");
  hipLaunchKernelGGL(gsr::inject_noise_kernel, dim3(gsr::ceil_div(N, 256)), dim3(256), 0,
                     (hipStream_t)stream, N, means, quats, log_scales, logit_opacities, noise,
                     scaler);
  GSR_CHECK_LAUNCH("inject_noise");
  return GSR_OK;
}

extern "C" int gsr_inverse4x4(int C, const float *in, float *out, float *in_translation,
                              float *out_translation, void *stream) {
  GSR_REQUIRE(C >= 0, "inverse4x4: bad C");
  if (C == 0) return GSR_OK;
  GSR_REQUIRE(in && out, "inverse4x4: null pointer");
  hipLaunchKernelGGL(gsr::inverse4x4_kernel, dim3(gsr::ceil_div(C, 64)), dim3(64), 0,
                     (hipStream_t)stream, C, in, out, in_translation, out_translation);
  GSR_CHECK_LAUNCH("inverse4x4");
  return GSR_OK;
}

// params/grads/exp_avg/exp_avg_sq: HOST arrays of n device pointers (16-byte
// aligned tensors); numel, step_size (= lr/(1-beta1^t)), bc2_sqrt
// (= sqrt(1-beta2^t)): HOST arrays of n entries. n <= 8.
extern "C" int gsr_adam_step(int n, void *const *params, const void *const *grads,
                             void *const *exp_avg, void *const *exp_avg_sq,
                             const int64_t *numel, const float *step_size,
                             const float *bc2_sqrt, double beta1_d, double beta2_d, double eps_d,
                             void *stream) {
  const float beta1 = (float)beta1_d, beta2 = (float)beta2_d, eps = (float)eps_d;
  GSR_REQUIRE(n >= 0 && n <= gsr::ADAM_MAX_TENSORS, "adam_step: n=%d (max %d)", n,
              gsr::ADAM_MAX_TENSORS);
  if (n == 0) return GSR_OK;
  GSR_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && step_size && bc2_sqrt,
              "adam_step: null array");
  gsr::AdamArgs a;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    GSR_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && numel[i] >= 0,
                "adam_step: tensor %d has a null pointer", i);
    GSR_REQUIRE(((uintptr_t)params[i] | (uintptr_t)grads[i] | (uintptr_t)exp_avg[i] |
                 (uintptr_t)exp_avg_sq[i]) % 16 == 0,
                "adam_step: tensor %d is not 16-byte aligned", i);
    a.p[i] = (float *)params[i];
    a.g[i] = (const float *)grads[i];
    a.m[i] = (float *)exp_avg[i];
    a.v[i] = (float *)exp_avg_sq[i];
    a.numel[i] = numel[i];
    a.step_size[i] = step_size[i];
    a.bc2_sqrt[i] = bc2_sqrt[i];
    a.block_start[i] = blocks;
    int64_t nb = gsr::ceil_div64(numel[i], gsr::ADAM_ELEMS_PER_BLOCK);
    GSR_REQUIRE(blocks + nb < 2147483647LL, "adam_step: too many elements");
    blocks += (int)nb;
  }
  a.block_start[n] = blocks;
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.omb1 = (float)(1.0 - (double)beta1_d);
  a.omb2 = (float)(1.0 - (double)beta2_d);
  a.n = n;
  if (blocks == 0) return GSR_OK;
  hipLaunchKernelGGL(gsr::adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  GSR_CHECK_LAUNCH("adam_step");
  return GSR_OK;
}
