// adam_math.h -- the Adam update shared by gsr_adam_step (train_ops.hip) and the
// optimizer-in-backward variant of the projection backward (project.hip).
#pragma once
#include "common.h"

namespace gsr {

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float omb1,
                                         float beta2, float omb2, float eps, float step_size,
                                         float bc2_sqrt) {
  // same operation order as torch.optim.Adam (_single_tensor_adam)
  m = m + (g - m) * omb1;                            // exp_avg.lerp_(grad, 1 - beta1)
  v = v * beta2 + omb2 * g * g;                      // mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);                   // addcdiv_(exp_avg, denom, -step_size)
}

// Parameter tensors of the optimizer-in-backward path, in this order.
enum { AF_MEANS = 0, AF_QUATS, AF_SCALES, AF_OPAC, AF_SH0, AF_SHN, AF_COUNT };
struct AdamFused {
  float *p[AF_COUNT];          // parameters (updated in place)
  float *m[AF_COUNT];          // exp_avg
  float *v[AF_COUNT];          // exp_avg_sq
  float step_size[AF_COUNT];   // lr / (1 - beta1^t)
  float bc2_sqrt[AF_COUNT];    // sqrt(1 - beta2^t)
  float beta2, eps, omb1, omb2;
  // What the "mcmc" preset adds to a step (trainer.py:83-92, runner.py:535-545, 649-656), folded into the same pass:
  // the position noise of MCMCStrategy (gsplat inject_noise_to_position: means += covar . (randn * gate(1 - opacity)
  // * lr * noise_lr), from the PRE-update parameters, added before the Adam update of the means) and the gradients
  // of the two regularisers opacity_reg * mean(sigmoid(o)) and scale_reg * mean(exp(s)).
  const float *noise;          // NULL, or [N,3] standard-normal draws
  float noise_scale;           // lr(means) * noise_lr
  float opacity_reg;           // opacity_reg / N        (x sigmoid' = s (1 - s) in the kernel)
  float scale_reg;             // scale_reg / (3 N)      (x exp(s) in the kernel)
  // DefaultStrategy's per-step statistics (gsplat DefaultStrategy._update_state, runner.py:639-647), taken while the
  // gradient row and the radii are in registers anyway: grad2d[i] += |(g.x sx, g.y sy)|, count[i] += 1 per camera that
  // renders Gaussian i, radii[i] = max(radii[i], max(rx, ry) / max(W, H)). NULL: not wanted.
  float *stat_grad2d, *stat_count, *stat_radii;
  float stat_sx, stat_sy, stat_inv_max_wh;
  int stat_abs;                // 1: from the absgrad fields of the row (GSR_GR_ABS) instead of GSR_GR_MEAN2D
};

// gate(1 - opacity) * scaler of gsplat's inject_noise_to_position (op_sigmoid with k = 100, x0 = 0.995)
__device__ __forceinline__ float mcmc_noise_gate(float opacity_act, float scaler) {
  return 1.0f / (1.0f + expf(-100.0f * ((1.0f - opacity_act) - 0.995f))) * scaler;
}

}  // namespace gsr
