"""CPU oracle for the MCMC strategy's two per-Gaussian ops.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the ops live in gsplat 1.5.2 (`gsplat.relocation.compute_relocation`,
`gsplat.strategy.ops.inject_noise_to_position`), absent here; the reference only calls
the strategy (gs_init_compare/runner.py:649-658). Restated from the published method
("3D Gaussian Splatting as Markov Chain Monte Carlo", eq. 9 and section 4.1)."""
import math

import torch


def compute_relocation(opacities, scales, ratios, n_max: int = 51):
    """Plain-python restatement, fp64. opacities [n], scales [n,3], ratios [n] int."""
    n = opacities.shape[0]
    new_o = torch.zeros(n, dtype=torch.float64)
    new_s = torch.zeros(n, 3, dtype=torch.float64)
    for i in range(n):
        N = int(min(max(int(ratios[i]), 1), n_max))
        o = float(opacities[i])
        no = 1.0 - (1.0 - o) ** (1.0 / N)
        denom = 0.0
        for a in range(1, N + 1):
            for k in range(a):
                denom += math.comb(a - 1, k) * ((-1) ** k / math.sqrt(k + 1)) * no ** (k + 1)
        new_o[i] = no
        new_s[i] = scales[i].double() * (o / denom)
    return new_o, new_s


def inject_noise(means, quats, log_scales, logit_opac, noise, scaler: float):
    from oracle.rasterization_oracle import quat_scale_to_covar
    op = torch.sigmoid(logit_opac.double())
    cov = quat_scale_to_covar(quats.double(), torch.exp(log_scales.double()))
    gate = 1.0 / (1.0 + torch.exp(-100.0 * ((1.0 - op) - 0.995)))
    nz = noise.double() * gate[:, None] * scaler
    return means.double() + torch.einsum("bij,bj->bi", cov, nz)
