"""CPU checks of oracle/rasterization_oracle.py itself (no GPU, no HIP).

The arithmetic of this path lives in gsplat 1.5.2, which is neither under the
reference nor installable here, and the reference holds no fixtures for it
(SURVEY.md section 8c): the oracle stays "parity unpinned" with respect to real gsplat
output. What CAN be pinned is that the restatement is the published math, by
checking each stage against an INDEPENDENT formulation:

  * quaternion -> rotation        vs scipy.spatial.transform.Rotation
  * SH basis (degree <= 3)        vs scipy.special.sph_harm_y (real SH, the
                                     3DGS sign convention = (-1)^m times the
                                     Condon-Shortley-free table)
  * rgb_to_sh constant            vs runner_utils.py:149-151 (C0)
  * EWA projection                vs autograd Jacobian of the pinhole map and
                                     a Monte-Carlo second moment of projected samples
  * tile lists                    vs a per-tile brute-force overlap test
  * compositing                   vs a scalar per-pixel loop over ALL Gaussians
                                     in depth order (no tiles, no vectorisation)
  * gradients                     vs torch.autograd.gradcheck in fp64
                                     (<= 64 Gaussians, 32x32)
"""
import math

import numpy as np
import pytest
import torch

from oracle import rasterization_oracle as O
from tests import scenes


def test_quat_to_rotmat_matches_scipy():
    from scipy.spatial.transform import Rotation
    g = torch.Generator().manual_seed(0)
    q = torch.randn(50, 4, generator=g, dtype=torch.float64)
    R = O.quat_to_rotmat(q).numpy()
    qn = (q / q.norm(dim=-1, keepdim=True)).numpy()
    R_ref = Rotation.from_quat(qn[:, [1, 2, 3, 0]]).as_matrix()      # scipy is xyzw
    np.testing.assert_allclose(R, R_ref, atol=1e-12)
    # covariance = R S S^T R^T
    s = torch.rand(50, 3, generator=g, dtype=torch.float64) + 0.1
    cov = O.quat_scale_to_covar(q, s).numpy()
    ref = R_ref @ (s.numpy()[:, :, None] ** 2 * np.eye(3)) @ R_ref.transpose(0, 2, 1)
    np.testing.assert_allclose(cov, ref, atol=1e-12)


def test_sh_basis_matches_real_spherical_harmonics():
    from scipy.special import sph_harm_y
    g = torch.Generator().manual_seed(1)
    d = torch.randn(200, 3, generator=g, dtype=torch.float64)
    dn = (d / d.norm(dim=-1, keepdim=True)).numpy()
    theta = np.arccos(np.clip(dn[:, 2], -1, 1))                      # polar
    phi = np.arctan2(dn[:, 1], dn[:, 0])                             # azimuth
    k = 0
    for l in range(4):
        for m in range(-l, l + 1):
            coeffs = torch.zeros(200, 16, 3, dtype=torch.float64)
            coeffs[:, k, :] = 1.0
            basis = O.eval_sh(3, d, coeffs)[:, 0].numpy()
            Y = sph_harm_y(l, abs(m), theta, phi)                    # complex, with Condon-Shortley phase
            ref = Y.real if m == 0 else math.sqrt(2.0) * (Y.imag if m < 0 else Y.real)
            np.testing.assert_allclose(basis, ref, atol=1e-12, err_msg=f"l={l} m={m}")
            k += 1
    # lower degrees read only the first (d+1)^2 coefficients
    full = torch.randn(200, 16, 3, generator=g, dtype=torch.float64)
    for deg in range(3):
        cut = full.clone()
        cut[:, (deg + 1) ** 2:] = 0
        assert torch.allclose(O.eval_sh(deg, d, full), O.eval_sh(3, d, cut), atol=1e-14)
    # runner_utils.py:149-151: rgb -> sh0 -> colour round trip through C0 and the +0.5 offset
    rgb = torch.rand(200, 3, generator=g, dtype=torch.float64)
    sh0 = (rgb - 0.5) / scenes.SH_C0
    c = torch.zeros(200, 16, 3, dtype=torch.float64)
    c[:, 0] = sh0
    assert torch.allclose(O.eval_sh(0, d, c) + 0.5, rgb, atol=1e-12)


def _pinhole(p, fx, fy, cx, cy):
    return torch.stack([fx * p[0] / p[2] + cx, fy * p[1] / p[2] + cy])


def test_projection_matches_jacobian_and_sample_moments():
    g = torch.Generator().manual_seed(2)
    N = 6
    means = torch.randn(N, 3, generator=g, dtype=torch.float64) * 0.3
    quats = torch.randn(N, 4, generator=g, dtype=torch.float64)
    scl = torch.rand(N, 3, generator=g, dtype=torch.float64) * 0.02 + 0.01
    vm, K = scenes.cameras([3, 40], width=256, height=192, f=300.0)
    vm, K = vm.double(), K.double()
    cov = O.quat_scale_to_covar(quats, scl)
    radii, m2d, depths, conics, comps = O.project_gaussians(means, cov, vm, K, 256, 192, opacities=torch.full((N,), 0.9, dtype=torch.float64))
    for c in range(2):
        R, t = vm[c, :3, :3], vm[c, :3, 3]
        fx, fy, cx, cy = K[c, 0, 0], K[c, 1, 1], K[c, 0, 2], K[c, 1, 2]
        for i in range(N):
            pc = R @ means[i] + t
            J = torch.autograd.functional.jacobian(lambda p: _pinhole(p, fx, fy, cx, cy), pc)
            cov2 = J @ (R @ cov[i] @ R.T) @ J.T
            det0 = torch.det(cov2)
            cov2b = cov2 + 0.3 * torch.eye(2, dtype=torch.float64)
            inv = torch.linalg.inv(cov2b)
            assert torch.allclose(m2d[c, i], _pinhole(pc, fx, fy, cx, cy), atol=1e-10)
            assert torch.allclose(depths[c, i], pc[2], atol=1e-12)
            assert torch.allclose(conics[c, i], torch.stack([inv[0, 0], inv[0, 1], inv[1, 1]]), rtol=1e-10)
            assert torch.allclose(comps[c, i], torch.sqrt(det0 / torch.det(cov2b)), rtol=1e-10)
            # opacity-aware extent: min(3.33, sqrt(2 ln(255 o))) sigmas per axis, ceil'd
            ext = min(O.EXTENT_MAX, math.sqrt(2 * math.log(0.9 * 255)))
            assert int(radii[c, i, 0]) == math.ceil(ext * math.sqrt(float(cov2b[0, 0])))
            assert int(radii[c, i, 1]) == math.ceil(ext * math.sqrt(float(cov2b[1, 1])))
    # Monte-Carlo: the second moment of projected samples of a SMALL Gaussian -> J Sigma J^T
    i, c = 0, 0
    L = torch.linalg.cholesky(cov[i] * 1e-4)         # tiny, so the local linearisation is exact to O(1e-4)
    smp = means[i] + (torch.randn(400_000, 3, generator=g, dtype=torch.float64) @ L.T)
    pc = smp @ vm[c, :3, :3].T + vm[c, :3, 3]
    uv = torch.stack([K[c, 0, 0] * pc[:, 0] / pc[:, 2], K[c, 1, 1] * pc[:, 1] / pc[:, 2]], -1)
    emp = torch.cov(uv.T) * 1e4
    inv = torch.linalg.inv(emp + 0.3 * torch.eye(2, dtype=torch.float64))
    assert torch.allclose(conics[c, i], torch.stack([inv[0, 0], inv[0, 1], inv[1, 1]]), rtol=2e-2)


def test_culling_rules():
    vm = torch.eye(4)[None]
    K = torch.tensor([[[100.0, 0, 32], [0, 100.0, 32], [0, 0, 1]]])
    means = torch.tensor([[0, 0, 2.0],        # visible
                          [0, 0, 0.005],      # nearer than near_plane
                          [0, 0, -1.0],       # behind
                          [50.0, 0, 2.0],     # far off-screen
                          [0, 0, 2.0]])       # opacity below 1/255
    quats = torch.tensor([[1.0, 0, 0, 0]]).repeat(5, 1)
    scl = torch.full((5, 3), 0.05)
    op = torch.tensor([0.5, 0.5, 0.5, 0.5, 0.003])
    cov = O.quat_scale_to_covar(quats, scl)
    radii = O.project_gaussians(means, cov, vm, K, 64, 64, opacities=op)[0][0]
    assert (radii[0] > 0).all() and (radii[1:] == 0).all()
    # radius_clip: a footprint no wider than the clip is dropped
    r_small = O.project_gaussians(means[:1], cov[:1] * 1e-6, vm, K, 64, 64, radius_clip=3.0, opacities=op[:1])[0]
    assert (r_small == 0).all()


def test_tile_lists_against_bruteforce():
    sc, vm, K, W, H = scenes.config_c1(N=300, seed=5)
    W, H = 100, 72                                  # ragged: 7 x 5 tiles, last ones partial
    K = K.clone(); K[0, 0, 2] = 50; K[0, 1, 2] = 36
    cov = O.quat_scale_to_covar(sc["quats"], sc["scales"])
    radii, m2d, depths, conics, _ = O.project_gaussians(sc["means"], cov, vm, K, W, H, opacities=sc["opacities"])
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = O.isect_tiles(m2d, radii, depths, 16, tw, th)
    tpg2, ids2, flat2 = O.isect_tiles_fast(m2d, radii, depths, 16, tw, th)
    assert torch.equal(tpg, tpg2) and torch.equal(ids, ids2) and torch.equal(flat, flat2)
    offs = O.isect_offset_encode(ids, 1, tw, th).reshape(-1).tolist() + [flat.numel()]
    for t in range(tw * th):
        ty, tx = divmod(t, tw)
        want = []
        for i in range(300):
            rx, ry = int(radii[0, i, 0]), int(radii[0, i, 1])
            if rx <= 0 or ry <= 0:
                continue
            x, y = float(m2d[0, i, 0]), float(m2d[0, i, 1])
            # the Gaussian's pixel rectangle [x-rx, x+rx] x [y-ry, y+ry] overlaps the tile (open ends)
            if x - rx < (tx + 1) * 16 and x + rx > tx * 16 and y - ry < (ty + 1) * 16 and y + ry > ty * 16:
                want.append(i)
        got = flat[offs[t]:offs[t + 1]].tolist()
        assert sorted(got) == sorted(want), f"tile {t}"
        d = depths[0, got]
        assert (d[1:] >= d[:-1]).all()


def _composite_scalar(m2d, conics, colors, op, depths, radii, W, H, bg=None):
    """Appendix A.4 written as a scalar loop: every pixel visits EVERY visible Gaussian
    in depth order (index breaks ties)."""
    N, D = colors.shape
    order = sorted([i for i in range(N) if radii[i, 0] > 0 and radii[i, 1] > 0],
                   key=lambda i: (float(np.float32(float(depths[i]))), i))   # keys hold fp32 depth bits
    img = np.zeros((H, W, D)); alpha = np.zeros((H, W))
    m2d, conics, colors, op = (a.double().numpy() for a in (m2d, conics, colors, op))
    for y in range(H):
        for x in range(W):
            T = 1.0
            px, py = x + 0.5, y + 0.5
            for i in order:
                dx, dy = m2d[i, 0] - px, m2d[i, 1] - py
                sigma = 0.5 * (conics[i, 0] * dx * dx + conics[i, 2] * dy * dy) + conics[i, 1] * dx * dy
                a = min(O.ALPHA_MAX, op[i] * math.exp(-sigma))
                if sigma < 0 or a < O.ALPHA_THRESHOLD:
                    continue
                Tn = T * (1 - a)
                if Tn <= O.T_THRESHOLD:
                    break
                img[y, x] += colors[i] * a * T
                T = Tn
            alpha[y, x] = 1 - T
            if bg is not None:
                img[y, x] += T * bg
    return img, alpha


@pytest.mark.parametrize("seed,opaque", [(7, False), (8, True)])
def test_compositing_against_scalar_loop(seed, opaque):
    sc, vm, K, _, _ = scenes.config_c1(N=120, seed=seed)
    W, H = 40, 24
    K = K.clone(); K[0, 0, 0] = K[0, 1, 1] = 40.0; K[0, 0, 2] = 20; K[0, 1, 2] = 12
    sc = {k: v.double() for k, v in sc.items()}
    if opaque:                      # force early termination (T <= 1e-4) to occur
        sc["opacities"] = torch.full_like(sc["opacities"], 0.995)
        sc["scales"] = sc["scales"] * 4
    colors = torch.rand(120, 3, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
    bg = torch.tensor([[0.1, 0.2, 0.3]], dtype=torch.float64)
    rc, ra, meta = O.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], colors,
                                   vm.double(), K.double(), W, H, backgrounds=bg)
    img, alpha = _composite_scalar(meta["means2d"][0], meta["conics"][0], colors, sc["opacities"],
                                   meta["depths"][0], meta["radii"][0], W, H, bg=bg[0].numpy())
    np.testing.assert_allclose(rc[0].numpy(), img, atol=1e-12)
    np.testing.assert_allclose(ra[0, ..., 0].numpy(), alpha, atol=1e-12)
    if opaque:
        assert alpha.max() > 1 - 1e-3
    # expected depth channel (runner.py:479-482): sum(w z) / max(alpha, 1e-10)
    rc4, ra4, _ = O.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], colors,
                                  vm.double(), K.double(), W, H, render_mode="RGB+ED")
    zimg, _ = _composite_scalar(meta["means2d"][0], meta["conics"][0], meta["depths"][0][:, None],
                                sc["opacities"], meta["depths"][0], meta["radii"][0], W, H)
    np.testing.assert_allclose(rc4[0, ..., 3].numpy(), zimg[..., 0] / np.maximum(alpha, 1e-10), atol=1e-10)


def test_single_gaussian_closed_form():
    # isotropic Gaussian on the optical axis: Sigma2 = (f s / z)^2 I + 0.3 I
    f, z, s, o = 64.0, 2.0, 0.05, 0.8
    vm = torch.eye(4, dtype=torch.float64)[None]
    K = torch.tensor([[[f, 0, 16.5], [0, f, 16.5], [0, 0, 1]]], dtype=torch.float64)
    rc, ra, _ = O.rasterization(torch.tensor([[0, 0, z]], dtype=torch.float64),
                                torch.tensor([[1.0, 0, 0, 0]], dtype=torch.float64),
                                torch.full((1, 3), s, dtype=torch.float64), torch.tensor([o], dtype=torch.float64),
                                torch.tensor([[0.2, 0.5, 1.0]], dtype=torch.float64), vm, K, 33, 33)
    var = (f * s / z) ** 2 + 0.3
    yy, xx = np.meshgrid(np.arange(33) + 0.5 - 16.5, np.arange(33) + 0.5 - 16.5, indexing="ij")
    a = o * np.exp(-0.5 * (xx ** 2 + yy ** 2) / var)
    a = np.where(a >= 1 / 255, np.minimum(a, 0.999), 0.0)
    np.testing.assert_allclose(ra[0, ..., 0].numpy(), a, atol=1e-12)
    np.testing.assert_allclose(rc[0].numpy(), a[..., None] * np.array([0.2, 0.5, 1.0]), atol=1e-12)


def test_gradcheck_fp64():
    """the oracle's gradients are the derivative of its forward."""
    sc, vm, K, _, _ = scenes.config_c1(N=48, seed=11)
    W, H = 32, 32
    K = K.clone(); K[0, 0, 0] = K[0, 1, 1] = 40.0; K[0, 0, 2] = K[0, 1, 2] = 16
    p = {k: v.double().requires_grad_(True) for k, v in sc.items()}
    vm, K = vm.double(), K.double()
    g = torch.Generator().manual_seed(12)
    w = torch.randn(1, H, W, 4, generator=g, dtype=torch.float64)
    wa = torch.randn(1, H, W, 1, generator=g, dtype=torch.float64)

    def f(means, quats, scales, opacities, sh0, shN):
        rc, ra, _ = O.rasterization(means, quats, scales, opacities, torch.cat([sh0, shN], 1), vm, K,
                                    W, H, sh_degree=3, render_mode="RGB+ED")
        return (rc * w).sum() + (ra * wa).sum()

    args = tuple(p[k] for k in ("means", "quats", "scales", "opacities", "sh0", "shN"))
    # eps small enough that no pair crosses the 1/255 cut-off between the two evaluations
    assert torch.autograd.gradcheck(f, args, eps=1e-7, atol=1e-5, rtol=1e-3, nondet_tol=0.0)
