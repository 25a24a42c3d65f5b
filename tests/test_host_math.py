"""CPU check of the hand-derived projection / SH math in csrc/gs_math.h: the
header is compiled for the host with g++ and compared with the autograd oracle
(oracle/rasterization_oracle.py). No GPU involved; the host build is a test
harness only."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import rasterization_oracle as O

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "host_math" / "host_math.cpp"
INC = ROOT / "3dgs_monocular_depth_init_amd" / "csrc"


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    out = tmp_path_factory.mktemp("hm") / "libhostmath.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", f"-I{INC}", str(SRC), "-o", str(out)],
                   check=True)
    return C.CDLL(str(out))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _scene(N, seed, W=160, H=120):
    g = torch.Generator().manual_seed(seed)
    means = (torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor([1.6, 1.2, 0.6])
    quats = torch.rand(N, 4, generator=g) + 0.05
    scales = torch.exp(torch.randn(N, 3, generator=g) * 0.5 - 3.0)
    opac = torch.rand(N, generator=g) * 0.95 + 0.02
    vm = torch.eye(4)
    th = 0.2
    vm[:3, :3] = torch.tensor([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    vm[:3, 3] = torch.tensor([0.1, -0.05, 2.0])
    K = torch.tensor([[150.0, 0, W / 2 + 3], [0, 140.0, H / 2 - 2], [0, 0, 1]])
    return means, quats, scales, opac, vm, K, W, H


@pytest.mark.parametrize("comp", [False, True])
def test_projection_forward_and_backward(hm, comp):
    N = 400
    means, quats, scales, opac, vm, K, W, H = _scene(N, 3)
    means_t, quats_t, scales_t = (t.clone().requires_grad_(True) for t in (means, quats, scales))
    covs = O.quat_scale_to_covar(quats_t, scales_t)
    radii, m2d, dep, con, cmp_ = O.project_gaussians(
        means_t, covs, vm[None], K[None], W, H, 0.3, 0.01, 1e10, 0.0, opac, comp_scales_opacity=comp)
    f = lambda t: np.ascontiguousarray(t.detach().numpy().astype(np.float32))
    o_r = np.zeros((N, 2), np.int32); o_m = np.zeros((N, 2), np.float32)
    o_d = np.zeros(N, np.float32); o_c = np.zeros((N, 3), np.float32); o_k = np.zeros(N, np.float32)
    a = [f(means), f(quats), f(scales), f(opac), f(vm), f(K)]
    hm.hm_project_fwd(N, *map(_p, a), W, H, C.c_float(0.3), C.c_float(0.01), C.c_float(1e10),
                      C.c_float(0.0), int(comp), _p(o_r), _p(o_m), _p(o_d), _p(o_c), _p(o_k))
    vis = (radii[0] > 0).all(-1).numpy()
    assert vis.sum() > 50 and (~vis).sum() > 5           # both branches exercised
    assert np.array_equal(o_r, radii[0].numpy())
    np.testing.assert_allclose(o_m[vis], m2d[0].detach().numpy()[vis], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(o_d[vis], dep[0].detach().numpy()[vis], rtol=1e-6)
    np.testing.assert_allclose(o_c[vis], con[0].detach().numpy()[vis], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(o_k[vis], cmp_[0].detach().numpy()[vis], rtol=1e-4, atol=1e-6)

    # backward: random cotangents on the visible pairs
    g = torch.Generator().manual_seed(7)
    vmask = torch.from_numpy(vis)
    v_m = torch.randn(N, 2, generator=g) * vmask[:, None]
    v_d = torch.randn(N, generator=g) * vmask
    v_c = torch.randn(N, 3, generator=g) * vmask[:, None]
    v_k = (torch.randn(N, generator=g) * vmask) if comp else torch.zeros(N)
    loss = (m2d[0] * v_m).sum() + (dep[0] * v_d).sum() + (con[0] * v_c).sum() + (cmp_[0] * v_k).sum()
    loss.backward()
    g_m = np.zeros((N, 3), np.float32); g_q = np.zeros((N, 4), np.float32); g_s = np.zeros((N, 3), np.float32)
    hm.hm_project_bwd(N, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[4]), _p(a[5]), W, H, C.c_float(0.3),
                      _p(o_r), _p(f(v_m)), _p(f(v_d)), _p(f(v_c)), _p(f(v_k)), _p(g_m), _p(g_q), _p(g_s))
    for got, ref, name in ((g_m, means_t.grad, "means"), (g_q, quats_t.grad, "quats"),
                           (g_s, scales_t.grad, "scales")):
        ref = ref.numpy()
        scale = np.abs(ref).max()
        err = np.abs(got - ref).max() / scale
        assert err < 1e-3, f"{name}: rel err {err}"


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_sh_forward_and_backward(hm, degree):
    N = 300
    g = torch.Generator().manual_seed(degree)
    dirs = torch.randn(N, 3, generator=g) * 2.0
    coeffs = torch.randn(N, 16, 3, generator=g)
    v_out = torch.randn(N, 3, generator=g)
    d_t, c_t = dirs.clone().requires_grad_(True), coeffs.clone().requires_grad_(True)
    ref = O.eval_sh(degree, d_t, c_t)
    (ref * v_out).sum().backward()
    f = lambda t: np.ascontiguousarray(t.detach().numpy().astype(np.float32))
    out = np.zeros((N, 3), np.float32)
    hm.hm_sh_fwd(N, degree, _p(f(dirs)), _p(f(coeffs)), _p(out))
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    v_c = np.zeros((N, 16, 3), np.float32); v_d = np.zeros((N, 3), np.float32)
    hm.hm_sh_bwd(N, degree, _p(f(dirs)), _p(f(coeffs)), _p(f(v_out)), _p(v_c), _p(v_d))
    np.testing.assert_allclose(v_c, c_t.grad.numpy(), rtol=1e-4, atol=1e-5)
    ref_d = d_t.grad.numpy() if d_t.grad is not None else np.zeros((N, 3), np.float32)
    np.testing.assert_allclose(v_d, ref_d, rtol=1e-3, atol=1e-4)
