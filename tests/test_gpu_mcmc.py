"""MCMC strategy: HIP relocation / noise kernels against
oracle/strategy_oracle.py, the method's defining property, and an end-to-end run."""
import importlib
import math

import pytest
import torch

from oracle import strategy_oracle as SO
from tests import scenes

pytestmark = pytest.mark.gpu


def S():
    return importlib.import_module("3dgs_monocular_depth_init_amd.strategy")


def test_relocation_kernel_vs_oracle_and_defining_property():
    g = torch.Generator().manual_seed(0)
    n = 500
    op = torch.rand(n, generator=g) * 0.98 + 0.01
    sc = torch.rand(n, 3, generator=g) * 0.1 + 0.001
    ratios = torch.randint(1, 60, (n,), generator=g)          # > n_max exercises the clamp
    st = S().MCMCStrategy().initialize_state()
    new_o, new_s = S().compute_relocation(op.cuda(), sc.cuda(), ratios.cuda(), st["binoms"])
    ref_o, ref_s = SO.compute_relocation(op, sc, ratios)
    assert torch.allclose(new_o.cpu().double(), ref_o, rtol=2e-5, atol=1e-7)
    # the alternating binomial sum cancels heavily for large N: fp32 keeps ~3 digits there
    small = ratios <= 12
    assert torch.allclose(new_s.cpu().double()[small], ref_s[small], rtol=2e-3)
    assert torch.isfinite(new_s).all()
    # N copies of the new opacity composite to the old one: 1 - (1 - o')^N = o
    N = ratios.clamp(1, 51).double()
    assert torch.allclose(1 - (1 - new_o.cpu().double()) ** N, op.double(), atol=2e-5)
    # N = 1 is the identity
    o1, s1 = S().compute_relocation(op.cuda(), sc.cuda(), torch.ones(n, dtype=torch.long).cuda(), st["binoms"])
    assert torch.allclose(o1.cpu(), op, atol=1e-6) and torch.allclose(s1.cpu(), sc, rtol=1e-5)


def test_inject_noise_kernel_vs_oracle():
    lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
    g = torch.Generator().manual_seed(1)
    N = 4001
    means = torch.randn(N, 3, generator=g)
    quats = torch.randn(N, 4, generator=g)
    ls = torch.randn(N, 3, generator=g) * 0.5 - 3.0
    lo = torch.randn(N, generator=g) * 3.0 - 3.0               # mostly low opacity: the gate opens
    noise = torch.randn(N, 3, generator=g)
    scaler = 1.6e-4 * 5e5
    ref = SO.inject_noise(means, quats, ls, lo, noise, scaler)
    m, qd, lsd, lod, nd = means.cuda().clone(), quats.cuda(), ls.cuda(), lo.cuda(), noise.cuda()
    lib.call("gsr_inject_noise", N, m.data_ptr(), qd.data_ptr(), lsd.data_ptr(), lod.data_ptr(),
             nd.data_ptr(), scaler, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    moved = (ref - means.double()).abs().max()
    assert moved > 1e-3                                         # the test is not vacuous
    assert torch.allclose(m.cpu().double(), ref, rtol=1e-4, atol=1e-6 + 1e-4 * float(moved))


def test_mcmc_training_relocates_dead_and_grows_to_cap():
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
    N = 2000
    sc = scenes.make_scene(N, 4, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    W, H = 96, 64
    vm, K = scenes.cameras([0, 20, 40, 60], width=W, height=H, f=90.0, dist=2.5)
    c2w, K = torch.linalg.inv(vm).cuda(), K.cuda()
    target = torch.rand(4, H, W, 3, generator=torch.Generator().manual_seed(5)).cuda()
    logit_op = torch.logit(sc["opacities"])
    logit_op[:300] = -8.0                                       # 300 dead Gaussians (opacity 3e-4)
    splats, opts = runner.create_splats_with_optimizers(
        sc["means"], torch.rand(N, 3, generator=torch.Generator().manual_seed(0)),
        torch.log(sc["scales"]), quats=sc["quats"], opacities_logit=logit_op, shN=sc["shN"])
    fused = D.fuse_optimizers(splats, opts)
    strat = S().MCMCStrategy(cap_max=2300, refine_start_iter=2, refine_every=3, refine_stop_iter=100)
    strat.check_sanity(splats, fused)
    state = strat.initialize_state()
    losses = []
    for step in range(1, 14):
        k = step % 4
        loss, info = runner.train_step(splats, fused, c2w[k:k + 1], K[k:k + 1], target[k:k + 1],
                                       step=step, strategy=strat,
                                       strategy_state=state, opacity_reg=0.01, scale_reg=0.01)
        losses.append(float(loss))
        n = len(splats["means"])
        assert all(len(p) == n for p in splats.values())
        for name, p in splats.items():
            assert torch.isfinite(p).all(), name
            st = fused[name].state.get(p, {})
            if "exp_avg" in st:
                assert st["exp_avg"].shape == p.shape
    assert len(splats["means"]) == 2300                         # grew 5 % per refinement up to the cap
    # nothing is left at the dead opacity: relocated onto alive Gaussians (>= min_opacity)
    assert float(torch.sigmoid(splats["opacities"].detach()).min()) >= strat.min_opacity * 0.5
    assert all(math.isfinite(x) for x in losses)


def _mcmc_world(n, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = {"means": (n, 3), "quats": (n, 4), "scales": (n, 3), "opacities": (n,), "sh0": (n, 1, 3), "shN": (n, 15, 3)}
    params = torch.nn.ParameterDict({k: torch.nn.Parameter((torch.randn(*s, generator=g) * 0.5).cuda()) for k, s in shapes.items()})
    params["scales"].data.sub_(3.0)
    opts = {k: torch.optim.Adam([params[k]], lr=1e-3) for k in shapes}
    for k, o in opts.items():                                   # non-trivial moments
        o.state[params[k]] = {"step": torch.tensor(7.0), "exp_avg": torch.randn(*shapes[k], generator=g).cuda(),
                              "exp_avg_sq": torch.rand(*shapes[k], generator=g).cuda()}
    return params, opts


@pytest.mark.parametrize("op", ["relocate", "sample_add", "reset_opa"])
def test_gather_launch_paths_equal_the_tensor_ops(op, monkeypatch):
    """relocate / sample_add rebuilt by `gsr_refine_gather` and reset_opa by `gsr_reset_opacity` leave
    exactly the parameters, Adam moments and strategy state of the tensor-op formulation they
    replace (gsplat.strategy.ops relocate / sample_add / reset_opa), same multinomial draws."""
    st_ = S()
    n = 3000
    outs = []
    for use_kernels in (False, True):
        params, opts = _mcmc_world(n, 5)
        state = {"binoms": st_.MCMCStrategy().initialize_state()["binoms"], "stat": torch.arange(n, dtype=torch.float32).cuda()}
        if not use_kernels:
            monkeypatch.setattr(st_, "_gather_ok", lambda p: False)
        gen = torch.Generator(device="cuda").manual_seed(11)
        if op == "relocate":
            dead = torch.zeros(n, dtype=torch.bool, device="cuda")
            dead[::7] = True
            st_.relocate(params, opts, state, dead, state["binoms"], generator=gen)
        elif op == "sample_add":
            st_.sample_add(params, opts, state, 450, state["binoms"], generator=gen)
        else:
            if not use_kernels:        # the tensor-op formulation, spelled out
                mx = math.log(0.01 / 0.99)
                st_._update_param_with_optimizer(lambda nm, p: torch.clamp(p, max=mx), lambda k, v: torch.zeros_like(v),
                                                 params, opts, names=["opacities"])
            else:
                st_.reset_opa(params, opts, state, value=0.01)
        monkeypatch.undo()
        rec = {k: p.detach().clone() for k, p in params.items()}
        for k, o in opts.items():
            s = o.state[params[k]]
            assert float(s["step"]) == 7.0
            rec[k + ".m"], rec[k + ".v"] = s["exp_avg"].clone(), s["exp_avg_sq"].clone()
        rec["stat"] = state["stat"].clone()
        outs.append(rec)
    a, b = outs
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    if op == "sample_add":
        assert len(b["means"]) == n + 450 and float(b["means.m"][n:].abs().max()) == 0.0
    if op == "reset_opa":
        assert float(b["opacities"].max()) <= math.log(0.01 / 0.99) + 1e-6 and float(b["opacities.v"].abs().max()) == 0.0


@pytest.mark.parametrize("N", [2000, 2043])      # whole waves only / a 59-Gaussian tail through the generic kernel
def test_mcmc_step_fused_into_backward_equals_reference_order(N):
    """The "mcmc" preset's step with the optimizer in the backward (gsr_project_bwd_adam_ex: Adam + the position
    noise from the pre-update parameters + the gradients of the opacity / scale regularisers in ONE pass) against the
    reference's order -- backward, regularisers through autograd, strategy.step_post_backward (noise), optimizer.step
    (runner.py:535-547, 649-679) -- on two copies of a scene, same noise generator seed, across refine steps (which
    take the reference order on both sides). Same loss values, same parameters and moments to rounding."""
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
    sc = scenes.make_scene(N, 4, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    W, H = 96, 64
    vm, K = scenes.cameras([0, 20, 40, 60], width=W, height=H, f=90.0, dist=2.5)
    c2w, K = torch.linalg.inv(vm).cuda(), K.cuda()
    target = torch.rand(4, H, W, 3, generator=torch.Generator().manual_seed(5)).cuda()

    logit_op = torch.logit(sc["opacities"])
    logit_op[:300] = math.log(0.01 / 0.99)        # nearly transparent: the noise gate is 0.38 there (1e-22 at opacity 0.5)
    logit_op[300:340] = -8.0                      # dead: relocated on the refine steps

    def world(fuse):
        splats, opts = runner.create_splats_with_optimizers(
            sc["means"], torch.rand(N, 3, generator=torch.Generator().manual_seed(0)),
            torch.log(sc["scales"]), quats=sc["quats"], opacities_logit=logit_op.clone(), shN=sc["shN"])
        fused = D.fuse_optimizers(splats, opts)
        fused.fuse_into_backward(fuse)
        strat = S().MCMCStrategy(cap_max=N, refine_start_iter=2, refine_every=5, refine_stop_iter=100, noise_lr=5e4)
        return splats, fused, strat, strat.initialize_state()

    calls = []
    L = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
    real_call = L.call
    R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")

    def spy(name, *a):
        calls.append(name)
        return real_call(name, *a)

    out = {}
    for fuse in (True, False):
        splats, fused, strat, state = world(fuse)
        losses = []
        R.call = spy
        try:
            for step in range(1, 13):
                k = step % 4
                loss, _ = runner.train_step(splats, fused, c2w[k:k + 1], K[k:k + 1], target[k:k + 1], step=step,
                                            strategy=strat, strategy_state=state, opacity_reg=0.01, scale_reg=0.01)
                losses.append(float(loss))
        finally:
            R.call = real_call
            fused.fuse_into_backward(False)
        out[fuse] = (losses, {k: v.detach().clone() for k, v in splats.items()},
                     {k: fused[k].state[splats[k]]["exp_avg_sq"].clone() for k in splats})
        if fuse:
            assert calls.count("gsr_project_bwd_adam_ex") == 10        # every step but the two refine steps (5, 10)
            calls.clear()
    (l1, p1, v1), (l0, p0, v0) = out[True], out[False]
    assert max(abs(a - b) for a, b in zip(l1, l0)) < 1e-6
    moved = (p0["means"].cpu() - sc["means"])[:300].abs().max()
    assert moved > 5e-3                                                   # the noise is far above the tolerance below
    for k in p0:
        scale = float(p0[k].abs().max())
        assert float((p1[k] - p0[k]).abs().max()) <= 2e-5 * scale + 1e-7, k
        assert float((v1[k] - v0[k]).abs().max()) <= 1e-4 * float(v0[k].abs().max()) + 1e-12, k
