"""CPU tests of the densification strategy (F2): parameter / optimizer-state
bookkeeping of duplicate, split, remove, reset_opa and the refine schedule."""
import importlib
import math

import torch

S = importlib.import_module("3dgs_monocular_depth_init_amd.strategy")


def _make(N=50, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = {"means": (N, 3), "scales": (N, 3), "quats": (N, 4), "opacities": (N,),
              "sh0": (N, 1, 3), "shN": (N, 15, 3)}
    params = torch.nn.ParameterDict({k: torch.nn.Parameter(torch.randn(s, generator=g)) for k, s in shapes.items()})
    opts = {k: torch.optim.Adam([{"params": params[k], "lr": 1e-2, "name": k}]) for k in params}
    for k, p in params.items():
        p.grad = torch.randn(p.shape, generator=g)
    for o in opts.values():
        o.step()
    return params, opts


def _consistent(params, opts, n):
    for k, p in params.items():
        assert p.shape[0] == n, k
        st = opts[k].state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
        assert opts[k].param_groups[0]["params"][0] is p


def test_duplicate_split_remove_keep_optimizer_state_aligned():
    params, opts = _make()
    state = {"grad2d": torch.arange(50.0), "count": torch.ones(50)}
    old_means = params["means"].detach().clone()
    old_avg = opts["means"].state[params["means"]]["exp_avg"].clone()
    mask = torch.zeros(50, dtype=torch.bool); mask[[3, 7]] = True
    S.duplicate(params, opts, state, mask)
    _consistent(params, opts, 52)
    assert torch.equal(params["means"][50:], old_means[[3, 7]])
    assert (opts["means"].state[params["means"]]["exp_avg"][50:] == 0).all()
    assert torch.equal(opts["means"].state[params["means"]]["exp_avg"][:50], old_avg)
    assert torch.equal(state["grad2d"][50:], torch.tensor([3.0, 7.0]))

    mask = torch.zeros(52, dtype=torch.bool); mask[[0, 10, 51]] = True
    scales_before = params["scales"].detach().clone()
    S.split(params, opts, state, mask, generator=torch.Generator().manual_seed(1))
    _consistent(params, opts, 52 - 3 + 6)
    # children have scales / 1.6 and come in pairs after the untouched ones
    assert torch.allclose(params["scales"][-6:-3], scales_before[[0, 10, 51]] - math.log(1.6), atol=1e-6)
    assert (opts["scales"].state[params["scales"]]["exp_avg"][-6:] == 0).all()
    assert state["count"].shape[0] == 55

    mask = torch.zeros(55, dtype=torch.bool); mask[:5] = True
    S.remove(params, opts, state, mask)
    _consistent(params, opts, 50)
    assert float(opts["means"].state[params["means"]]["step"]) == 1

    S.reset_opa(params, opts, state, value=0.01)
    assert (torch.sigmoid(params["opacities"]) <= 0.01 + 1e-6).all()
    assert (opts["opacities"].state[params["opacities"]]["exp_avg"] == 0).all()
    _consistent(params, opts, 50)


def test_refine_schedule_and_statistics():
    params, opts = _make(N=40)
    with torch.no_grad():
        params["scales"].fill_(math.log(0.001))            # all "small" -> duplicates
        params["opacities"].fill_(2.0)
    strat = S.DefaultStrategy(refine_start_iter=2, refine_every=3, reset_every=1000, verbose=False)
    strat.check_sanity(params, opts)
    state = strat.initialize_state(scene_scale=1.0)
    W = H = 64

    def fake_info(n):
        m2d = torch.zeros(1, n, 2, requires_grad=True)
        radii = torch.ones(1, n, 2, dtype=torch.int32)
        radii[0, n // 2:] = 0                               # second half invisible
        info = {"means2d": m2d, "radii": radii, "width": W, "height": H, "n_cameras": 1}
        strat.step_pre_backward(params, opts, state, 0, info)
        (m2d * torch.tensor([1.0, 0.0])).sum().backward()   # grad = (1,0) for every Gaussian
        return info

    n = 40
    for step in range(1, 4):
        info = fake_info(n)
        strat.step_post_backward(params, opts, state, step, info)
        if step < 3:
            # norm of (1 * W/2, 0) accumulated for the visible half only
            assert torch.allclose(state["grad2d"][: n // 2], torch.full((n // 2,), step * W / 2.0))
            assert (state["count"][n // 2:] == 0).all()
    # step 3: > refine_start, % refine_every == 0 -> visible half (grad 32 > 2e-4, small) duplicated
    assert len(params["means"]) == 60
    assert (state["grad2d"] == 0).all() and state["count"].shape[0] == 60
    _consistent(params, opts, 60)


def test_prune_low_opacity():
    params, opts = _make(N=30)
    with torch.no_grad():
        params["opacities"][:10] = -10.0                    # sigmoid << prune_opa
        params["scales"].fill_(math.log(0.001))
    strat = S.DefaultStrategy(refine_start_iter=0, refine_every=1, reset_every=1000, grow_grad2d=1e9)
    state = strat.initialize_state()
    m2d = torch.zeros(1, 30, 2, requires_grad=True)
    info = {"means2d": m2d, "radii": torch.ones(1, 30, 2, dtype=torch.int32), "width": 8, "height": 8,
            "n_cameras": 1}
    strat.step_pre_backward(params, opts, state, 1, info)
    m2d.sum().backward()
    strat.step_post_backward(params, opts, state, 1, info)
    assert len(params["means"]) == 20
    _consistent(params, opts, 20)
