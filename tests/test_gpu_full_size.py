"""Full-size checks (BASELINE configs c2: 100k Gaussians x 4 views @1080p, c4: 1M
Gaussians @1080p, c5: 2M Gaussians @1080p) through size-independent properties -- the CPU oracle cannot run
at these sizes in test time:
  * batching: C cameras in one call == C single-camera calls (forward exact,
    parameter gradients = sum of the per-camera gradients);
  * permutation invariance: shuffling the Gaussians leaves the image unchanged and
    permutes the gradients;
  * linearity: the backward is linear in the cotangent; the forward is linear in
    caller-supplied colours;
  * directional finite differences of the loss agree with the analytic gradient;
  * ranges: 0 <= alpha <= 1, colours finite, every list sorted by depth.
"""
import importlib

import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


@pytest.fixture(scope="module")
def R():
    return importlib.import_module("3dgs_monocular_depth_init_amd.rendering")


def _gpu_scene(N, seed=0):
    sc = scenes.make_scene(N, seed)
    return {k: v.cuda() for k, v in sc.items()}


def _render(R, p, vm, K, **kw):
    return R.rasterization(p["means"], p["quats"], p["scales"], p["opacities"],
                           (p["sh0"], p["shN"]), vm, K, W, H, sh_degree=3, packed=False, **kw)


def _grads(R, p, vm, K, w):
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    rc, ra, _ = _render(R, leaves, vm, K)
    (rc * w).sum().backward()
    return {k: v.grad for k, v in leaves.items()}


def test_c2_batched_cameras_equal_single_camera_calls(R):
    p = _gpu_scene(100_000)
    vm, K = scenes.cameras([0, 25, 50, 75])
    vm, K = vm.cuda(), K.cuda()
    rc, ra, meta = _render(R, p, vm, K)
    assert rc.shape == (4, H, W, 3) and torch.isfinite(rc).all()
    assert float(ra.min()) >= 0.0 and float(ra.max()) <= 1.0 + 1e-6
    singles = [_render(R, p, vm[i:i + 1], K[i:i + 1]) for i in range(4)]
    for i, (rci, rai, mi) in enumerate(singles):
        assert torch.equal(rc[i], rci[0]) and torch.equal(ra[i], rai[0])
        assert torch.equal(meta["radii"][i], mi["radii"][0])
    # lists are depth-sorted inside every tile
    off = meta["isect_offsets"].reshape(-1).long()
    ids = meta["flatten_ids"].long()
    d = meta["depths"].reshape(-1)[ids]
    tile_of = torch.bucketize(torch.arange(ids.numel(), device=ids.device), off, right=True)
    same = tile_of[1:] == tile_of[:-1]
    assert (d[1:][same] >= d[:-1][same]).all()
    # gradients: batch == sum over cameras
    g = torch.Generator().manual_seed(1)
    w = torch.randn(4, H, W, 3, generator=g).cuda()
    gb = _grads(R, p, vm, K, w)
    gs = None
    for i in range(4):
        gi = _grads(R, p, vm[i:i + 1], K[i:i + 1], w[i:i + 1])
        gs = gi if gs is None else {k: gs[k] + gi[k] for k in gs}
    for k in gb:
        rel = float((gb[k] - gs[k]).norm() / gs[k].norm().clamp_min(1e-20))
        assert rel < 1e-4, f"{k}: {rel:.2e}"


def test_c4_permutation_invariance_and_linearity(R):
    p = _gpu_scene(1_000_000)
    vm, K = scenes.cameras([7])
    vm, K = vm.cuda(), K.cuda()
    g = torch.Generator().manual_seed(2)
    w = torch.randn(1, H, W, 3, generator=g).cuda()
    # Equal fp32 depths are composited in index order (the tie-break of the sort key),
    # which a permutation changes; among 1M random depths thousands of pairs collide,
    # so drop every Gaussian whose depth is not unique before testing invariance.
    d = _render(R, p, vm, K)[2]["depths"][0]
    _, inv, cnt = torch.unique(d, return_inverse=True, return_counts=True)
    keep = cnt[inv] == 1
    assert int(keep.sum()) > 500_000
    p = {k: v[keep].contiguous() for k, v in p.items()}
    n = int(keep.sum())
    rc, ra, meta = _render(R, p, vm, K)
    g1 = _grads(R, p, vm, K, w)
    perm = torch.randperm(n, generator=g).cuda()
    q = {k: v[perm].contiguous() for k, v in p.items()}
    rc_p, ra_p, _ = _render(R, q, vm, K)
    assert float((rc_p - rc).abs().max()) <= 2e-6 and float((ra_p - ra).abs().max()) <= 2e-6
    gp = _grads(R, q, vm, K, w)
    for k in g1:
        rel = float((gp[k] - g1[k][perm]).norm() / g1[k].norm().clamp_min(1e-20))
        assert rel < 1e-4, f"{k}: {rel:.2e}"
    # the backward is linear in the cotangent
    g2 = _grads(R, p, vm, K, 2.0 * w)
    for k in g1:
        rel = float((g2[k] - 2.0 * g1[k]).norm() / g1[k].norm().clamp_min(1e-20))
        assert rel < 1e-4, f"{k}: {rel:.2e}"
    # forward linear in caller-supplied colours (non-SH path, packed on the fly)
    cols = torch.rand(n, 3, generator=torch.Generator().manual_seed(3)).cuda()
    ra1 = R.rasterization(p["means"], p["quats"], p["scales"], p["opacities"], cols, vm, K, W, H)[0]
    ra3 = R.rasterization(p["means"], p["quats"], p["scales"], p["opacities"], 3.0 * cols, vm, K, W, H)[0]
    assert float((ra3 - 3.0 * ra1).abs().max()) <= 1e-5 * float(ra3.abs().max())


@pytest.mark.parametrize("N", [1_000_000, 2_000_000], ids=["c4_1M", "c5_2M"])
def test_directional_finite_difference(R, N):
    p = _gpu_scene(N)
    vm, K = scenes.cameras([31])
    vm, K = vm.cuda(), K.cuda()
    g = torch.Generator().manual_seed(4)
    # A SMOOTH weight image: the alpha >= 1/255 cut-off makes the loss piecewise smooth
    # (pixels enter/leave a footprint with a 1/255 jump); under a white-noise weight
    # those jumps are first-order noise in the finite difference, under a smooth one
    # they cancel around each footprint.
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    w = torch.stack([torch.sin(xx / 97.0 + yy / 131.0 + c) for c in range(3)], -1)[None].cuda() / (H * W)
    grads = _grads(R, p, vm, K, w)

    def loss(pp):
        with torch.no_grad():
            return float((_render(R, pp, vm, K)[0].double() * w.double()).sum())

    for name, eps in (("means", 2e-4), ("sh0", 1e-2), ("opacities", 2e-3)):
        # random magnitudes, signs taken from the gradient: the directional derivative is then
        # a sum of same-sign terms, well above the noise floor of the finite difference (a fully
        # random direction can land on a near-cancellation -- at 2 M Gaussians it did: analytic
        # -3.4e-4 under a +-1e-3 finite-difference noise)
        d = torch.randn(p[name].shape, generator=g).cuda().abs()
        if name == "means":
            # move in the camera's image plane only: a depth change would swap the
            # compositing order of overlapping pairs (another jump the analytic
            # gradient does not and should not model)
            g_cam = grads[name] @ vm[0, :3, :3].T           # gradient in camera axes
            d = d * torch.sign(g_cam)
            d[:, 2] = 0.0
            d = d @ vm[0, :3, :3]            # world direction = R^T (dx, dy, 0)
        else:
            d = d * torch.sign(grads[name])
        plus = dict(p); plus[name] = p[name] + eps * d
        minus = dict(p); minus[name] = p[name] - eps * d
        if name == "opacities":
            plus[name] = plus[name].clamp(0.01, 0.99); minus[name] = minus[name].clamp(0.01, 0.99)
            d = (plus[name] - minus[name]) / (2 * eps)
        fd = (loss(plus) - loss(minus)) / (2 * eps)
        an = float((grads[name].double() * d.double()).sum())
        assert fd == pytest.approx(an, rel=5e-2, abs=1e-7), f"{name}: fd {fd:.4e} vs analytic {an:.4e}"


def test_c5_ranges_sorted_lists_and_linearity(R):
    """BASELINE config c5's Gaussian side: S(2 000 000) at 1080p. Ranges, per-tile depth order
    of the 5.4 M-entry lists (bit-level sortedness), and linearity of the backward."""
    p = _gpu_scene(2_000_000)
    vm, K = scenes.cameras([63])
    vm, K = vm.cuda(), K.cuda()
    rc, ra, meta = _render(R, p, vm, K)
    assert rc.shape == (1, H, W, 3) and torch.isfinite(rc).all() and float(rc.min()) >= 0.0
    assert float(ra.min()) >= 0.0 and float(ra.max()) <= 1.0 + 1e-6
    off = meta["isect_offsets"].reshape(-1).long()
    ids = meta["flatten_ids"].long()
    assert ids.numel() > 4_000_000 and int(ids.max()) < 2_000_000
    d = meta["depths"].reshape(-1)[ids]
    tile_of = torch.bucketize(torch.arange(ids.numel(), device=ids.device), off, right=True)
    same = tile_of[1:] == tile_of[:-1]
    assert (d[1:][same] >= d[:-1][same]).all()
    # ties in depth keep index order (the reference's stable sort)
    tie = same & (d[1:] == d[:-1])
    assert (ids[1:][tie] > ids[:-1][tie]).all()
    # every visible pair is listed once per tile it touches: list length == sum of tile counts
    r = meta["radii"][0].float()
    m2 = meta["means2d"][0].detach()
    vis = (meta["radii"][0] > 0).all(-1)
    x0 = ((m2[:, 0] - r[:, 0]) / 16).floor().clamp(0, 120); x1 = ((m2[:, 0] + r[:, 0]) / 16).ceil().clamp(0, 120)
    y0 = ((m2[:, 1] - r[:, 1]) / 16).floor().clamp(0, 68); y1 = ((m2[:, 1] + r[:, 1]) / 16).ceil().clamp(0, 68)
    assert int((((x1 - x0) * (y1 - y0))[vis]).sum()) == ids.numel()
    g = torch.Generator().manual_seed(6)
    w = torch.randn(1, H, W, 3, generator=g).cuda()
    g1, g2 = _grads(R, p, vm, K, w), _grads(R, p, vm, K, -0.5 * w)
    for k in g1:
        assert torch.isfinite(g1[k]).all()
        rel = float((g2[k] + 0.5 * g1[k]).norm() / g1[k].norm().clamp_min(1e-20))
        assert rel < 1e-4, f"{k}: {rel:.2e}"


@pytest.mark.parametrize("outliers", [False, True])
def test_c4_clustered_depths_lists_are_the_full_sort(R, outliers):
    """The c4 scene with its Gaussians on two thin shells (and, second case, 0.2 % outliers stretching the depth range
    30x): 850 of the 1020 buckets of the tile-list sort re-draw their depth bins from the tiles' own distribution
    (isect_bucket.hip 'Equalised bins'; profiles/r04_sort_paths.jsonl). The 2.6 M-entry lists must be exactly the
    (tile, depth, index) order: compared with a stable sort of the same pairs, element by element."""
    p = _gpu_scene(1_000_000)
    g = torch.Generator().manual_seed(11)
    z = torch.where(torch.rand(1_000_000, generator=g) < 0.5, -0.6, 0.6) + 0.01 * torch.randn(1_000_000, generator=g)
    if outliers:
        z[:2000] = -1.0 + 60.0 * torch.rand(2000, generator=g)
    p["means"] = p["means"].clone()
    p["means"][:, 2] = z.cuda()
    vm, K = scenes.cameras([7])
    with torch.no_grad():
        _, _, meta = _render(R, p, vm.cuda(), K.cuda())
    off = meta["isect_offsets"].reshape(-1).long()
    ids = meta["flatten_ids"].long()
    assert ids.numel() > 2_000_000
    tile_of = torch.bucketize(torch.arange(ids.numel(), device=ids.device), off, right=True) - 1
    depth_bits = meta["depths"].reshape(-1)[ids].view(torch.int32).long()          # positive floats: bit order = value order
    key = (tile_of << 32 | depth_bits)
    order = torch.sort(key, stable=True).indices                                   # stable: ties keep list order ...
    assert torch.equal(order, torch.arange(ids.numel(), device=ids.device))        # ... and the lists are already sorted
    same = key[1:] == key[:-1]
    assert (ids[1:][same] > ids[:-1][same]).all()                                   # equal (tile, depth): index order
    # nothing lost or duplicated: every visible Gaussian is listed once per tile of its rectangle
    r = meta["radii"][0].float()
    m2 = meta["means2d"][0]
    vis = (meta["radii"][0] > 0).all(-1)
    x0 = ((m2[:, 0] - r[:, 0]) / 16).floor().clamp(0, 120); x1 = ((m2[:, 0] + r[:, 0]) / 16).ceil().clamp(0, 120)
    y0 = ((m2[:, 1] - r[:, 1]) / 16).floor().clamp(0, 68); y1 = ((m2[:, 1] + r[:, 1]) / 16).ceil().clamp(0, 68)
    assert int((((x1 - x0) * (y1 - y0))[vis]).sum()) == ids.numel()
    per_gauss = torch.bincount(ids, minlength=1_000_000)
    assert torch.equal(per_gauss[vis], ((x1 - x0) * (y1 - y0))[vis].long())


def _c2_vs_oracle(width, height, cx, cy, n_gauss=100_000, sh_degree=3, timed_path=False, render_mode="RGB",
                  antialiased=False, img_atol=1e-4):
    """Oracle (CPU, ONE run) against the HIP path on camera 25's window of the 1080p frame.

    default path: `rendering.rasterization` with gsplat's rectangle-rule lists and activated inputs.
    timed_path=True ALSO runs the configuration bench.py times -- `runner.rasterize_splats`: raw
    log-scales / logit opacities with exp / sigmoid and their chain rule inside the projection
    kernels, `RasterConfig(tight_tiles=True)` lists, no `flatten_ids` -- on the same scene, and
    checks it against the same oracle run: images as before, gradients w.r.t. the RAW parameters
    against autograd through torch.exp / torch.sigmoid + the oracle (the pattern of
    test_gpu_rasterization.py::test_runner_rasterize_splats_fused_activations, at c2 / c5 scale)."""
    import time
    from oracle import rasterization_oracle as O
    from tests.test_gpu_rasterization import _check
    Rm = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    sc = scenes.make_scene(n_gauss, 0)
    vm, K = scenes.cameras([25])
    K = K.clone()
    K[0, 0, 2], K[0, 1, 2] = cx, cy
    names = ["means", "quats", "scales", "opacities", "sh0", "shN"]
    raw = dict(sc, scales=torch.log(sc["scales"]), opacities=torch.logit(sc["opacities"]))
    t0 = time.perf_counter()
    # oracle: raw leaves -> exp / sigmoid (autograd) -> oracle; the activated intermediates keep their
    # gradients too, so one backward serves both comparisons
    cpu_raw = {k: raw[k].clone().requires_grad_(True) for k in names}
    act = dict(cpu_raw, scales=torch.exp(cpu_raw["scales"]), opacities=torch.sigmoid(cpu_raw["opacities"]))
    act["scales"].retain_grad()
    act["opacities"].retain_grad()
    out_c = O.rasterization(act["means"], act["quats"], act["scales"], act["opacities"],
                            torch.cat([act["sh0"], act["shN"]], 1), vm, K, width, height, sh_degree=sh_degree,
                            render_mode=render_mode, rasterize_mode="antialiased" if antialiased else "classic")
    g = torch.Generator().manual_seed(11)
    w_c, w_a = torch.randn(out_c[0].shape, generator=g), torch.randn(out_c[1].shape, generator=g)
    ((out_c[0] * w_c).sum() + (out_c[1] * w_a).sum()).backward()
    t_oracle = time.perf_counter() - t0

    def hip(run):
        rc, ra, meta = run()
        meta["means2d"].retain_grad()
        ((rc * w_c.cuda()).sum() + (ra * w_a.cuda()).sum()).backward()
        torch.cuda.synchronize()
        return rc, ra, meta

    # ~40 fp32 terms per pixel: the mean error sits at 1e-6, the per-pixel bound stays 1e-4.
    # Threshold flips (a pair within rounding of alpha = 1/255 blended by one side only; sigma
    # is evaluated in a different but equivalent order here) touch ~12 of the 100 k Gaussians
    # at the full frame, i.e. 1.2e-4 of the gradient elements: allowed fraction 2e-4.
    # (a) default lists, activated inputs
    gpu = {k: act[k].detach().clone().cuda().requires_grad_(True) for k in names}
    out_g = hip(lambda: Rm.rasterization(gpu["means"], gpu["quats"], gpu["scales"], gpu["opacities"],
                                         (gpu["sh0"], gpu["shN"]), vm.cuda(), K.cuda(), width, height,
                                         sh_degree=sh_degree, packed=False, render_mode=render_mode,
                                         rasterize_mode="antialiased" if antialiased else "classic"))
    _check(act, gpu, out_c, out_g, mean_frac=5e-2, flip_frac=2e-4, img_atol=img_atol)
    if timed_path:
        # (b) the timed configuration: raw parameters, tight lists, through runner.rasterize_splats
        gpu_raw = torch.nn.ParameterDict({k: torch.nn.Parameter(raw[k].clone().cuda()) for k in names})
        cfg = runner.RasterConfig(sh_degree=3, tight_tiles=True, antialiased=antialiased)
        out_r = hip(lambda: runner.rasterize_splats(gpu_raw, torch.linalg.inv(vm).cuda(), K.cuda(), width, height,
                                                    cfg, sh_degree=sh_degree, render_mode=render_mode))
        _check(cpu_raw, gpu_raw, out_c, out_r, mean_frac=5e-2, flip_frac=2e-4, img_atol=img_atol)
    print(f"{n_gauss} Gaussians, view {width}x{height}, SH degree {sh_degree}: oracle fwd+bwd {t_oracle:.1f} s, "
          f"total {time.perf_counter() - t0:.1f} s")


def test_c2_window_vs_oracle():
    """BASELINE config c2 (100 k Gaussians, camera 25 of the 1080p rig) against the CPU oracle,
    forward and backward, on a 512x512 window of the frame (principal point shifted, same
    focal length): footprints and list lengths are those of the full frame, the oracle's
    tile loop is 8x shorter. Both the gsplat-rule path and the configuration bench.py times."""
    _c2_vs_oracle(512, 512, 960.0 - 704.0, 540.0 - 284.0, timed_path=True)


@pytest.mark.parametrize("sh_degree", [0, 1, 2])
def test_c2_window_vs_oracle_lower_sh_degrees(sh_degree):
    """The same window at the SH degrees steps 0-2999 of every training run use (runner.py:464): degrees 1
    and 2 read shN band-wise (a different code path from degree 3's LDS-DMA slab), degree 0 reads sh0 only;
    the unused bands must receive exactly zero gradient. Both paths, as above."""
    _c2_vs_oracle(512, 512, 960.0 - 704.0, 540.0 - 284.0, sh_degree=sh_degree, timed_path=True)


def test_c2_window_depth_channel_antialiased_vs_oracle():
    """The other two modes the reference's callers use, at c2 scale: render_mode "RGB+ED" (the depth-loss branch
    runner.py:476-482 and the nerfbaselines render, method.py:754-764: expected depth as a fourth channel, divided
    by alpha) with rasterize_mode "antialiased" (config.py:149: the compensation factor multiplies the opacity, the
    sigmoid chain rule then runs outside the projection kernel). Both paths; the depth channel's magnitude is ~4,
    hence the absolute tolerance 5e-4 (as in the tiny-scene test of the same mode)."""
    _c2_vs_oracle(512, 512, 960.0 - 704.0, 540.0 - 284.0, timed_path=True, render_mode="RGB+ED", antialiased=True,
                  img_atol=5e-4)


def test_c5_window_vs_oracle():
    """BASELINE config c5's Gaussian side (2 M Gaussians) against the CPU oracle, forward and
    backward, on a 384x384 window of the 1080p frame (oracle time 1-3 min depending on the box's
    host share; 512x512 took up to 5.6 min on a contended box). Both the gsplat-rule path and the
    configuration bench.py times (raw parameters + tight lists through runner.rasterize_splats)."""
    _c2_vs_oracle(384, 384, 960.0 - 704.0, 540.0 - 284.0, n_gauss=2_000_000, timed_path=True)


def test_c2_one_view_full_size_vs_oracle():
    """The same at the real 1920x1080: one full c2 view against the CPU oracle (1.5-5 min of oracle
    time on the GPU box's 16 host threads, depending on the box's host share)."""
    _c2_vs_oracle(W, H, 960.0, 540.0, timed_path=True)


def test_c4_fused_adam_backward_equals_backward_then_adam_step():
    """`gsr_project_bwd_adam` (optimizer in backward: what bench.py's headline step runs) against
    `gsr_project_bwd` + `gsr_adam_step` at c4's full size (1 M Gaussians, 1080p, SH degree 3), two steps.

    The compositing backward's float atomics are order-nondeterministic, so two complete steps never see
    bit-identical gradient rows. Here ONE real step produces the [N,16] rows (L1 loss against the seeded
    noise target, tight lists, raw parameters -- the bench's configuration); both variants then consume
    that SAME buffer through the projection backward's autograd node, so every difference left is the
    fused kernel itself: parameters and both Adam moments of all 59 M elements must agree to rounding."""
    Rm = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
    losses = importlib.import_module("3dgs_monocular_depth_init_amd.losses")
    N = 1_000_000
    sc = scenes.make_scene(N, 0)
    vm, K = scenes.cameras([3])
    c2w, K = torch.linalg.inv(vm).cuda(), K.cuda()
    target = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(2)).cuda()

    def fresh():
        splats, opts = runner.create_splats_with_optimizers(
            sc["means"], torch.rand(N, 3, generator=torch.Generator().manual_seed(0)), torch.log(sc["scales"]),
            quats=sc["quats"], opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
        return splats, D.fuse_optimizers(splats, opts)

    # one real step's gradient rows: the buffer the compositing backward hands to the projection backward
    seen = []
    orig = Rm._rows_from_grads

    def spy(*a, **k):
        out = orig(*a, **k)
        seen.append(out)
        return out

    Rm._rows_from_grads = spy
    request_restore = lambda: setattr(Rm, "_rows_from_grads", orig)
    try:
        splats, _ = fresh()
        rc, _, info = runner.rasterize_splats(splats, c2w, K, W, H, runner.RasterConfig(), sh_degree=3)
        losses.l1_loss(rc, target).backward()
        rows, fast = seen.pop()
        assert fast and rows.shape == (N, 16)
        rows = rows.clone()
    except BaseException:
        request_restore()
        raise
    vis = (info["radii"][0] > 0).all(-1)
    assert int(vis.sum()) > 800_000 and float(rows[:, :9].abs().max()) > 0
    del rc, info, splats

    viewmats, campos = Rm.inverse4x4(c2w, translation_of="input")
    cfg = (W, H, 0.3, 0.01, 1e10, 0.0, False, 3, 3, -1, Rm.ACT_EXP_SCALES | Rm.ACT_SIGMOID_OPAC, 0, 0)

    def run(fuse):
        splats, fused = fresh()
        try:
            if fuse:
                fused.fuse_into_backward(True)
            for _ in range(2):
                out = Rm._ProjectSH.apply(splats["means"], splats["quats"], splats["scales"], splats["opacities"],
                                          splats["sh0"], splats["shN"], viewmats, K, campos, cfg)
                _, means2d, _, conics, _, colors, opac_act, _, _ = out
                r = rows.clone()     # (the views must share ONE base: the fast, copy-free hand-over)
                torch.autograd.backward(
                    [means2d, conics, colors, opac_act],
                    [r[:, 0:2].view(1, N, 2), r[:, 2:5].view(1, N, 3), r[:, 6:9].view(1, N, 3), r[:, 5].view(1, N).reshape(N)])
                if fuse:
                    assert all(p.grad is None for p in splats.values())
                fused.step()
                fused.zero_grad(set_to_none=True)
        finally:
            Rm.set_backward_optimizer(None)
        torch.cuda.synchronize()
        st = {n: fused[n].state[splats[n]] for n in splats}
        return ({n: p.detach() for n, p in splats.items()}, {n: s["exp_avg"] for n, s in st.items()},
                {n: s["exp_avg_sq"] for n, s in st.items()}, {n: float(s["step"]) for n, s in st.items()})

    try:
        p0, m0, v0, s0 = run(False)
        p1, m1, v1, s1 = run(True)
    finally:
        request_restore()
    assert len(seen) == 4 and all(f for _, f in seen)     # every backward took the copy-free hand-over
    from tests import parity_log
    for n in p0:
        assert s0[n] == s1[n] == 2.0, n
        for what, a, b in (("param", p0[n], p1[n]), ("exp_avg", m0[n], m1[n]), ("exp_avg_sq", v0[n], v1[n])):
            scale = float(a.abs().max())
            err = (a - b).abs()
            n_off = int((err > 1e-6 * scale + 1e-5 * a.abs()).sum())
            parity_log.record("fused_adam_c4", tensor=n, what=what, elements=a.numel(), max_abs_err=float(err.max()),
                              ref_max=scale, n_off=n_off)
            # same gradient arithmetic, same adam_one: differences are fma contraction at most
            assert n_off <= a.numel() // 100_000, (n, what, n_off, float(err.max()), scale)
            assert float(err.max()) <= 1e-4 * scale, (n, what, float(err.max()), scale)
    for n in ("means", "quats", "shN"):      # and the parameters did move
        assert float((p1[n] - sc[n].cuda()).abs().max()) > 0, n


def test_c4_l1_loss_in_the_compositing_forward_at_full_size():
    """gsr_rasterize_fwd_l1 at c4's size (1 M Gaussians, 1080p: 8 160 tiles, the last row of tiles half outside the
    image): the loss and d loss / d render it writes against the render of gsr_rasterize_fwd + the L1 loss launches on
    the same scene -- the loss to 1e-7, the gradient image exactly (sign(render - target) / n wherever |render - target|
    is not within rounding of zero), and the alphas bit for bit."""
    Rm = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    losses = importlib.import_module("3dgs_monocular_depth_init_amd.losses")
    N = 1_000_000
    sc = scenes.make_scene(N, 0)
    vm, K = scenes.cameras([3])
    c2w, K = torch.linalg.inv(vm).contiguous().cuda(), K.cuda()
    target = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(2)).cuda()
    splats, _ = runner.create_splats_with_optimizers(
        sc["means"], torch.rand(N, 3, generator=torch.Generator().manual_seed(0)), torch.log(sc["scales"]),
        quats=sc["quats"], opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
    grabbed = {}
    orig_bwd = Rm._Rasterize.backward

    def spy(ctx, *grads):
        grabbed["l1_grad"] = None if ctx.l1_grad is None else ctx.l1_grad.clone()
        return orig_bwd(ctx, *grads)

    with torch.no_grad():
        rc, ra, _ = runner.rasterize_splats(splats, c2w, K, W, H, runner.RasterConfig(), sh_degree=3)
    ref_loss = float(losses.l1_loss(rc, target))
    Rm._Rasterize.backward = staticmethod(spy)
    try:
        none, ra2, info = runner.rasterize_splats(splats, c2w, K, W, H, runner.RasterConfig(), sh_degree=3, _l1_target=target)
        assert none is None
        info["l1_loss"].backward()
    finally:
        Rm._Rasterize.backward = staticmethod(orig_bwd)
    assert abs(float(info["l1_loss"]) - ref_loss) < 1e-7
    assert torch.equal(ra2, ra)
    g = grabbed["l1_grad"]
    n = float(rc.numel())
    d = rc - target
    sure = d.abs() > 1e-6
    assert torch.equal(g[sure], (torch.sign(d) / n)[sure])
    assert float(g.abs().max()) <= 1.0 / n * (1 + 1e-6)
    assert all(torch.isfinite(p.grad).all() for p in splats.values())
