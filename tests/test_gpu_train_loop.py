"""runner.train: the reference's training LOOP (runner.py:367-709) around the hot path -- ExponentialLR on the
means (runner.py:381-386, 687-689), SH-degree schedule from step 0 (runner.py:464), L1 + 0.2 (1 - SSIM)
(runner.py:506-510), DefaultStrategy with the reference's default schedule scaled by Config.adjust_steps
(config.py:204-221), checkpoints before the update at save_steps (runner.py:592-637), evaluation at eval_steps.
A 3 000-step run (steps_scaler 0.1) on a small scene; the full-length c5 rehearsal is tools/c5_rehearsal.py."""
import importlib
import math

import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu
P = "3dgs_monocular_depth_init_amd."


def _dataset(W, H, n_gt=2500, n_views=14):
    runner = importlib.import_module(P + "runner")
    gt = scenes.make_scene(n_gt, 7, box=(1.0, 0.7, 0.4), scale_mean=0.04)
    vms, Ks = scenes.cameras(range(0, 98, 98 // n_views), width=W, height=H, f=130.0, dist=2.5)
    c2ws = torch.linalg.inv(vms)
    gt_splats, _ = runner.create_splats_with_optimizers(
        gt["means"], torch.rand(n_gt, 3), torch.log(gt["scales"]), quats=gt["quats"],
        opacities_logit=torch.logit(gt["opacities"]), shN=gt["shN"])
    with torch.no_grad():
        gt_splats["sh0"].copy_(gt["sh0"].cuda())
        frames = []
        for i in range(len(vms)):
            img = runner.rasterize_splats(gt_splats, c2ws[i:i + 1].cuda(), Ks[i:i + 1].cuda(), W, H, sh_degree=3)[0]
            # the reference's dataset hands over 0-255 images (datasets/colmap.py:381-412; runner.py:447 divides)
            frames.append({"camtoworld": c2ws[i].cuda(), "K": Ks[i].cuda(), "image": (img[0].clamp(0, 1) * 255.0),
                           "image_id": i})
    return gt, frames


def test_train_loop_reference_schedule(tmp_path):
    runner = importlib.import_module(P + "runner")
    cfgm = importlib.import_module(P + "config")
    S = importlib.import_module(P + "strategy")
    gs_io = importlib.import_module(P + "io")
    knn = importlib.import_module(P + "knn")
    torch.manual_seed(0)
    W, H = 160, 112
    gt, frames = _dataset(W, H)
    train_frames, val_frames = frames[:-2], frames[-2:]
    cfg = cfgm.Config()
    assert isinstance(cfg.strategy, S.DefaultStrategy)
    assert (cfg.max_steps, cfg.sh_degree_interval, cfg.strategy.refine_start_iter, cfg.strategy.refine_stop_iter,
            cfg.strategy.refine_every, cfg.strategy.reset_every) == (30_000, 1000, 500, 15_000, 100, 3000)
    cfg.adjust_steps(0.1)                                                  # trainer.py:43-47
    assert (cfg.max_steps, cfg.sh_degree_interval, cfg.strategy.refine_start_iter, cfg.strategy.refine_stop_iter,
            cfg.strategy.refine_every, cfg.strategy.reset_every) == (3000, 100, 50, 1500, 10, 300)
    assert cfg.save_steps == [700, 3000] and cfg.eval_steps == [700, 3000]
    cfg.strategy.grow_grad2d = 1e-4          # (a 160x112 image: fewer pixels per Gaussian than the default presumes)
    n0 = 1200
    pts = gt["means"][:n0] + 0.02 * torch.randn(n0, 3)
    splats, opts = runner.create_splats_with_optimizers(pts, torch.rand(n0, 3), knn.initial_log_scales(pts.cuda()).cpu(),
                                                        init_opacity=cfg.init_opa)
    lr0 = opts["means"].param_groups[0]["lr"]
    seen = []
    stats = runner.train(splats, opts, train_frames, cfg, valset=val_frames, result_dir=tmp_path,
                         progress=lambda step, rec: seen.append(step), progress_every=100)
    # SH degree 0 -> 1 -> 2 -> 3 at multiples of the scaled interval, from step 0 (runner.py:464)
    assert stats["sh_degree_switches"] == [(0, 0), (100, 1), (200, 2), (300, 3)]
    # the means' learning rate ends at 0.01 of its initial value (runner.py:381-386), the others never move
    assert stats["final_lr_means"] == pytest.approx(0.01 * lr0, rel=1e-6)
    lrs = [r["lr_means"] for r in stats["intervals"]]
    assert all(b < a for a, b in zip(lrs, lrs[1:]))
    assert opts["scales"].param_groups[0]["lr"] == pytest.approx(5e-3)
    # strategy: (1500 - 50) / 10 refine steps (minus those that coincide with a reset), 4 resets before the stop
    assert stats["refine_steps"] + stats["reset_steps"] == len(
        [s for s in range(3000) if cfg.strategy.mutates_params(s)])
    assert stats["reset_steps"] == 4 and stats["refine_steps"] > 100
    counts = [r["num_GS"] for r in stats["intervals"]]
    assert len(set(counts)) > 1, "densification never changed the Gaussian count"
    losses = [r["loss"] for r in stats["intervals"]]
    assert all(math.isfinite(x) for x in losses) and losses[-1] < 0.5 * losses[0], losses
    assert seen == list(range(99, 3000, 100))
    # checkpoints at step 699 and 2999, in the reference's format, written BEFORE that step's update
    names = sorted(p.name for p in (tmp_path / "ckpts").iterdir())
    assert names == ["ckpt_2999_rank0.pt", "ckpt_699_rank0.pt", "splats_2999.ply", "splats_699.ply"]
    ck = gs_io.load_checkpoints([tmp_path / "ckpts" / "ckpt_2999_rank0.pt"])
    assert ck["step"] == 2999 and set(ck["splats"]) == {"means", "scales", "quats", "opacities", "sh0", "shN"}
    assert ck["splats"]["means"].shape == splats["means"].shape
    moved = float((ck["splats"]["means"].cuda() - splats["means"].detach()).abs().max())
    assert 0.0 < moved < 1e-2             # the last Adam step came after the save, and is small (lr = 1.6e-6)
    ply = gs_io.load_ply(tmp_path / "ckpts" / "splats_2999.ply")
    assert torch.equal(ply["means"], ck["splats"]["means"].cpu())
    # evaluation on the two held-out views at steps 699 and 2999
    assert [e["step"] for e in stats["evals"]] == [699, 2999]
    assert stats["evals"][-1]["psnr"] > stats["evals"][0]["psnr"] - 0.5 and stats["evals"][-1]["psnr"] > 20.0, stats["evals"]
    assert 0.0 < stats["evals"][-1]["ssim"] <= 1.0
    # every parameter still has matching optimizer state
    for k, p in splats.items():
        stt = opts[k].state[p]
        assert stt["exp_avg"].shape == p.shape and float(stt["step"]) > 0


def test_train_step_random_background_and_depth_loss():
    """The optional loss terms of the loop body: random background (runner.py:493-495) and the disparity-space
    depth loss through grid_sample on the RGB+ED render (runner.py:511-529). Gradients reach every parameter
    and the depth term changes them."""
    runner = importlib.import_module(P + "runner")
    W, H = 160, 112
    gt, frames = _dataset(W, H, n_gt=800, n_views=2)
    f = frames[0]
    c2w, K, pixels = f["camtoworld"][None], f["K"][None], f["image"][None] / 255.0
    g = torch.Generator().manual_seed(5)
    # (sample points inside the rendered footprint: where nothing is rendered the expected depth is 0 and the
    # reference's torch.where(d > 0, 1 / d, 0) back-propagates 0 * inf = NaN -- its behaviour, kept)
    pts2d = torch.stack([50.0 + torch.rand(1, 64, generator=g) * 60.0, 36.0 + torch.rand(1, 64, generator=g) * 40.0], -1).cuda()
    d_gt = (2.0 + torch.rand(1, 64, generator=g)).cuda()

    def grads(**kw):
        splats, _ = runner.create_splats_with_optimizers(
            gt["means"], torch.rand(800, 3, generator=torch.Generator().manual_seed(1)), torch.log(gt["scales"]),
            quats=gt["quats"], opacities_logit=torch.logit(gt["opacities"]), shN=gt["shN"])
        torch.manual_seed(3)
        loss, info = runner.train_step(splats, None, c2w, K, pixels, step=5000, ssim_lambda=0.2, **kw)
        return float(loss), {k: p.grad.clone() for k, p in splats.items()}

    l0, g0 = grads()
    l1, g1 = grads(random_background=True)
    l2, g2 = grads(depth_points=pts2d, depth_gt=d_gt, depth_lambda=1e-2, scene_scale=1.0)
    for gg in (g0, g1, g2):
        assert all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for v in gg.values())
    assert l1 != l0 and float((g1["opacities"] - g0["opacities"]).abs().max()) > 0
    assert l2 > l0 and float((g2["means"] - g0["means"]).abs().max()) > 0
