#!/usr/bin/env python3
"""Full-length rehearsal of BASELINE config c5 on ONE MI355X: 2 M-Gaussian scene, 100 views at 1080p, monocular-depth
initialisation, the reference's 30 000-iteration schedule (runner.train = runner.py:367-709).

  1. A hidden ground-truth scene S(2 M, seed 3) is rendered into 100 training images + 4 held-out views, with
     its expected-depth maps (render_mode "RGB+ED").
  2. Initialisation through the depth-init pipeline (monocular_depth_init.pts_and_rgb_from_frames: B1-B9 + F3):
     the "predicted" depth of every image is the ground-truth depth render in a different scale and shift with
     1 % noise (Metric3D-L with the deterministic hash weights predicts nothing alignable: the network is timed by
     tools/bench_depthnet.py and pinned by tests/test_gpu_depthnet.py instead), "SfM" points are 4 000 points of each
     view's visible surface (ground-truth depth unprojected, 2 mm of noise); RANSAC alignment, stride-10 subsampling, patch mask,
     unprojection, kNN scales -- the reference's defaults.
  3. runner.train with config.Config() defaults (steps_scaler 1: 30 000 steps, SH degree every 1000, DefaultStrategy
     refine 500 -> 15 000 every 100, reset every 3000, L1 + 0.2 (1 - SSIM), ExponentialLR on the means, checkpoint +
     PLY at steps 6 999 / 29 999, evaluation on the held-out views).

Prints one JSON record (-> profiles/r04_c5_rehearsal.json): wall time, ms/step per 1000-step interval, Gaussian
count over time, losses, held-out PSNR / SSIM, peak memory.   python tools/c5_rehearsal.py [--steps-scaler 1.0]"""
import argparse
import importlib
import json
import math
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps-scaler", type=float, default=1.0)
ap.add_argument("--gaussians", type=int, default=2_000_000)
ap.add_argument("--views", type=int, default=100)
ap.add_argument("--result-dir", default="/tmp/c5_rehearsal")      # (checkpoints of a few GB: not under gpurun_out)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--isect-stats", action="store_true", help="print the tile-list / bucket length distribution of the seed cloud and of the trained scene")
args = ap.parse_args()
P = "3dgs_monocular_depth_init_amd."
runner = importlib.import_module(P + "runner")
cfgm = importlib.import_module(P + "config")
mdi = importlib.import_module(P + "monocular_depth_init")
dpi = importlib.import_module(P + "depth_prediction.predictors.depth_predictor_interface")
knn = importlib.import_module(P + "knn")
torch.manual_seed(42)                                   # runner.py:147 (42 + local_rank)
W, H, N = args.width, args.height, args.gaussians
t_all = time.time()

# ---- 1. hidden ground truth -> images + depth ------------------------------------------------------------
gt = scenes.make_scene(N, 3)
f = 1200.0 * W / 1920
ids = [i * 100.0 / args.views for i in range(args.views)] + [12.5, 37.5, 62.5, 87.5]
vms, Ks = scenes.cameras(ids, width=W, height=H, f=f)
c2ws = torch.linalg.inv(vms).cuda()
Ks = Ks.cuda()
gt_splats, _ = runner.create_splats_with_optimizers(
    gt["means"], torch.rand(N, 3), torch.log(gt["scales"]), quats=gt["quats"],
    opacities_logit=torch.logit(gt["opacities"]), shN=gt["shN"])
frames, depths, alphas = [], [], []
with torch.no_grad():
    gt_splats["sh0"].copy_(gt["sh0"].cuda())
    for i in range(len(ids)):
        r, a, _ = runner.rasterize_splats(gt_splats, c2ws[i:i + 1], Ks[i:i + 1], W, H, sh_degree=3, render_mode="RGB+ED")
        frames.append({"camtoworld": c2ws[i], "K": Ks[i], "image": (r[0, ..., :3].clamp(0, 1) * 255.0).contiguous(), "image_id": i})
        depths.append(r[0, ..., 3].contiguous())
        alphas.append(a[0, ..., 0].contiguous())
del gt_splats
torch.cuda.synchronize()
train_frames, val_frames = frames[:args.views], frames[args.views:]
t_render = time.time() - t_all

# ---- 2. depth-init pipeline ---------------------------------------------------------------------------------
g = torch.Generator().manual_seed(5)
gd = torch.Generator(device="cuda").manual_seed(5)
SFM_PER_VIEW = 4000


def surface_points(i, n):
    """n points of view i's visible surface: random pixels with alpha > 0.5, unprojected with the ground-truth depth
    render (+ 2 mm of noise) -- what SfM would triangulate there. (Rounds 2-4 used centres of the hidden Gaussians:
    in this volumetric scene they do not lie on the expected-depth surface the depth maps show, 12 % of them were
    RANSAC inliers, every image ran the full 2 500 iterations and the recovered scale / shift were arbitrary.)"""
    ok = torch.nonzero(alphas[i] > 0.5)
    pick = ok[torch.randint(len(ok), (n,), device="cuda", generator=gd)]
    v, u = pick[:, 0].float(), pick[:, 1].float()
    z = depths[i][pick[:, 0], pick[:, 1]]
    cam = torch.stack([(u + 0.5 - W / 2) / f * z, (v + 0.5 - H / 2) / f * z, z], -1)
    world = (cam - vms[i][:3, 3].cuda()) @ vms[i][:3, :3].cuda()          # R^T (x - t)
    return (world + 0.002 * torch.randn(world.shape, device="cuda", generator=gd),
            train_frames[i]["image"][pick[:, 0], pick[:, 1]])


view_pts, view_rgb = zip(*[surface_points(i, SFM_PER_VIEW) for i in range(args.views)])
sfm_pts = torch.cat([p[:300] for p in view_pts])                            # the scene's "SfM cloud" (include_sfm_points)
sfm_rgb = torch.cat([c[:300] for c in view_rgb])


class GroundTruthDepth(dpi.DepthPredictor):
    """Stands in for Metric3d.predict_depth: the scene's depth render in another scale / shift, 1 % noise."""

    def __init__(self, config=None, device="cuda"):
        self.k = 0

    @property
    def name(self):
        return "gt_depth_render"

    def predict_depth(self, img, intrinsics):
        d, a = depths[self.k], alphas[self.k]
        self.k += 1
        noise = 1.0 + 0.01 * torch.randn(d.shape, device=d.device)
        return dpi.PredictedDepth(depth=((d - 0.4) / 1.7 * noise).float(), mask=a > 0.5)


def frames_with_sfm():
    for i, fr in enumerate(train_frames):
        yield mdi.Frame(image=fr["image"], image_name=f"view{i:03d}.png", camtoworld=fr["camtoworld"].cpu(), K=fr["K"].cpu(),
                        sfm_points=view_pts[i])


cfg = cfgm.Config()
cfg.mdi.cache_dir = None
cam_centres = c2ws[:args.views, :3, 3]
scene_scale = float((cam_centres - cam_centres.mean(0)).norm(dim=1).max()) * 1.1        # datasets/colmap.py scene_scale * 1.1 (runner.py:180)
t0 = time.time()
pts, rgbs, _ = mdi.pts_and_rgb_from_frames(cfg, frames_with_sfm(), GroundTruthDepth(), "cuda:0", sfm_points=sfm_pts,
                                          sfm_points_rgb=sfm_rgb, scene_scale=scene_scale)
torch.cuda.synchronize()
t_init = time.time() - t0
t0 = time.time()
log_scales = knn.initial_log_scales(pts, cfg.init_scale)                               # runner.py:88-91
torch.cuda.synchronize()
t_knn = time.time() - t0
# how good is the seed cloud: distance of each seed to the nearest hidden Gaussian centre (sampled)
with torch.no_grad():
    samp = pts[torch.randperm(len(pts), generator=g)[:2000].cuda()]
    ref = gt["means"].cuda()
    d_seed = torch.stack([(ref - s).norm(dim=1).min() for s in samp[:500]])
splats, opts = runner.create_splats_with_optimizers(pts, rgbs, log_scales, init_opacity=cfg.init_opa, scene_scale=scene_scale,
                                                    sh_degree=cfg.sh_degree, batch_size=cfg.batch_size)
n_init = len(pts)
del pts, rgbs, log_scales, depths, alphas

def isect_stats(tag):
    """Lengths of the tile lists and of the 8-tile buckets the sort kernel works on (isect_bucket.hip), five views."""
    out = []
    with torch.no_grad():
        for i in range(0, args.views, max(args.views // 5, 1)):
            _, _, info = runner.rasterize_splats(splats, c2ws[i:i + 1], Ks[i:i + 1], W, H, sh_degree=0)
            off = info["isect_offsets"].reshape(-1).long()
            total = int(info["flatten_ids"].shape[0])
            lens = torch.diff(torch.cat([off, off.new_tensor([total])])).view(info["tile_height"], info["tile_width"])
            tw = lens.shape[1]
            pad = (-tw) % 8
            b = torch.nn.functional.pad(lens, (0, pad)).view(lens.shape[0], -1, 8).sum(-1).flatten().float()
            q = lambda x, f: int(torch.quantile(x.flatten().float(), f))
            out.append({"view": i, "pairs": total, "tile_max": int(lens.max()), "tile_p99": q(lens, 0.99), "bucket_mean": int(b.mean()),
                        "bucket_p50": q(b, 0.5), "bucket_p90": q(b, 0.9), "bucket_max": int(b.max()),
                        "buckets_over_8192": int((b > 8192).sum()), "buckets": int(b.numel()),
                        "pairs_in_buckets_over_8192": round(float(b[b > 8192].sum() / b.sum()), 3)})
    print(json.dumps({"isect_stats": tag, "num_GS": len(splats["means"]), "views": out}), flush=True)


if args.isect_stats:
    isect_stats("seed cloud")

# ---- 3. the loop ----------------------------------------------------------------------------------------------
cfg.adjust_steps(args.steps_scaler)
every = max(int(1000 * args.steps_scaler), 1)


def progress(step, rec):
    print(json.dumps({"progress": rec, "minutes": round((time.time() - t_all) / 60, 2)}), flush=True)


torch.cuda.reset_peak_memory_stats()
stats = runner.train(splats, opts, train_frames, cfg, valset=val_frames, scene_scale=scene_scale, result_dir=args.result_dir,
                     progress=progress, progress_every=every)
if args.isect_stats:
    isect_stats("after %d steps" % stats["steps"])
ck = sorted(Path(args.result_dir, "ckpts").glob("*"))
losses = [r["loss"] for r in stats["intervals"]]
rec = {
    "what": "c5 rehearsal on one MI355X: hidden S(%d) -> %d views %dx%d, depth-init pipeline -> runner.train, reference schedule x %g"
            % (N, args.views, W, H, args.steps_scaler),
    "steps": stats["steps"], "train_seconds": round(stats["seconds"], 1), "wall_seconds": round(time.time() - t_all, 1),
    "render_targets_seconds": round(t_render, 1),
    "init": {"seed_points": n_init, "pipeline_seconds": round(t_init, 2), "knn_scales_seconds": round(t_knn, 2),
             "median_distance_to_nearest_hidden_centre": float(d_seed.median()), "scene_scale": scene_scale},
    "ms_per_step_by_interval": [round(r["ms_per_step"], 3) for r in stats["intervals"]],
    "num_GS_by_interval": [r["num_GS"] for r in stats["intervals"]],
    "loss_by_interval": [round(x, 5) for x in losses],
    "sh_degree_switches": stats["sh_degree_switches"], "refine_steps": stats["refine_steps"], "reset_steps": stats["reset_steps"],
    "lr_means_first_last": [stats["intervals"][0]["lr_means"], stats["final_lr_means"]],
    "evals": stats["evals"], "final_num_GS": stats["num_GS"], "peak_mem_gib": round(stats["peak_mem_gib"], 2),
    "checkpoints": [{"file": p.name, "mib": round(p.stat().st_size / 2 ** 20, 1)} for p in ck],
    "ok": all(math.isfinite(x) for x in losses) and losses[-1] < losses[0],
}
print(json.dumps(rec))
try:        # a counting build of the tile-list sort (tools/sort_paths.py): which path did the buckets of the whole run take?
    import ctypes
    lib = importlib.import_module(P + "_lib").load()
    out = (ctypes.c_ulonglong * 8)()
    lib.gsr_debug_sort_paths.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    lib.gsr_debug_sort_paths(out, 0)
    print(json.dumps({"sort_paths": dict(zip(["buckets", "equalised_parked", "equalised_streamed", "several_groups", "networks",
                                              "global_network"], [int(x) for x in out[:6]]))}))
except AttributeError:
    pass
sys.exit(0 if rec["ok"] else 1)
