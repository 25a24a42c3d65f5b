"""Stage-wise error report + kernel timings (development aid; run on the GPU box)."""
import importlib
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import rasterization_oracle as O  # noqa: E402
from tests import scenes  # noqa: E402

R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def report(tag, sc, vm, K, W, H, **kw):
    names = ["means", "quats", "scales", "opacities", "sh0", "shN"]
    cpu = {k: sc[k].clone().requires_grad_(True) for k in names}
    gpu = {k: sc[k].clone().cuda().requires_grad_(True) for k in names}
    rc_c, ra_c, mc = O.rasterization(cpu["means"], cpu["quats"], cpu["scales"], cpu["opacities"],
                                     torch.cat([cpu["sh0"], cpu["shN"]], 1), vm, K, W, H,
                                     sh_degree=3, **kw)
    rc_g, ra_g, mg = R.rasterization(gpu["means"], gpu["quats"], gpu["scales"], gpu["opacities"],
                                     (gpu["sh0"], gpu["shN"]), vm.cuda(), K.cuda(), W, H,
                                     sh_degree=3, packed=False, **kw)
    g = torch.Generator().manual_seed(1)
    w = torch.randn(rc_c.shape, generator=g)
    wa = torch.randn(ra_c.shape, generator=g)
    ((rc_c * w).sum() + (ra_c * wa).sum()).backward()
    ((rc_g * w.cuda()).sum() + (ra_g * wa.cuda()).sum()).backward()
    torch.cuda.synchronize()
    vis_c = (mc["radii"] > 0).all(-1)
    vis_g = (mg["radii"].cpu() > 0).all(-1)
    print(f"[{tag}] vis cpu {int(vis_c.sum())} gpu {int(vis_g.sum())} differ {int((vis_c != vis_g).sum())}"
          f" isects cpu {mc['flatten_ids'].numel()} gpu {mg['flatten_ids'].numel()}")
    both = vis_c & vis_g
    print(f"[{tag}] radii equal {torch.equal(mc['radii'][both], mg['radii'].cpu()[both])}"
          f" means2d rel {rel(mg['means2d'].detach().cpu()[both], mc['means2d'].detach()[both]):.2e}"
          f" conics rel {rel(mg['conics'].detach().cpu()[both], mc['conics'].detach()[both]):.2e}")
    if mc["flatten_ids"].numel() == mg["flatten_ids"].numel():
        print(f"[{tag}] flatten_ids equal {torch.equal(mc['flatten_ids'], mg['flatten_ids'].cpu())}"
              f" offsets equal {torch.equal(mc['isect_offsets'], mg['isect_offsets'].cpu())}")
    ec = (rc_g.detach().cpu() - rc_c.detach()).abs()
    ea = (ra_g.detach().cpu() - ra_c.detach()).abs()
    print(f"[{tag}] image max err {ec.max():.3e} mean {ec.mean():.3e} (#>1e-4: {int((ec > 1e-4).sum())})"
          f" alpha max err {ea.max():.3e}")
    for k in names:
        if cpu[k].grad is not None and gpu[k].grad is not None:
            print(f"[{tag}] grad {k:10s} rel {rel(gpu[k].grad.cpu(), cpu[k].grad):.3e}")
        else:
            print(f"[{tag}] grad {k:10s} cpu {cpu[k].grad is not None} gpu {gpu[k].grad is not None}")


def timing(N=1_000_000, W=1920, H=1080, iters=10):
    sc = scenes.make_scene(N, 0)
    vm, K = scenes.cameras([0])
    names = ["means", "quats", "scales", "opacities", "sh0", "shN"]
    gpu = {k: sc[k].clone().cuda().requires_grad_(True) for k in names}
    vm, K = vm.cuda(), K.cuda()
    target = torch.rand(1, H, W, 3, device="cuda")

    def step():
        rc, ra, meta = R.rasterization(gpu["means"], gpu["quats"], gpu["scales"], gpu["opacities"],
                                       (gpu["sh0"], gpu["shN"]), vm, K, W, H, sh_degree=3, packed=False)
        loss = (rc - target).abs().mean()
        loss.backward()
        return meta

    for _ in range(3):
        meta = step()
    torch.cuda.synchronize()
    print(f"[timing] N={N} visible {int((meta['radii'] > 0).all(-1).sum())} isects {meta['flatten_ids'].numel()}")
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"[timing] fwd+bwd {dt * 1e3:.3f} ms/iter -> {1 / dt:.1f} it/s")
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))


if __name__ == "__main__":
    sc = scenes.make_scene(600, 5, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    vm = torch.eye(4)[None].clone()
    vm[0, 2, 3] = 2.5
    K = torch.tensor([[[80.0, 0, 35], [0, 80.0, 25], [0, 0, 1]]])
    report("tiny", sc, vm, K, 70, 50)
    report("tiny-ED", sc, vm, K, 70, 50, render_mode="RGB+ED")
    sc, vm, K, W, H = scenes.config_c1()
    report("c1", sc, vm, K, W, H)
    if "--timing" in sys.argv:
        timing()
