"""Two REAL ranks of the view-parallel path on the one GPU of the test box (gloo backend moving
CUDA tensors; RCCL refuses two ranks on one device): each rank renders its own camera through
the HIP path, the gradient arena is reduced in 4 asynchronous chunks, the fused Adam consumes
them chunk by chunk. After three steps both replicas must hold identical parameters, equal to a
single process stepping on the two-camera batch (sum of the per-view losses)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N, W, H = 2000, 96, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    from tests import scenes
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
    sc = scenes.make_scene(N, 1, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    vm, K = scenes.cameras(range(0, 60, 10), width=W, height=H, f=90.0, dist=2.5)
    c2w, K = torch.linalg.inv(vm).contiguous().cuda(), K.cuda()
    target = torch.rand(6, H, W, 3, generator=torch.Generator().manual_seed(3)).cuda()
    splats, opts = runner.create_splats_with_optimizers(
        sc["means"], torch.rand(N, 3, generator=torch.Generator().manual_seed(0)),
        torch.log(sc["scales"]), quats=sc["quats"], opacities_logit=torch.logit(sc["opacities"]),
        shN=sc["shN"], batch_size=1, world_size=2)
    return runner, D, splats, D.fuse_optimizers(splats, opts), c2w, K, target


def _worker_gather(rank, world, port, q, chunks=3, rows="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        runner, D, splats, fused, c2w, K, target = _setup()
        # 3 pipelined Gaussian ranges (boundaries on multiples of 64, the last one ragged)
        sync = D.GatherRowsSync(fused, world, rank, chunks=chunks, min_chunk=512, rows=rows)
        assert len(sync.chunk_bounds(N)) == chunks
        try:
            for step in range(3):
                cams = [D.shard_views(6, step, r, world) for r in range(world)]
                sync.set_views(c2w[cams], K[cams])
                cam = cams[rank]
                runner.train_step(splats, fused, c2w[cam:cam + 1], K[cam:cam + 1], target[cam:cam + 1],
                                  step=5000 + step, grad_sync=sync)
                assert all(p.grad is None for p in splats.values())
        finally:
            sync.close()
        torch.cuda.synchronize()
        q.put((rank, {k: p.detach().cpu().numpy() for k, p in splats.items()}, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
        runner, D, splats, fused, c2w, K, target = _setup()
        sync = D.GradSync(splats, world, chunks=4)
        sync.attach(fused)
        for step in range(3):
            cam = D.shard_views(6, step, rank, world)
            runner.train_step(splats, fused, c2w[cam:cam + 1], K[cam:cam + 1], target[cam:cam + 1],
                              step=5000 + step, grad_sync=sync)
        torch.cuda.synchronize()
        q.put((rank, {k: p.detach().cpu().numpy() for k, p in splats.items()}, "ok"))   # by value
        R.set_grad_arena(None)
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _two_camera_batch_reference():
    """One process, both cameras of each step in one batch (loss = sum of per-view means)."""
    runner, D, splats, fused, c2w, K, target = _setup()
    for step in range(3):
        cams = [D.shard_views(6, step, r, 2) for r in range(2)]
        for p in splats.values():
            p.grad = None
        total = 0
        for cam in cams:
            renders, alphas, info = runner.rasterize_splats(
                splats, c2w[cam:cam + 1], K[cam:cam + 1], W, H, sh_degree=3)
            total = total + (renders - target[cam:cam + 1]).abs().mean()
        total.backward()
        fused.step()
        fused.zero_grad(set_to_none=True)
    return {k: p.detach().cpu() for k, p in splats.items()}


def _run_two(worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, params, msg = q.get(timeout=300)
        assert msg == "ok", msg
        res[rank] = {k: torch.from_numpy(v) for k, v in params.items()}
    for p in procs:
        p.join(60)
    return res


def test_two_ranks_gathered_view_space_rows_match_two_camera_batch():
    """GatherRowsSync: each rank packs its 36-byte view-space gradient rows, one all-gather,
    every rank runs the projection backward + Adam over BOTH cameras. Replicas bit-identical,
    and equal to one process stepping on the two-camera batch."""
    res = _run_two(_worker_gather)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), f"replicas diverged in {k}"
    ref = _two_camera_batch_reference()
    for k in ref:
        assert torch.allclose(res[0][k], ref[k], rtol=1e-4, atol=1e-6), f"{k}: gathered rows != two-camera batch"


def _worker_gather_monolithic(rank, world, port, q):
    _worker_gather(rank, world, port, q, chunks=1)


def _worker_gather_fp16(rank, world, port, q):
    _worker_gather(rank, world, port, q, chunks=3, rows="fp16")


def test_two_ranks_half_precision_rows_within_tolerance():
    """GatherRowsSync(rows="fp16"): the rows travel as a shared exponent + nine IEEE halves (20 bytes
    instead of 36). Replicas stay bit-identical (every rank decodes the same bytes). Against the
    fp32 exchange, three Adam steps from fresh moments move every parameter by about lr per step
    whatever the gradient's size (m / sqrt(v) is +-1 at first), so a value that 11-bit rows round
    differently can shift a parameter by a fraction of lr, and a gradient at the edge of the half
    range (Adam turns even a 1e-12 gradient into a full step, eps = 1e-15) by a whole one: the bar is
    1 % of the three-step travel in the mean and at most one element in a thousand off by more than 10 %."""
    half = _run_two(_worker_gather_fp16)
    full = _run_two(_worker_gather)
    lrs = {"means": 1.6e-4, "scales": 5e-3, "quats": 1e-3, "opacities": 5e-2, "sh0": 2.5e-3, "shN": 2.5e-3 / 20}
    for k in half[0]:
        assert torch.equal(half[0][k], half[1][k]), f"replicas diverged in {k}"
        travel = 3 * lrs[k] * 2 ** 0.5                       # lr is scaled by sqrt(batch size 2)
        d = (half[0][k] - full[0][k]).abs()
        frac = float((d > 0.1 * travel).float().mean())
        assert float(d.mean()) <= 0.01 * travel and frac <= 1e-3, (k, float(d.max()), float(d.mean()), frac)


def test_pipelined_row_exchange_equals_monolithic_exchange():
    """the exchange split into 3 Gaussian ranges (all-gather of range k+1 in
    flight during the projection backward + Adam of range k) against the single all-gather +
    single backward launch. Replicas of one run are bit-identical (same gathered rows, same
    per-Gaussian arithmetic); two RUNS differ in the last bits because the compositing
    backward's float atomics land in a different order, so across runs the bar is 1e-4."""
    pipe = _run_two(_worker_gather)
    mono = _run_two(_worker_gather_monolithic)
    for k in pipe[0]:
        assert torch.equal(pipe[0][k], pipe[1][k]) and torch.equal(mono[0][k], mono[1][k]), k
        assert torch.allclose(pipe[0][k], mono[0][k], rtol=1e-4, atol=1e-6), k


def test_two_ranks_one_gpu_pipelined_allreduce_matches_two_camera_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, params, msg = q.get(timeout=300)
        if params is None and ("CUDA tensor" in msg or "not supported" in msg or "gloo" in msg.lower() and "cuda" in msg.lower()):
            for p in procs:
                p.join(10)
            pytest.skip("gloo in this build cannot move CUDA tensors: " + msg.splitlines()[-1])
        assert msg == "ok", msg
        res[rank] = {k: torch.from_numpy(v) for k, v in params.items()}
    for p in procs:
        p.join(60)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), f"replicas diverged in {k}"
    # single process, both cameras of each step in one batch (loss = sum of per-view means)
    runner, D, splats, fused, c2w, K, target = _setup()
    for step in range(3):
        cams = [D.shard_views(6, step, r, 2) for r in range(2)]
        for p in splats.values():
            p.grad = None
        total = 0
        for cam in cams:
            renders, alphas, info = runner.rasterize_splats(
                splats, c2w[cam:cam + 1], K[cam:cam + 1], W, H, sh_degree=3)
            total = total + (renders - target[cam:cam + 1]).abs().mean()
        total.backward()
        fused.step()
        fused.zero_grad(set_to_none=True)
    for k, p in splats.items():
        ref = p.detach().cpu()
        assert torch.allclose(res[0][k], ref, rtol=1e-4, atol=1e-6), f"{k}: two ranks != two-camera batch"


def _worker_strategy(rank, world, port, q):
    """View-space row exchange + DefaultStrategy: refine steps (one-pass densification) and an opacity
    reset inside the window; the fused update is suspended on those steps (runner.train_step)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = importlib.import_module("3dgs_monocular_depth_init_amd.strategy")
        runner, D, splats, fused, c2w, K, target = _setup()
        sync = D.GatherRowsSync(fused, world, rank, chunks=2, min_chunk=512)
        strat = S.DefaultStrategy(refine_start_iter=0, refine_every=3, reset_every=6, grow_grad2d=1e-5,
                                  grow_scale3d=0.03, prune_opa=0.06)
        state = strat.initialize_state(scene_scale=1.0)
        strat.check_sanity(splats, fused)
        sizes = []
        try:
            for step in range(1, 11):
                cams = [D.shard_views(6, step, r, world) for r in range(world)]
                sync.set_views(c2w[cams], K[cams])
                cam = cams[rank]
                runner.train_step(splats, fused, c2w[cam:cam + 1], K[cam:cam + 1], target[cam:cam + 1],
                                  step=step, grad_sync=sync, strategy=strat, strategy_state=state)
                sizes.append(len(splats["means"]))
        finally:
            sync.close()
        torch.cuda.synchronize()
        out = {k: p.detach().cpu().numpy() for k, p in splats.items()}
        out["sizes"] = torch.tensor(sizes).numpy()
        q.put((rank, out, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_replicas_stay_identical_through_densification():
    """Two ranks, ten steps with three refine steps (duplicate / split / prune in one pass on the
    GPU, decisions from all-reduced statistics, split noise from identically seeded generators) and
    an opacity reset: the replicas must hold bit-identical parameters of the same, changed size."""
    res = _run_two(_worker_strategy)
    a, b = res[0], res[1]
    assert torch.equal(a["sizes"], b["sizes"])
    assert len(set(a["sizes"].tolist())) >= 3 and int(a["sizes"][-1]) != N, a["sizes"]
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k


# ------------------------------------------------------------------------------------------------- #
# W = 8, functionally, on one GPU: one process plays the eight ranks in turn. GatherRowsSync's own
# code runs unchanged (chunk_bounds, packing, the chunked gsr_project_bwd_adam(C = 8) loop of
# rendering._ProjectSH.backward); only the collective is stood in for (`_all_gather`).
# ------------------------------------------------------------------------------------------------- #
def _world8_run(rows_mode):
    """Returns (params after one 8-view step through the chunked exchange, as rank 0 computes
    them; the same from rank 5's point of view)."""
    from tests import scenes
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
    R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
    WORLD, n8 = 8, 6000
    sc = scenes.make_scene(n8, 2, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    vm, K = scenes.cameras(range(0, 96, 12), width=W, height=H, f=90.0, dist=2.5)
    c2w, K = torch.linalg.inv(vm).contiguous().cuda(), K.cuda()
    target = torch.rand(WORLD, H, W, 3, generator=torch.Generator().manual_seed(3)).cuda()

    def fresh():
        splats, opts = runner.create_splats_with_optimizers(
            sc["means"], torch.rand(n8, 3, generator=torch.Generator().manual_seed(0)),
            torch.log(sc["scales"]), quats=sc["quats"], opacities_logit=torch.logit(sc["opacities"]),
            shN=sc["shN"], batch_size=1, world_size=WORLD)
        return splats, D.fuse_optimizers(splats, opts)

    recorded = {}          # (rank, start, n) -> that rank's packed rows of the range

    class OneOfEight(D.GatherRowsSync):
        record_only = True

        def _all_gather(self, buf, mine, start, n):
            if self.record_only:
                recorded[(self.rank, start, n)] = mine.clone()
            for r in range(self.world):
                # final pass: every segment -- this rank's own too -- is the row set recorded for that
                # rank, as in a real exchange where a rank's rows exist once (recomputing them would
                # differ in the last bits: the compositing atomics are order-nondeterministic)
                if r != self.rank or not self.record_only:
                    seg = recorded.get((r, start, n))
                    buf[r * n:(r + 1) * n] = seg if seg is not None else 0   # recording pass: others unknown yet
            return lambda: None

    def one_rank_step(rank, record_only):
        """Rank `rank`'s step. Recording pass: the gathered buffer is incomplete, so the update is
        thrown away (fresh parameters every time) -- only this rank's packed rows are kept."""
        splats, fused = fresh()
        sync = OneOfEight(fused, WORLD, rank, chunks=4, min_chunk=512, rows=rows_mode)
        sync.record_only = record_only
        assert len(sync.chunk_bounds(n8)) == 4
        try:
            sync.set_views(c2w, K)
            runner.train_step(splats, fused, c2w[rank:rank + 1], K[rank:rank + 1], target[rank:rank + 1],
                              step=7000, grad_sync=sync)
        finally:
            sync.close()
        torch.cuda.synchronize()
        return {k: p.detach().clone() for k, p in splats.items()}

    for r in range(WORLD):
        one_rank_step(r, True)
    assert len(recorded) == WORLD * 4
    return one_rank_step(0, False), one_rank_step(5, False), (runner, D, fresh, c2w, K, target, n8)


def test_world8_chunked_exchange_equals_eight_camera_batch():
    """Eight views per step through GatherRowsSync's chunked exchange (4 Gaussian ranges, C = 8 in
    gsr_project_bwd_adam) give (a) the same parameters on every rank (checked for ranks 0 and 5:
    bit-identical, both sum the views in rank order), (b) the parameters one process gets from the
    eight-camera batch (loss = sum of the per-view means) with the ordinary fused step."""
    p0, p5, (runner, D, fresh, c2w, K, target, n8) = _world8_run("fp32")
    for k in p0:
        assert torch.equal(p0[k], p5[k]), f"ranks 0 and 5 disagree in {k}"
    splats, fused = fresh()
    renders, _, _ = runner.rasterize_splats(splats, c2w, K, W, H, sh_degree=3)      # C = 8 in one call
    (renders - target).abs().mean(dim=(1, 2, 3)).sum().backward()
    fused.step()
    torch.cuda.synchronize()
    for k in p0:
        ref = splats[k].detach()
        assert torch.allclose(p0[k], ref, rtol=1e-4, atol=1e-6), f"{k}: chunked W=8 exchange != 8-camera batch"
        assert float((ref - fresh()[0][k].detach()).abs().max()) > 0, f"{k} did not move"


def test_world8_half_rows_within_tolerance():
    """The same with rows="fp16" (20-byte rows): ranks still bit-identical. Against the fp32 exchange:
    the FIRST Adam step moves every element by about lr * sign(gradient), so an element whose
    gradient is within half-precision rounding of zero can land on the other side (a difference of
    twice the travel); all but a few per cent of the elements must agree to a quarter of the travel."""
    h0, h5, _ = _world8_run("fp16")
    f0, _, (runner, D, fresh, *_rest) = _world8_run("fp32")
    init = {k: p.detach() for k, p in fresh()[0].items()}
    for k in h0:
        assert torch.equal(h0[k], h5[k]), f"ranks 0 and 5 disagree in {k}"
        travel = float((f0[k] - init[k]).abs().max())
        off = ((h0[k] - f0[k]).abs() > 0.25 * travel).float().mean()
        assert float(off) <= 0.05, (k, float(off))
        assert float((h0[k] - f0[k]).abs().max()) <= 2.0 * travel * (1 + 1e-3) + 1e-12, k
