"""World-size-2 CPU (gloo) tests of the view-parallel path: camera sharding,
gradient all-reduce (sum), and replica consistency after identical Adam steps.
The data path has exactly one collective (the gradient all-reduce)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
        torch.manual_seed(0)                                   # identical replicas
        N = 257
        shapes = {"means": (N, 3), "scales": (N, 3), "quats": (N, 4), "opacities": (N,),
                  "sh0": (N, 1, 3), "shN": (N, 15, 3)}
        splats = torch.nn.ParameterDict({k: torch.nn.Parameter(torch.randn(s)) for k, s in shapes.items()})
        opts = {k: torch.optim.Adam([splats[k]], lr=1e-2) for k in splats}
        sync = D.GradSync(splats, world)
        cams = []
        for step in range(3):
            cams.append(D.shard_views(100, step, rank, world))
            g = torch.Generator().manual_seed(1000 * step + rank)  # rank-dependent "view" gradient
            for k, p in splats.items():
                p.grad = torch.randn(p.shape, generator=g)
            local = {k: p.grad.clone() for k, p in splats.items()}
            sync()
            # expected = sum over ranks of each rank's local gradient
            for k, p in splats.items():
                exp = torch.zeros_like(p)
                for r in range(world):
                    gr = torch.Generator().manual_seed(1000 * step + r)
                    for kk, pp in splats.items():
                        t = torch.randn(pp.shape, generator=gr)
                        if kk == k:
                            exp += t
                assert torch.allclose(p.grad, exp, atol=1e-6), (k, step)
            for o in opts.values():
                o.step()
        flat = torch.cat([p.detach().reshape(-1) for p in splats.values()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], x) for x in gathered), "replicas diverged"
        q.put((rank, cams, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, None, repr(e)))
    finally:
        dist.destroy_process_group()


def test_grad_allreduce_and_replica_consistency_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    res.sort()
    assert all(r[2] == "ok" for r in res), res
    # view sharding: step k -> cameras (k*W + r) % n; disjoint across ranks per step
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3, 5]


def test_shard_views_covers_dataset():
    D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
    seen = set()
    for step in range(25):
        for r in range(4):
            seen.add(D.shard_views(100, step, r, 4))
    assert seen == set(range(100))
    perm = list(reversed(range(10)))
    assert D.shard_views(10, 0, 1, 2, perm) == 8


def test_lr_scaling_rule_matches_reference():
    """runner.py:128-137: lr*sqrt(BS), eps/sqrt(BS), betas 1-BS(1-b)."""
    R = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    pts = torch.rand(10, 3)
    splats, opts = R.create_splats_with_optimizers(pts, torch.rand(10, 3), torch.zeros(10, 3),
                                                   device="cpu", world_size=4, batch_size=1)
    g = opts["means"].param_groups[0]
    assert g["lr"] == pytest.approx(1.6e-4 * 2.0)
    assert g["eps"] == pytest.approx(1e-15 / 2.0)
    assert g["betas"] == (pytest.approx(1 - 4 * 0.1), pytest.approx(1 - 4 * 0.001))
    assert opts["shN"].param_groups[0]["lr"] == pytest.approx(2.5e-3 / 20 * 2.0)
    assert set(splats.keys()) == {"means", "scales", "quats", "opacities", "sh0", "shN"}
    assert splats["sh0"].shape == (10, 1, 3) and splats["shN"].shape == (10, 15, 3)


def test_chunk_to_parameter_pieces_cover_every_element_once():
    """The pipelined optimizer steps arbitrary flat ranges of the gradient arena: the
    range -> (parameter, offset, count) mapping must tile every parameter exactly."""
    optim = importlib.import_module("3dgs_monocular_depth_init_amd.optim")
    N = 1237
    numel = {"shN": 45 * N, "sh0": 3 * N, "means": 3 * N, "quats": 4 * N, "scales": 3 * N, "opacities": N}
    segs, off = [], 0
    for k, n in numel.items():                       # arena layout: 16-byte aligned segments
        segs.append((k, off, n))
        off += (n + 3) // 4 * 4
    total = off
    for chunks in (1, 3, 4, 7):
        step = (total // chunks + 3) // 4 * 4
        bounds, a = [], 0
        while a < total:
            b = min(total, a + step) if len(bounds) < chunks - 1 else total
            bounds.append((a, b)); a = b
        seen = {k: torch.zeros(n, dtype=torch.int32) for k, n in numel.items()}
        for a, b in bounds:
            pieces = optim.FusedAdam.pieces_for_range(a, b, segs)
            assert len(pieces) <= 8
            for k, start, cnt in pieces:
                assert start % 4 == 0 and cnt > 0
                seen[k][start:start + cnt] += 1
        assert all(bool((v == 1).all()) for v in seen.values())


def _worker8(rank, world, port, q):
    """World size 8 (gloo, CPU): the host logic of the view-parallel step at the size the driver's
    scaling run uses -- camera sharding incl. the wrap-around of bench.py's camera table, the
    Gaussian-range chunking, and the segment layout of the row exchange (rank r's rows in
    rows_all[r*n:(r+1)*n] on every rank, summed in rank order)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
        n_cams = 100
        table = torch.arange(n_cams)
        table_w = torch.cat([table, table[:world]])            # bench.py: c2ws_w = cat([c2ws, c2ws[:world]])
        mine = []
        for step in range(30):                                 # 240 views: the table wraps twice
            a = (step * world) % n_cams
            views = table_w[a:a + world]                       # the slice every rank hands to set_views
            cam = D.shard_views(n_cams, step, rank, world)
            assert int(views[rank]) == cam == (step * world + rank) % n_cams
            assert views.numel() == world and len(set(views.tolist())) == world
            mine.append(cam)
        # chunking of the Gaussian range (no GPU needed: chunk_bounds is host arithmetic)
        sync = object.__new__(D.GatherRowsSync)
        sync.chunks, sync.min_chunk = 4, 4096
        for N in (1_000_000, 999_999, 4096, 4097, 16385, 100, 64, 1):
            cb = sync.chunk_bounds(N)
            assert cb[0][0] == 0 and cb[-1][1] == N and len(cb) <= 4
            assert all(b0 == a1 for (_, b0), (a1, _) in zip(cb[:-1], cb[1:]))
            assert all(a % 64 == 0 for a, _ in cb) and all(b > a for a, b in cb)
        # exchange layout: every rank contributes rows that encode (rank, Gaussian); after the
        # all-gather the sum over segments in rank order is the same tensor on all ranks
        N = 1000
        for a, b in sync.chunk_bounds(N) if False else [(0, 448), (448, 1000)]:
            n = b - a
            buf = torch.zeros(world * n, 9)
            seg = (torch.arange(a, b, dtype=torch.float32)[:, None] * 0.001 + rank + 1).expand(n, 9).contiguous()
            outs = [buf[r * n:(r + 1) * n] for r in range(world)]
            dist.all_gather(outs, seg)
            for r in range(world):
                assert torch.equal(buf[r * n:(r + 1) * n, 0], torch.arange(a, b, dtype=torch.float32) * 0.001 + r + 1)
            total = torch.zeros(n, 9)
            for r in range(world):
                total += buf[r * n:(r + 1) * n]
            ref = [torch.zeros_like(total) for _ in range(world)]
            dist.all_gather(ref, total)
            assert all(torch.equal(ref[0], x) for x in ref)
        q.put((rank, mine, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world8_view_sharding_chunking_and_exchange_layout():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    res.sort()
    assert all(r[2] == "ok" for r in res), [r[2] for r in res if r[2] != "ok"][:1]
    # per step the eight ranks cover eight distinct consecutive cameras (mod 100)
    for step in range(30):
        cams = [res[r][1][step] for r in range(world)]
        assert cams == [(step * world + r) % 100 for r in range(world)]
