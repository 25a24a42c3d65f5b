"""`from gsplat.distributed import cli` (reference: gs_init_compare/trainer.py:10, 58 --
`cli(main, cfg, verbose=True)`): the process launcher around the training entry point
`fn(local_rank, world_rank, world_size, args)`.

One process per GPU, `torch.distributed` backend "nccl" (= RCCL on ROCm), rendezvous on
127.0.0.1. Three ways in, as with gsplat's launcher: already inside a `torchrun` job
(RANK / LOCAL_RANK / WORLD_SIZE in the environment), one visible GPU (plain call), or several
visible GPUs (one spawned process each). Only the launcher is provided here; gsplat's
Gaussian-sharded all-to-all helpers are not (this build is view-parallel).
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(local_rank, fn, world_size, args, port, world_rank=None):
    world_rank = local_rank if world_rank is None else world_rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=world_rank, world_size=world_size,
                            device_id=torch.device("cuda", local_rank))
    try:
        out = fn(local_rank, world_rank, world_size, args)
    except BaseException:
        # a failing rank must not wait in a barrier the healthy ranks never reach (they would
        # block in their own collectives): tear the group down and re-raise so that the
        # launcher (mp.spawn / torchrun) ends the whole job with a non-zero code
        try:
            dist.destroy_process_group()
        finally:
            raise
    dist.barrier()
    dist.destroy_process_group()
    return out


def cli(fn, args, verbose: bool = False):
    if not torch.cuda.is_available():
        raise RuntimeError("gsplat.distributed.cli: no ROCm device is visible; this build has no CPU path")
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:          # launched by torchrun
        world_size = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ["RANK"])
        local_rank = int(os.environ.get("LOCAL_RANK", rank))
        if verbose:
            print(f"[cli] torchrun rank {rank}/{world_size} on cuda:{local_rank}")
        return _worker(local_rank, fn, world_size, args, int(os.environ.get("MASTER_PORT", 29500)), rank)
    world_size = torch.cuda.device_count()
    if world_size == 1:
        return fn(0, 0, 1, args)
    port = _free_port()
    if verbose:
        print(f"[cli] spawning {world_size} processes, rendezvous 127.0.0.1:{port}")
    mp.spawn(_worker, args=(fn, world_size, args, port), nprocs=world_size, join=True)
