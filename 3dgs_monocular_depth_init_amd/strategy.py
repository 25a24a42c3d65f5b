"""Densification strategy: the step either side of the rasterizer.

`DefaultStrategy` keeps the interface the reference drives
(/root/reference/gs_init_compare/runner.py:208-217 `check_sanity` /
`initialize_state`, :497-503 `step_pre_backward`, :639-647
`step_post_backward`; defaults scaled by config.py:204-221) -- i.e. gsplat's
`gsplat.strategy.DefaultStrategy`, third-party and absent here (parity unpinned:
restated from gsplat's published behaviour).

What it consumes from the hot path: `info["means2d"].grad` (pixel-space
gradient of the 2-D means, populated because `rasterization()` returns
`means2d` as an autograd intermediate), `info["radii"]`, `width`, `height`,
`n_cameras`.

Multi-GPU replicas (distributed.py): every rank must take the SAME
densification decisions, so the accumulated statistics are all-reduced before
use and the split noise comes from a generator seeded identically on all ranks
(the reference seeds ranks differently, runner.py:147 -- right for its
Gaussian-sharded scheme, wrong for replicas).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Callable, Dict, Optional

import torch
import torch.distributed as dist
from torch import Tensor
from ._lib import current_stream as _raw_stream


def _quat_to_rotmat(q: Tensor) -> Tensor:
    q = torch.nn.functional.normalize(q, dim=-1)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def _opt_of(optimizers, name):
    return optimizers[name]


@torch.no_grad()
def _update_param_with_optimizer(param_fn: Callable[[str, Tensor], Tensor],
                                 optimizer_fn: Callable[[str, Tensor], Tensor], params, optimizers,
                                 names=None) -> None:
    """Replace every parameter by param_fn(name, p) and re-key its Adam state
    (exp_avg / exp_avg_sq mapped through optimizer_fn, `step` kept)."""
    names = list(params.keys()) if names is None else names
    for name in names:
        old = params[name]
        new = torch.nn.Parameter(param_fn(name, old), requires_grad=old.requires_grad)
        opt = _opt_of(optimizers, name)
        for g in opt.param_groups:
            for i, p in enumerate(g["params"]):
                if p is old:
                    st = opt.state.pop(p, {})
                    for k in list(st.keys()):
                        if k != "step":
                            st[k] = optimizer_fn(k, st[k])
                    g["params"][i] = new
                    if st:
                        opt.state[new] = st
        params[name] = new


@torch.no_grad()
def duplicate(params, optimizers, state: Dict[str, Tensor], mask: Tensor) -> None:
    sel = torch.where(mask)[0]
    _update_param_with_optimizer(
        lambda n, p: torch.cat([p, p[sel]]),
        lambda k, v: torch.cat([v, torch.zeros((len(sel), *v.shape[1:]), device=v.device, dtype=v.dtype)]),
        params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor) and v.dim() > 0:
            state[k] = torch.cat([v, v[sel]])


@torch.no_grad()
def split(params, optimizers, state: Dict[str, Tensor], mask: Tensor, revised_opacity: bool = False,
          generator: Optional[torch.Generator] = None) -> None:
    dev = mask.device
    sel, rest = torch.where(mask)[0], torch.where(~mask)[0]
    scales = torch.exp(params["scales"][sel])
    rotmats = _quat_to_rotmat(params["quats"][sel])
    noise = torch.randn(2, len(scales), 3, device=dev, generator=generator)
    samples = torch.einsum("nij,nj,bnj->bni", rotmats, scales, noise)

    def param_fn(name: str, p: Tensor) -> Tensor:
        reps = [2] + [1] * (p.dim() - 1)
        if name == "means":
            p_split = (p[sel] + samples).reshape(-1, 3)
        elif name == "scales":
            p_split = torch.log(scales / 1.6).repeat(2, 1)
        elif name == "opacities" and revised_opacity:
            new_o = 1.0 - torch.sqrt(1.0 - torch.sigmoid(p[sel]))
            p_split = torch.logit(new_o).repeat(reps)
        else:
            p_split = p[sel].repeat(reps)
        return torch.cat([p[rest], p_split])

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        return torch.cat([v[rest], torch.zeros((2 * len(sel), *v.shape[1:]), device=dev, dtype=v.dtype)])

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor) and v.dim() > 0:
            reps = [2] + [1] * (v.dim() - 1)
            state[k] = torch.cat([v[rest], v[sel].repeat(reps)])


@torch.no_grad()
def remove(params, optimizers, state: Dict[str, Tensor], mask: Tensor) -> None:
    keep = torch.where(~mask)[0]
    _update_param_with_optimizer(lambda n, p: p[keep], lambda k, v: v[keep], params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor) and v.dim() > 0:
            state[k] = v[keep]


@torch.no_grad()
def reset_opa(params, optimizers, state: Dict[str, Tensor], value: float) -> None:
    max_logit = math.log(value / (1.0 - value))
    p = params["opacities"]
    if p.is_cuda and p.is_contiguous() and p.dtype == torch.float32:
        # in place, one launch: the clamp and the two moments (gsr_reset_opacity)
        stt = _opt_of(optimizers, "opacities").state.get(p, {})
        m, v = stt.get("exp_avg"), stt.get("exp_avg_sq")
        if all(t is None or (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == p.numel()) for t in (m, v)):
            from ._lib import ptr
            _call()("gsr_reset_opacity", p.numel(), ptr(p.detach()), ptr(m), ptr(v), float(max_logit),
                    _raw_stream())
            return
    _update_param_with_optimizer(
        lambda n, p: torch.clamp(p, max=max_logit),
        lambda k, v: torch.zeros_like(v), params, optimizers, names=["opacities"])


@torch.no_grad()
def _rebuild_by_gather(params, optimizers, M: int, src: Tensor, kind: Tensor,
                       src_m: Optional[Tensor] = None, kind_m: Optional[Tensor] = None) -> None:
    """Every parameter and both of its Adam moments rebuilt with M rows by `gsr_refine_gather` (the
    multi-tensor row gather of the one-pass refine step): parameter rows new[r] = old[src[r]]; moment
    rows new[r] = old[src_m[r]], or zero where kind_m[r] != 0 (src_m / kind_m default to src / kind).
    Then the tensors are handed to the parameter dict and the optimizers as
    `_update_param_with_optimizer` does."""
    import ctypes as C

    from ._lib import ptr
    call = _call()
    dev = src.device
    st = _raw_stream()
    src_m = src if src_m is None else src_m
    kind_m = kind if kind_m is None else kind_m
    N = len(next(iter(params.values())))
    p_jobs, m_jobs, new_params, new_moments = [], [], {}, {}
    for name, p in params.items():
        old = p.detach()
        if not (old.is_contiguous() and old.dtype == torch.float32):
            raise ValueError(f"rebuild: parameter {name!r} must be contiguous fp32")
        L = max(1, int(old.numel() // max(N, 1)))
        new_params[name] = torch.empty((M, *old.shape[1:]), dtype=torch.float32, device=dev)
        p_jobs.append((old, new_params[name], L, 0))
        for k, v in _opt_of(optimizers, name).state.get(p, {}).items():
            if k != "step":
                nv = torch.empty((M, *v.shape[1:]), dtype=torch.float32, device=dev)
                new_moments[(name, k)] = nv
                m_jobs.append((v.contiguous(), nv, L, 1))
    if M > 0 and N > 0:
        for jobs, s_, k_ in ((p_jobs, src, kind), (m_jobs, src_m, kind_m)):
            for a in range(0, len(jobs), 24):
                part = jobs[a:a + 24]
                n = len(part)
                PA, IA = C.c_void_p * n, C.c_int32 * n
                call("gsr_refine_gather", n, M, ptr(s_), ptr(k_), PA(*[ptr(j[0]) for j in part]),
                     PA(*[ptr(j[1]) for j in part]), IA(*[j[2] for j in part]), IA(*[j[3] for j in part]), st)
    for name in list(params.keys()):
        old = params[name]
        new = torch.nn.Parameter(new_params[name], requires_grad=old.requires_grad)
        opt = _opt_of(optimizers, name)
        for g in opt.param_groups:
            for i, p in enumerate(g["params"]):
                if p is old:
                    stt = opt.state.pop(p, {})
                    for k in list(stt.keys()):
                        if k != "step":
                            stt[k] = new_moments[(name, k)]
                    g["params"][i] = new
                    if stt:
                        opt.state[new] = stt
        params[name] = new


def _gather_ok(params) -> bool:
    return all(p.is_cuda and p.is_contiguous() and p.dtype == torch.float32 for p in params.values())


@dataclass
class DefaultStrategy:
    prune_opa: float = 0.005
    grow_grad2d: float = 0.0002
    grow_scale3d: float = 0.01
    grow_scale2d: float = 0.05
    prune_scale3d: float = 0.1
    prune_scale2d: float = 0.15
    refine_scale2d_stop_iter: int = 0
    refine_start_iter: int = 500
    refine_stop_iter: int = 15_000
    reset_every: int = 3000
    refine_every: int = 100
    pause_refine_after_reset: int = 0
    absgrad: bool = False
    revised_opacity: bool = False
    verbose: bool = False
    key_for_gradient: str = "means2d"
    seed: int = 42                      # shared by all replicas (see module docstring)
    one_pass: bool = True               # GPU: decide -> scan -> one multi-tensor gather (_refine_one_pass)

    def initialize_state(self, scene_scale: float = 1.0) -> Dict[str, Any]:
        return {"grad2d": None, "count": None, "scene_scale": scene_scale, "radii": None,
                "generator": None}

    def check_sanity(self, params, optimizers) -> None:
        for key in ("means", "scales", "quats", "opacities"):
            assert key in params, f"{key} is required in params"
        assert set(params.keys()) <= set(optimizers.keys()), "every parameter needs an optimizer"

    def step_pre_backward(self, params, optimizers, state, step: int, info: Dict[str, Any]) -> None:
        assert self.key_for_gradient in info, "The 2D means of the Gaussians is required but missing."
        m2d = info[self.key_for_gradient]
        if getattr(m2d, "_gsr_grad_in_backward", False):
            # this package's rasterization: its backward leaves `.grad` behind itself, a view into the 64-byte gradient
            # rows (retain_grad's hook clones the gradient: one strided copy launch per step)
            m2d._gsr_keep_grad = True
        else:
            m2d.retain_grad()

    def mutates_params(self, step: int) -> bool:
        """True on the steps whose step_post_backward edits parameters / optimizer state
        (refine and opacity-reset steps): those must run strategy-before-optimizer as in
        runner.py:638-679, so the optimizer-in-backward fusion is suspended for them."""
        if step >= self.refine_stop_iter:
            return False
        refine = (step > self.refine_start_iter and step % self.refine_every == 0
                  and step % self.reset_every >= self.pause_refine_after_reset)
        return refine or (step % self.reset_every == 0 and step > 0)

    def stats_for_fused_backward(self, params, state, info):
        """The launch arguments of this step's statistics for a backward that takes them along
        (optim.FusedAdam.set_step_extras(stats=...)): (grad2d, count, radii_state or None, sx, sy, 1 / max(W, H),
        absgrad), the accumulators allocated as `_update_state` would. None when they do not apply."""
        if self.key_for_gradient != "means2d" or not params["means"].is_cuda:
            return None
        n = len(params["means"])
        dev = params["means"].device
        if state["grad2d"] is None:
            state["grad2d"] = torch.zeros(n, device=dev)
            state["count"] = torch.zeros(n, device=dev)
        if self.refine_scale2d_stop_iter > 0 and state["radii"] is None:
            state["radii"] = torch.zeros(n, device=dev)
        return (state["grad2d"], state["count"], state["radii"] if self.refine_scale2d_stop_iter > 0 else None,
                info["width"] / 2.0 * info["n_cameras"], info["height"] / 2.0 * info["n_cameras"],
                1.0 / float(max(info["width"], info["height"])), bool(self.absgrad))

    def step_post_backward(self, params, optimizers, state, step: int, info: Dict[str, Any],
                           packed: bool = False, stats_done: bool = False) -> None:
        if step >= self.refine_stop_iter:
            return
        if not stats_done:          # (True: the fused backward accumulated them, runner.train_step)
            self._update_state(params, state, info, packed=packed)
        if (step > self.refine_start_iter and step % self.refine_every == 0
                and step % self.reset_every >= self.pause_refine_after_reset):
            self._sync_state(state)
            if self.one_pass and params["means"].is_cuda:
                n_dupli, n_split, n_prune = self._refine_one_pass(params, optimizers, state, step)
            else:          # tensor-op formulation (CPU tensors: the unit tests of the bookkeeping)
                n_dupli, n_split = self._grow_gs(params, optimizers, state, step)
                n_prune = self._prune_gs(params, optimizers, state, step)
            if self.verbose:
                print(f"Step {step}: {n_dupli} GSs duplicated, {n_split} GSs split, {n_prune} GSs "
                      f"pruned. Now having {len(params['means'])} GSs.")
            state["grad2d"].zero_()
            state["count"].zero_()
            if state["radii"] is not None:
                state["radii"].zero_()
        if step % self.reset_every == 0 and step > 0:
            reset_opa(params, optimizers, state, value=self.prune_opa * 2.0)

    # ------------------------------------------------------------------ internals
    def _update_state(self, params, state, info, packed: bool = False) -> None:
        for key in ("width", "height", "n_cameras", "radii", self.key_for_gradient):
            assert key in info, f"{key} is required but missing."
        m2d = info[self.key_for_gradient]
        g = m2d.absgrad if self.absgrad else m2d.grad
        n = len(params["means"])
        dev = g.device
        if state["grad2d"] is None:
            state["grad2d"] = torch.zeros(n, device=dev)
            state["count"] = torch.zeros(n, device=dev)
        if self.refine_scale2d_stop_iter > 0 and state["radii"] is None:
            state["radii"] = torch.zeros(n, device=dev)
        radii = info["radii"]
        sx = info["width"] / 2.0 * info["n_cameras"]
        sy = info["height"] / 2.0 * info["n_cameras"]
        max_wh = float(max(info["width"], info["height"]))
        if g.is_cuda:
            # one launch, no host sync (the torch formulation below needs a nonzero()); the
            # screen-space radius statistic is the maximum over the cameras of the batch (the
            # indexed assignment below keeps an arbitrary camera's value when C > 1)
            C = radii.shape[0]
            if g.dim() == 3 and g.stride(-1) == 1 and g.stride(0) == n * g.stride(1):
                stride = g.stride(1)                 # dense (2) or the view into the 64-byte rows (16)
            else:
                g = g.contiguous()
                stride = 2
            rad = radii if radii.is_contiguous() else radii.contiguous()
            rs = state["radii"] if self.refine_scale2d_stop_iter > 0 else None
            _call()("gsr_strategy_accumulate", C, n, g.data_ptr(), stride, rad.data_ptr(), float(sx),
                    float(sy), state["grad2d"].data_ptr(), state["count"].data_ptr(),
                    rs.data_ptr() if rs is not None else None, max_wh,
                    _raw_stream())
            return
        # host-side formulation (CPU tensors: the unit tests of the schedule and bookkeeping)
        grads = g.clone()
        grads[..., 0] *= sx
        grads[..., 1] *= sy
        sel = (radii > 0).all(dim=-1)                       # [C,N]
        gs_ids = torch.where(sel)[1]
        gsel = grads[sel]
        state["grad2d"].index_add_(0, gs_ids, gsel.norm(dim=-1))
        state["count"].index_add_(0, gs_ids, torch.ones_like(gs_ids, dtype=torch.float32))
        if self.refine_scale2d_stop_iter > 0:
            r = radii[sel].max(dim=-1).values.float() / max_wh
            state["radii"][gs_ids] = torch.maximum(state["radii"][gs_ids], r)

    def _sync_state(self, state) -> None:
        """Replicas: sum the statistics over ranks so every rank decides identically."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(state["grad2d"])
            dist.all_reduce(state["count"])
            if state["radii"] is not None:
                dist.all_reduce(state["radii"], op=dist.ReduceOp.MAX)

    def _generator(self, state, device) -> torch.Generator:
        if state.get("generator") is None:
            g = torch.Generator(device=device)
            g.manual_seed(self.seed)
            state["generator"] = g
        return state["generator"]

    @torch.no_grad()
    def _refine_one_pass(self, params, optimizers, state, step: int):
        """duplicate -> split -> prune of a refine step as ONE rebuild of the tensors (csrc/train_ops.hip):
        `gsr_refine_decide` takes every per-Gaussian decision (including the prune test on the
        values a duplicate / a split child will carry), one scan gives the output positions, one host
        read gives the new size, and `gsr_refine_gather` rebuilds all parameters and Adam moments
        in a single launch -- instead of three passes of torch.cat / indexing over the 6 + 12
        tensors with four host round trips. Same result as `_grow_gs` + `_prune_gs`, same row
        order, same draws from the split generator (tests/test_gpu_train_ops.py)."""
        from ._lib import ptr
        call = _call()
        dev = params["means"].device
        st = _raw_stream()
        N = len(params["means"])
        scene = state["scene_scale"]
        use2d = step < self.refine_scale2d_stop_iter
        scales, opac = params["scales"].detach(), params["opacities"].detach()
        assert scales.is_contiguous() and opac.is_contiguous() and opac.numel() == N
        flags = torch.empty(5, N, dtype=torch.int32, device=dev)
        call("gsr_refine_decide", N, ptr(scales), ptr(opac), ptr(state["grad2d"]), ptr(state["count"]),
             ptr(state["radii"]) if use2d else None, float(self.grow_grad2d), float(self.grow_scale3d * scene),
             float(self.grow_scale2d) if use2d else -1.0, float(self.prune_opa),
             float(self.prune_scale3d * scene) if step > self.reset_every else -1.0,
             float(self.prune_scale2d) if use2d else -1.0, int(self.revised_opacity), ptr(flags), st)
        incl = torch.empty_like(flags)
        for r in range(5):          # (row by row: torch's scan of a few LONG rows in one call runs one workgroup per row, 3 ms at 1 M)
            torch.cumsum(flags[r], 0, dtype=torch.int32, out=incl[r])
        n0, n1, n2, ns, nd = (incl[:, -1].tolist() if N else (0, 0, 0, 0, 0))   # the one host round trip
        M = n0 + n1 + 2 * n2
        src = torch.empty(M, dtype=torch.int32, device=dev)
        kind = torch.empty(M, dtype=torch.uint8, device=dev)
        if M > 0:
            call("gsr_refine_plan", N, ptr(flags), ptr(incl), n0, n1, n2, ptr(src), ptr(kind), st)
        # the split noise: drawn for EVERY split Gaussian, in index order, as split() draws it
        samples = sel = None
        if ns > 0:
            sel = torch.nonzero_static(flags[3], size=ns).flatten()
            sc_sel = torch.exp(scales[sel])
            noise = torch.randn(2, ns, 3, device=dev, generator=self._generator(state, dev))
            samples = torch.einsum("nij,nj,bnj->bni", _quat_to_rotmat(params["quats"].detach()[sel]), sc_sel, noise)
        # every parameter and its two Adam moments through one gather launch
        jobs, new_params, new_moments = [], {}, {}
        for name, p in params.items():
            old = p.detach()
            if not (old.is_contiguous() and old.dtype == torch.float32):
                raise ValueError(f"refine: parameter {name!r} must be contiguous fp32")
            L = old[0].numel() if (old.dim() > 1 and N > 0) else max(1, int(old.numel() // max(N, 1)))
            new_params[name] = torch.empty((M, *old.shape[1:]), dtype=torch.float32, device=dev)
            jobs.append((old, new_params[name], L, 0))
            stt = _opt_of(optimizers, name).state.get(p, {})
            for k, v in stt.items():
                if k != "step":
                    nv = torch.empty((M, *v.shape[1:]), dtype=torch.float32, device=dev)
                    new_moments[(name, k)] = nv
                    jobs.append((v.contiguous(), nv, L, 1))
        if M > 0 and N > 0:
            import ctypes as C
            for a in range(0, len(jobs), 24):
                part = jobs[a:a + 24]
                n = len(part)
                PA = C.c_void_p * n
                IA = C.c_int32 * n
                call("gsr_refine_gather", n, M, ptr(src), ptr(kind), PA(*[ptr(j[0]) for j in part]),
                     PA(*[ptr(j[1]) for j in part]), IA(*[j[2] for j in part]), IA(*[j[3] for j in part]), st)
        if n2 > 0:          # what a split child differs in from its parent (rows [a, a + 2 n2): sample 0 block, sample 1 block)
            a = n0 + n1
            parent = src[a:a + n2].long()
            rank = (incl[3] - 1)[parent].long()
            new_params["means"][a:] += samples[:, rank].reshape(-1, 3)
            new_params["scales"][a:] = torch.log(torch.exp(scales[parent]) / 1.6).repeat(2, 1)
            if self.revised_opacity:
                new_o = 1.0 - torch.sqrt(1.0 - torch.sigmoid(opac[parent]))
                new_params["opacities"][a:] = torch.logit(new_o).repeat(2)
        # hand the rebuilt tensors to the parameter dict and the optimizers (as _update_param_with_optimizer)
        for name in list(params.keys()):
            old = params[name]
            new = torch.nn.Parameter(new_params[name], requires_grad=old.requires_grad)
            opt = _opt_of(optimizers, name)
            for g in opt.param_groups:
                for i, p in enumerate(g["params"]):
                    if p is old:
                        stt = opt.state.pop(p, {})
                        for k in list(stt.keys()):
                            if k != "step":
                                stt[k] = new_moments[(name, k)]
                        g["params"][i] = new
                        if stt:
                            opt.state[new] = stt
            params[name] = new
        for k, v in list(state.items()):
            if isinstance(v, Tensor) and v.dim() > 0:
                state[k] = torch.zeros(M, dtype=v.dtype, device=dev)      # (the caller zeroes them anyway)
        return nd, ns, N + nd + ns - M          # duplicated, split, pruned (as _grow_gs / _prune_gs count them)

    def _grow_gs(self, params, optimizers, state, step: int):
        count = state["count"]
        grads = state["grad2d"] / count.clamp_min(1)
        dev = grads.device
        is_grad_high = grads > self.grow_grad2d
        is_small = torch.exp(params["scales"]).max(dim=-1).values <= self.grow_scale3d * state["scene_scale"]
        is_dupli = is_grad_high & is_small
        n_dupli = int(is_dupli.sum().item())
        is_large = ~is_small
        is_split = is_grad_high & is_large
        if step < self.refine_scale2d_stop_iter:
            is_split |= state["radii"] > self.grow_scale2d
        n_split = int(is_split.sum().item())
        if n_dupli > 0:
            duplicate(params, optimizers, state, is_dupli)
        is_split = torch.cat([is_split, torch.zeros(n_dupli, dtype=torch.bool, device=dev)])
        if n_split > 0:
            split(params, optimizers, state, is_split, self.revised_opacity,
                  generator=self._generator(state, dev))
        return n_dupli, n_split

    def _prune_gs(self, params, optimizers, state, step: int) -> int:
        is_prune = torch.sigmoid(params["opacities"].flatten()) < self.prune_opa
        if step > self.reset_every:
            is_too_big = torch.exp(params["scales"]).max(dim=-1).values > self.prune_scale3d * state["scene_scale"]
            if step < self.refine_scale2d_stop_iter:
                is_too_big |= state["radii"] > self.prune_scale2d
            is_prune = is_prune | is_too_big
        n_prune = int(is_prune.sum().item())
        if n_prune > 0:
            remove(params, optimizers, state, is_prune)
        return n_prune


# ---------------------------------------------------------------------------------------
# MCMC strategy ("3D Gaussian Splatting as Markov Chain Monte Carlo"): the reference
# selects it with the "mcmc" preset (/root/reference/gs_init_compare/trainer.py:83-92:
# init_opa 0.5, init_scale 0.1, opacity_reg / scale_reg 0.01) and drives it at
# runner.py:214-215 (`initialize_state()`), 649-658 (`step_post_backward(..., lr=<means lr>)`).
# gsplat's class is third-party and absent: restated from its published behaviour
# (parity unpinned). The two per-Gaussian ops are HIP kernels (csrc/train_ops.hip).
# ---------------------------------------------------------------------------------------
def _call():
    from ._lib import call
    return call


def _multinomial_sample(weights: Tensor, n: int, generator: Optional[torch.Generator] = None) -> Tensor:
    """n draws with replacement, probability proportional to weights (torch.multinomial
    is limited to 2^24 categories: fall back to inverse-CDF sampling above that)."""
    if weights.numel() <= 2 ** 24:
        return torch.multinomial(weights, n, replacement=True, generator=generator)
    cdf = torch.cumsum(weights.double(), 0)
    u = torch.rand(n, device=weights.device, dtype=torch.float64, generator=generator) * cdf[-1]
    return torch.searchsorted(cdf, u).clamp_max(weights.numel() - 1)


@torch.no_grad()
def compute_relocation(opacities: Tensor, scales: Tensor, ratios: Tensor, binoms: Tensor):
    """gsplat.relocation.compute_relocation on the HIP kernel: activated opacities [n],
    scales [n,3], integer ratios [n] -> (new_opacities [n], new_scales [n,3])."""
    n_max = binoms.shape[0]
    opac = opacities.contiguous().float()
    sc = scales.contiguous().float()
    rat = ratios.clamp(1, n_max).to(torch.int32).contiguous()
    bn = binoms.to(device=opac.device, dtype=torch.float32).contiguous()
    new_o, new_s = torch.empty_like(opac), torch.empty_like(sc)
    _call()("gsr_relocation", opac.numel(), opac.data_ptr(), sc.data_ptr(), rat.data_ptr(),
            bn.data_ptr(), n_max, new_o.data_ptr(), new_s.data_ptr(),
            _raw_stream())
    return new_o, new_s


@torch.no_grad()
def inject_noise_to_position(params, optimizers, state, scaler: float,
                             generator: Optional[torch.Generator] = None) -> None:
    means = params["means"]
    noise = torch.randn(means.shape, device=means.device, dtype=means.dtype, generator=generator)
    _call()("gsr_inject_noise", means.shape[0], means.data_ptr(), params["quats"].data_ptr(),
            params["scales"].data_ptr(), params["opacities"].data_ptr(), noise.data_ptr(),
            float(scaler), _raw_stream())


@torch.no_grad()
def relocate(params, optimizers, state, mask: Tensor, binoms: Tensor, min_opacity: float = 0.005,
             generator: Optional[torch.Generator] = None) -> None:
    """Move the dead Gaussians (mask) onto alive ones sampled by opacity."""
    opacities = torch.sigmoid(params["opacities"])
    dead = mask.nonzero(as_tuple=True)[0]
    alive = (~mask).nonzero(as_tuple=True)[0]
    n = len(dead)
    eps = torch.finfo(torch.float32).eps
    sampled = alive[_multinomial_sample(opacities[alive].flatten(), n, generator)]
    new_o, new_s = compute_relocation(
        opacities[sampled], torch.exp(params["scales"])[sampled],
        torch.bincount(sampled, minlength=len(opacities))[sampled] + 1, binoms)
    new_o = torch.clamp(new_o, max=1.0 - eps, min=min_opacity)
    if _gather_ok(params):
        # one gather for the parameters (dead rows take their sampled row), one for the Adam moments
        # (rows kept, zero at the sampled rows), then the new opacity / scale of the sampled rows and of
        # their copies
        N = len(opacities)
        dev = opacities.device
        src = torch.arange(N, dtype=torch.int32, device=dev)
        src[dead] = sampled.int()
        kind = torch.zeros(N, dtype=torch.uint8, device=dev)
        kind_m = kind.clone()
        kind_m[sampled] = 1
        _rebuild_by_gather(params, optimizers, N, src, kind, torch.arange(N, dtype=torch.int32, device=dev), kind_m)
        lo, ls = torch.logit(new_o), torch.log(new_s)
        o, sc = params["opacities"].data, params["scales"].data
        o[sampled] = lo.reshape(o[sampled].shape)
        o[dead] = lo.reshape(o[dead].shape)
        sc[sampled] = ls
        sc[dead] = ls
        for k, v in state.items():
            if isinstance(v, Tensor) and v.dim() > 0 and v.shape[0] == N:
                v[sampled] = 0
        return

    def param_fn(name: str, p: Tensor) -> Tensor:
        p = p.detach().clone()
        if name == "opacities":
            p[sampled] = torch.logit(new_o)
        elif name == "scales":
            p[sampled] = torch.log(new_s)
        p[dead] = p[sampled]
        return p

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        v[sampled] = 0
        return v

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor) and v.dim() > 0 and v.shape[0] == len(opacities):
            v[sampled] = 0


@torch.no_grad()
def sample_add(params, optimizers, state, n: int, binoms: Tensor, min_opacity: float = 0.005,
               generator: Optional[torch.Generator] = None) -> None:
    """Grow by n Gaussians cloned from ones sampled by opacity."""
    opacities = torch.sigmoid(params["opacities"])
    eps = torch.finfo(torch.float32).eps
    sampled = _multinomial_sample(opacities.flatten(), n, generator)
    new_o, new_s = compute_relocation(
        opacities[sampled], torch.exp(params["scales"])[sampled],
        torch.bincount(sampled, minlength=len(opacities))[sampled] + 1, binoms)
    new_o = torch.clamp(new_o, max=1.0 - eps, min=min_opacity)
    if _gather_ok(params):
        # one gather per tensor group: rows [0, N) kept, rows [N, N + n) = the sampled rows (moments zero),
        # then the new opacity / scale of the sampled rows and of their copies
        N = len(opacities)
        dev = opacities.device
        src = torch.cat([torch.arange(N, dtype=torch.int32, device=dev), sampled.int()])
        kind = torch.zeros(N + n, dtype=torch.uint8, device=dev)
        kind[N:] = 1
        _rebuild_by_gather(params, optimizers, N + n, src, kind)
        lo, ls = torch.logit(new_o), torch.log(new_s)
        o, sc = params["opacities"].data, params["scales"].data
        o[sampled] = lo.reshape(o[sampled].shape)
        o[N:] = lo.reshape(o[N:].shape)
        sc[sampled] = ls
        sc[N:] = ls
        for k, v in state.items():
            if isinstance(v, Tensor) and v.dim() > 0 and v.shape[0] == N:
                state[k] = torch.cat([v, torch.zeros((n, *v.shape[1:]), device=v.device, dtype=v.dtype)])
        return

    def param_fn(name: str, p: Tensor) -> Tensor:
        p = p.detach().clone()
        if name == "opacities":
            p[sampled] = torch.logit(new_o)
        elif name == "scales":
            p[sampled] = torch.log(new_s)
        return torch.cat([p, p[sampled]])

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        return torch.cat([v, torch.zeros((len(sampled), *v.shape[1:]), device=v.device, dtype=v.dtype)])

    n_old = len(opacities)
    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor) and v.dim() > 0 and v.shape[0] == n_old:
            state[k] = torch.cat([v, torch.zeros((len(sampled), *v.shape[1:]), device=v.device, dtype=v.dtype)])


@dataclass
class MCMCStrategy:
    cap_max: int = 1_000_000
    noise_lr: float = 5e5
    refine_start_iter: int = 500
    refine_stop_iter: int = 25_000
    refine_every: int = 100
    min_opacity: float = 0.005
    verbose: bool = False
    seed: int = 42                      # shared by all replicas (sampling + noise must agree)

    N_MAX = 51

    def initialize_state(self) -> Dict[str, Any]:
        binoms = torch.zeros((self.N_MAX, self.N_MAX))
        for n in range(self.N_MAX):
            for k in range(n + 1):
                binoms[n, k] = math.comb(n, k)
        return {"binoms": binoms, "generator": None}

    def check_sanity(self, params, optimizers) -> None:
        for key in ("means", "scales", "quats", "opacities"):
            assert key in params, f"{key} is required in params"
        assert set(params.keys()) <= set(optimizers.keys()), "every parameter needs an optimizer"

    def step_pre_backward(self, params, optimizers, state, step: int, info) -> None:   # nothing to retain
        return None

    def _generator(self, state, device) -> torch.Generator:
        if state.get("generator") is None:
            g = torch.Generator(device=device)
            g.manual_seed(self.seed)
            state["generator"] = g
        return state["generator"]

    def refines(self, step: int) -> bool:
        return self.refine_start_iter < step < self.refine_stop_iter and step % self.refine_every == 0

    def mutates_params(self, step: int) -> bool:
        """Position noise is injected on every step, computed from the pre-update parameters
        (runner.py:649-656 before 676-679): not compatible with the plain optimizer-in-backward.
        (Between refine steps `runner.train_step` hands the noise to the fused backward instead --
        `draw_noise` + `FusedAdam.set_step_extras` -- which applies it in the reference's order.)"""
        return True

    def draw_noise(self, params, state) -> Tensor:
        """The standard-normal draws of this step's position noise, from the strategy's generator (the same draw
        `step_post_backward` -> `inject_noise_to_position` makes: on a step without relocation / addition it is the
        only one, so the generator's sequence is the same either way)."""
        means = params["means"]
        return torch.randn(means.shape, device=means.device, dtype=means.dtype, generator=self._generator(state, means.device))

    def step_post_backward(self, params, optimizers, state, step: int, info, lr: float, noise_done: bool = False,
                           **_) -> None:
        dev = params["means"].device
        if state["binoms"].device != dev:
            state["binoms"] = state["binoms"].to(dev)
        binoms, gen = state["binoms"], self._generator(state, dev)
        if self.refines(step):
            assert not noise_done, "the fused noise is for the steps between refinements"
            n_rel = self._relocate_gs(params, optimizers, binoms, gen)
            n_new = self._add_new_gs(params, optimizers, binoms, gen)
            if self.verbose:
                print(f"Step {step}: Relocated {n_rel} GSs. Added {n_new} GSs. "
                      f"Now having {len(params['means'])} GSs.")
        if not noise_done:
            inject_noise_to_position(params, optimizers, {}, scaler=lr * self.noise_lr, generator=gen)

    def _relocate_gs(self, params, optimizers, binoms, gen) -> int:
        dead = torch.sigmoid(params["opacities"].flatten()) <= self.min_opacity
        n = int(dead.sum().item())
        if 0 < n < dead.numel():
            relocate(params, optimizers, {}, dead, binoms, self.min_opacity, gen)
        return n

    def _add_new_gs(self, params, optimizers, binoms, gen) -> int:
        cur = len(params["means"])
        n = max(0, min(self.cap_max, int(1.05 * cur)) - cur)
        if n > 0:
            sample_add(params, optimizers, {}, n, binoms, self.min_opacity, gen)
        return n
