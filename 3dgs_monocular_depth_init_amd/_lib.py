"""ctypes binding of libgsrast.so (the C ABI declared in include/gsrast.h).

The product path has NO fallback: if the library is missing or fails to load,
every op raises. Pointers are passed as integers (`tensor.data_ptr()`), the
stream as `torch.cuda.current_stream().cuda_stream`.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

_LIB_PATH = Path(os.environ.get("GSRAST_LIB") or Path(__file__).resolve().parent / "lib" / "libgsrast.so")
_lock = threading.Lock()
_lib = None

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_i64 = C.c_int64

# name -> argtypes; every function returns int except the three below.
SIGNATURES = {
    "gsr_project_fwd": [_i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _i, _i, _p, _i,
                        _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _i, _p, _p, _p],
    "gsr_project_bwd": [_i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p, _i, _p, _i, _p, _p, _p,
                        _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p],
    "gsr_isect_count": [_i, _i, _p, _p, _i, _i, _p, _p, _p],
    "gsr_isect_scan": [_i, _p, _p, _p, _p],
    "gsr_isect_emit": [_i, _i, _p, _p, _p, _i, _i, _p, _p, _p, _i64, _p],
    "gsr_tile_sort": [_i, _p, _p, _p, _p, _p, _p],
    "gsr_bucket_layout": [_i, _i, _i, _p, _p],
    "gsr_isect_scan_clear": [_i, _p, _p, _p, _p, _p],
    "gsr_bucket_count": [_i, _i, _p, _p, _i, _i, _p, _p, _p, _i, _p, _p],
    "gsr_bucket_emit": [_i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p],
    "gsr_bucket_sort": [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _i, _p],
    "gsr_pair_masks": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p],
    "gsr_pack_records": [_i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p],
    "gsr_rasterize_fwd": [_i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i64, _p],
    "gsr_rasterize_fwd_planar": [_i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i64, _p],
    "gsr_rasterize_fwd_l1": [_i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p],
    "gsr_rasterize_bwd": [_i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "gsr_rasterize_bwd_planar": [_i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p],
}
SIGNATURES["gsr_ssim_workspace_doubles"] = [_i, _i, _i, _i]
SIGNATURES["gsr_ssim_l1_fwd"] = [_i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _f, _p, _p, _p, _p]
SIGNATURES["gsr_ssim_l1_bwd"] = [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p]
SIGNATURES["gsr_l1_fwd"] = [_i64, _p, _p, _p, _p, _p, _p]
SIGNATURES["gsr_l1_bwd"] = [_i64, _p, _p, _p, _f, _p, _p]
SIGNATURES["gsr_debug_tree_reduce8"] = [_p, _p, _p, _p]
SIGNATURES["gsr_inverse4x4"] = [_i, _p, _p, _p, _p, _p]
SIGNATURES["gsr_pack_grad_rows"] = [_i64, _p, _p, _p, _p]
SIGNATURES["gsr_pack_grad_rows_h"] = [_i64, _p, _p, _p, _p]
SIGNATURES["gsr_project_bwd_adam"] = [_i, _i, _p, _p, _p, _i, _i, _f, _i, _p, _p, _i, _p, _p, _i, _i, _p,
                                      _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_double, _p]


class StepExtras(C.Structure):      # gsr_step_extras (include/gsrast.h)
    _fields_ = [("noise", C.c_void_p), ("noise_scale", C.c_double), ("opacity_reg", C.c_double), ("scale_reg", C.c_double),
                ("stat_grad2d", C.c_void_p), ("stat_count", C.c_void_p), ("stat_radii", C.c_void_p),
                ("stat_sx", C.c_double), ("stat_sy", C.c_double), ("stat_inv_max_wh", C.c_double),
                ("stat_use_absgrad", C.c_int)]


SIGNATURES["gsr_project_bwd_adam_ex"] = SIGNATURES["gsr_project_bwd_adam"][:-1] + [C.POINTER(StepExtras), _p]
SIGNATURES["gsr_project_bwd_rows"] = [_i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p, _i, _p, _i, _p,
                                      _p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p]
SIGNATURES["gsr_strategy_accumulate"] = [_i, _i, _p, _i, _p, _f, _f, _p, _p, _p, _f, _p]
SIGNATURES["gsr_relocation"] = [_i, _p, _p, _p, _p, _i, _p, _p, _p]
SIGNATURES["gsr_inject_noise"] = [_i, _p, _p, _p, _p, _p, _f, _p]
SIGNATURES["gsr_sort_pairs_u64"] = [_i64, _p, _p, _p, _p, _p, _i64, _p]
SIGNATURES["gsr_rbf_workspace_bytes"] = [_i]          # returns int64 bytes (restype set in load())
SIGNATURES["gsr_rbf_fit"] = [_i, _p, _p, C.c_double, _i, _p, _i64, _p, _p, _p]
SIGNATURES["gsr_rbf_eval_grid"] = [_i, _p, _p, _p, _i, _i, _i, _p, _p]
SIGNATURES["gsr_bilinear_ac_t"] = [_i, _i, _p, _i, _i, _p, _p]
SIGNATURES["gsr_reset_opacity"] = [_i64, _p, _p, _p, _f, _p]
SIGNATURES["gsr_sparse_adam_step"] = [_i, _i64, _p, _p, _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_double, _p]
SIGNATURES["gsr_refine_decide"] = [_i, _p, _p, _p, _p, _p, _f, _f, _f, _f, _f, _f, _i, _p, _p]
SIGNATURES["gsr_refine_plan"] = [_i, _p, _p, _i, _i, _i, _p, _p, _p]
SIGNATURES["gsr_refine_gather"] = [_i, _i, _p, _p, _p, _p, _p, _p, _p]
SIGNATURES["gsr_adam_step"] = [_i, _p, _p, _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_double, _p]
SIGNATURES.update({
    "gsr_project_sfm": [_i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p],
    "gsr_gather_depth": [_i, _p, _i, _p, _p, _p],
    "gsr_lsq_sums": [_i, _i, _i, _p, _p, _p, _i, _p, _f, _p, _p],
    "gsr_solve_scale_shift": [_i, _p, _p, _p],
    "gsr_ransac_score": [_i, _i, _p, _p, _p, _f, _p, _p, _p, _p],
    "gsr_affine_depth": [_i64, _p, _p, _p, _p],
    "gsr_subsample_mask": [_i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p],
    "gsr_sfm_patch_mask": [_i, _i, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    "gsr_depth_grad": [_i, _i, _p, _p, _p],
    "gsr_tri_interp": [_i, _i, _i, _p, _p, _p, _p, _p],
    "gsr_region_margin_mask": [_i, _i, _i, _p, _p, _p, _p],
    "gsr_unproject_num_blocks": [_i, _i],
    "gsr_unproject_count": [_i, _i, _p, _p, _p, _p, _p, _p],
    "gsr_knn_cell_keys": [_i, _p, _p, _f, _p, _p],
    "gsr_knn_grid_idx": [_i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _f, _i, _p, _p, _p, _p],
    "gsr_lof": [_i, _i, _p, _p, C.c_double, _p, _p, _p, _p],
    "gsr_knn_grid": [_i, _i, _p, _p, _p, _p, _p, _i, _p, _f, _i, _p, _p, _p],
    "gsr_knn_brute": [_i, _i, _i, _p, _p, _p, _p],
    "gsr_m3d_preprocess": [_i, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "gsr_m3d_postprocess": [_i, _i, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _p, _p],
    "gsr_unproject_emit": [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
})
SIGNATURES.update({
    "gsr_dn_gemm": [_i, _i, _i, _p, _i, _p, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _i, _p],
    "gsr_dn_conv_gemm": [_i, _i, _i, _p, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p],
    "gsr_dn_layernorm": [_i, _i, _p, _i, _i, _p, _p, _f, _p, _i, _p, _i, _i, _p],
    "gsr_dn_attention": [_i, _i, _i, _p, _i, _p, _f, _p, _i, _p],
    "gsr_dn_patch_rows": [_i, _i, _i, _i, _p, _p, _p],
    "gsr_dn_im2col": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p],
    "gsr_dn_resize": [_i, _i, _i, _p, _i, _i, _i, _p, _i, _i, _p],
    "gsr_dn_avgpool3s2": [_i, _i, _i, _p, _i, _p, _i, _p],
    "gsr_dn_slice": [_i64, _i, _p, _i, _p, _i, _f, _i, _i, _p],
    "gsr_dn_gru_gate": [_i64, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p],
    "gsr_dn_swiglu": [_i64, _i, _p, _i, _p, _i, _p],
    "gsr_dn_conv3_head": [_i, _i, _i, _p, _i, _i, _p, _i, _p, _p, _i, _p],
    "gsr_dn_conv_gemm2": [_i, _i, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p],
    "gsr_dn_depth_expectation": [_i64, _i, _p, _i, _f, _f, _f, _p, _i, _p],
    "gsr_dn_normal_head": [_i64, _p, _i, _p, _i, _p, _i, _p],
    "gsr_dn_convex_upsample": [_i, _i, _i, _p, _p, _i, _f, _f, _f, _p, _p, _p, _p],
    "gsr_dn_cvt_f32_f16": [_i64, _i, _p, _i, _p, _i, _p],
})
SIGNATURES.update({
    "gsr_pc_min_extents": [_i, _i, _p, _p, _p, _p, _p, _p],
    "gsr_pc_subsample_workspace_bytes": [_i],          # returns int64 bytes (restype set in load())
    "gsr_pc_subsample": [_i, _p, _p, _p, _f, _f, _p, _i64, _p, _p, _p, _p],
})
OPTIONAL_SIGNATURES: dict = {}   # filled by modules that add entry points (init path, train ops)


class GsrastError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load libgsrast.so once; raise GsrastError loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not _LIB_PATH.exists():
            raise GsrastError(
                f"{_LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback."
            )
        lib = C.CDLL(str(_LIB_PATH))
        lib.gsr_last_error.restype = C.c_char_p
        lib.gsr_arch.restype = C.c_char_p
        lib.gsr_version.restype = C.c_int
        for name, args in {**SIGNATURES, **OPTIONAL_SIGNATURES}.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.gsr_pc_subsample_workspace_bytes.restype = C.c_int64
        lib.gsr_rbf_workspace_bytes.restype = C.c_int64
        _lib = lib
        return lib


# Optional per-entry-point device timing (bench.py): when TIMERS is a dict, every
# call is bracketed by events on torch's current stream (the stream the kernels
# are launched on); read with kernel_times_ms() after a synchronize.
TIMERS = None
TIMER_ONLY = None      # optional set of entry-point names to restrict the timing to
TIMER_EVERY = 1        # bracket every k-th call of a timed entry point only: an event record in front of and behind a
                       # kernel leaves the GPU idle for ~6 us each (bench.py's timed region samples every 4th step)
_timer_calls = {}


def call(name: str, *args) -> None:
    lib = load()
    timed = TIMERS is not None and (TIMER_ONLY is None or name in TIMER_ONLY)
    if timed and TIMER_EVERY > 1:
        k = _timer_calls.get(name, 0)
        _timer_calls[name] = k + 1
        timed = k % TIMER_EVERY == 0
    if timed:
        import torch
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        TIMERS.setdefault(name, []).append((e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise GsrastError(f"{name} failed ({rc}): {lib.gsr_last_error().decode()}")


def kernel_times_ms() -> dict:
    """{entry point: (n_calls, mean ms)} from the recorded events."""
    out = {}
    for name, evs in (TIMERS or {}).items():
        ts = [a.elapsed_time(b) for a, b in evs]
        out[name] = (len(ts), sum(ts) / max(len(ts), 1))
    return out


def current_stream() -> int:
    """Raw handle of torch's current HIP stream (the kernels are launched on it). One C call:
    `torch.cuda.current_stream().cuda_stream` builds a Stream object each time (~10 us, and a
    step issues ~20 launches)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def ptr(t) -> int | None:
    """data_ptr of a tensor or None."""
    return None if t is None else t.data_ptr()
