"""CPU oracle for the rasterization hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package never does.

PARITY UNPINNED: the arithmetic restated here lives in the third-party wheel
``gsplat==1.5.2`` (pinned at /root/reference/setup.py:15), whose source is not
under /root/reference and is not installed; the reference holds no tests,
fixtures or golden vectors for this path.  This
file therefore restates gsplat's *published* algorithm (3DGS / EWA splatting,
SURVEY.md Appendix A.1-A.4) anchored on the reference's own call site
``gs_init_compare/runner.py:341-362`` (kwargs), ``runner.py:479-482`` (depth
channel), ``runner.py:493-495`` (alphas) and ``runner.py:497-503,639-647``
(what the densification strategy reads from ``info``).  Gradients are pinned by
``torch.autograd`` over this restatement (and gradcheck in fp64 in the tests).

Every convention is a named constant so that a later comparison against a real
gsplat 1.5.2 build can flip it in one place.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

# ---- named conventions -------------------------------
ALPHA_THRESHOLD = 1.0 / 255.0   # A.4: skip contributions below this alpha
ALPHA_MAX = 0.999               # A.4: alpha clamp
T_THRESHOLD = 1e-4              # A.4: stop a pixel when next T <= this
EXTENT_MAX = 3.33               # A.1: extent cap in sigmas (gsplat 1.5.x)
FRUSTUM_GUARD = 0.3             # A.1: +30% tan(fov/2) clamp on the Jacobian
TILE_SIZE = 16                  # A.3
SH_C0 = 0.2820947917738781      # runner_utils.py:150 uses 0.28209479177387814


# ---- A.1 projection ---------------------------------------------------------
def quat_to_rotmat(quats: Tensor) -> Tensor:
    """wxyz quaternion (normalised here) -> rotation matrix [...,3,3]."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(quats.shape[:-1] + (3, 3))


def quat_scale_to_covar(quats: Tensor, scales: Tensor) -> Tensor:
    R = quat_to_rotmat(quats)
    M = R * scales[..., None, :]
    return M @ M.transpose(-1, -2)


def project_gaussians(
    means: Tensor,        # [N,3]
    covars: Tensor,       # [N,3,3]
    viewmats: Tensor,     # [C,4,4] world->camera
    Ks: Tensor,           # [C,3,3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    opacities: Optional[Tensor] = None,   # [N]
    comp_scales_opacity: bool = False,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Pinhole EWA projection.  Returns radii [C,N,2] int32, means2d [C,N,2],
    depths [C,N], conics [C,N,3], compensations [C,N].  Culled pairs have
    radii == 0 (their other outputs are unspecified but finite)."""
    R = viewmats[:, :3, :3]                      # [C,3,3]
    t = viewmats[:, :3, 3]                       # [C,3]
    pc = torch.einsum("cij,nj->cni", R, means) + t[:, None, :]   # [C,N,3]
    cov_c = torch.einsum("cij,njk,clk->cnil", R, covars, R)      # [C,N,3,3]
    x, y, z = pc.unbind(-1)
    fx, fy = Ks[:, 0, 0][:, None], Ks[:, 1, 1][:, None]
    cx, cy = Ks[:, 0, 2][:, None], Ks[:, 1, 2][:, None]
    tanx, tany = 0.5 * width / fx, 0.5 * height / fy
    lim_xp = (width - cx) / fx + FRUSTUM_GUARD * tanx
    lim_xn = cx / fx + FRUSTUM_GUARD * tanx
    lim_yp = (height - cy) / fy + FRUSTUM_GUARD * tany
    lim_yn = cy / fy + FRUSTUM_GUARD * tany
    zs = torch.where(z.abs() < 1e-12, torch.full_like(z, 1e-12), z)  # culled anyway
    rz = 1.0 / zs
    rz2 = rz * rz
    tx = zs * torch.minimum(lim_xp, torch.maximum(-lim_xn, x * rz))
    ty = zs * torch.minimum(lim_yp, torch.maximum(-lim_yn, y * rz))
    zero = torch.zeros_like(z)
    J = torch.stack(
        [fx * rz, zero, -fx * tx * rz2, zero, fy * rz, -fy * ty * rz2], dim=-1
    ).reshape(z.shape + (2, 3))
    cov2d = J @ cov_c @ J.transpose(-1, -2)      # [C,N,2,2]
    means2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)
    c00, c01, c11 = cov2d[..., 0, 0], cov2d[..., 0, 1], cov2d[..., 1, 1]
    det_orig = c00 * c11 - c01 * c01
    b00, b11 = c00 + eps2d, c11 + eps2d
    det = b00 * b11 - c01 * c01
    valid = (z >= near_plane) & (z <= far_plane) & (det > 0)
    det_safe = torch.where(det > 0, det, torch.ones_like(det))
    compensations = torch.sqrt(torch.clamp(det_orig / det_safe, min=0.0))
    conics = torch.stack([b11 / det_safe, -c01 / det_safe, b00 / det_safe], dim=-1)
    extent = torch.full_like(z, EXTENT_MAX)
    if opacities is not None:
        op = opacities[None, :].expand_as(z)
        if comp_scales_opacity:
            op = op * compensations
        valid = valid & (op >= ALPHA_THRESHOLD)
        op_safe = torch.clamp(op, min=ALPHA_THRESHOLD)
        extent = torch.minimum(extent, torch.sqrt(2.0 * torch.log(op_safe / ALPHA_THRESHOLD)))
    with torch.no_grad():
        rx = torch.ceil(extent * torch.sqrt(torch.clamp(b00, min=0)))
        ry = torch.ceil(extent * torch.sqrt(torch.clamp(b11, min=0)))
        valid = valid & ~((rx <= radius_clip) & (ry <= radius_clip))
        mx, my = means2d[..., 0], means2d[..., 1]
        valid = valid & ~(
            (mx + rx <= 0) | (mx - rx >= width) | (my + ry <= 0) | (my - ry >= height)
        )
        radii = torch.stack([rx, ry], dim=-1)
        radii = torch.where(valid[..., None], radii, torch.zeros_like(radii)).to(torch.int32)
    return radii, means2d, z, conics, compensations


# ---- A.2 spherical harmonics ------------------------------------------------
def eval_sh(degree: int, dirs: Tensor, coeffs: Tensor) -> Tensor:
    """dirs [...,3] (normalised here), coeffs [...,K,3] -> [...,3] (no +0.5)."""
    d = dirs / dirs.norm(dim=-1, keepdim=True).clamp_min(1e-20)
    x, y, z = d.unbind(-1)
    c = lambda k: coeffs[..., k, :]
    res = 0.2820947917738781 * c(0)
    if degree >= 1:
        res = res + 0.48860251190292 * (-y[..., None] * c(1) + z[..., None] * c(2) - x[..., None] * c(3))
    if degree >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2 * x * y
        res = (
            res
            + (0.5462742152960395 * fS1)[..., None] * c(4)
            + (fTmp0B * y)[..., None] * c(5)
            + (0.9461746957575601 * z2 - 0.3153915652525201)[..., None] * c(6)
            + (fTmp0B * x)[..., None] * c(7)
            + (0.5462742152960395 * fC1)[..., None] * c(8)
        )
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        res = (
            res
            + (-0.5900435899266435 * fS2)[..., None] * c(9)
            + (fTmp1B * fS1)[..., None] * c(10)
            + (fTmp0C * y)[..., None] * c(11)
            + (z * (1.865881662950577 * z2 - 1.119528997770346))[..., None] * c(12)
            + (fTmp0C * x)[..., None] * c(13)
            + (fTmp1B * fC1)[..., None] * c(14)
            + (-0.5900435899266435 * fC2)[..., None] * c(15)
        )
    return res


# ---- A.3 tiles & sort -------------------------------------------------------
@torch.no_grad()
def isect_tiles(
    means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_w: int, tile_h: int
) -> Tuple[Tensor, Tensor, Tensor]:
    """Returns tiles_per_gauss [C,N] int32, isect_ids [I] int64 (sorted keys:
    cam | tile | depth bits), flatten_ids [I] int32 (= c*N+i)."""
    C, N = depths.shape
    ts = float(tile_size)
    mx, my = means2d[..., 0] / ts, means2d[..., 1] / ts
    rx, ry = radii[..., 0].float() / ts, radii[..., 1].float() / ts
    vis = (radii > 0).all(-1)
    x0 = torch.clamp(torch.floor(mx - rx), 0, tile_w).to(torch.int64)
    x1 = torch.clamp(torch.ceil(mx + rx), 0, tile_w).to(torch.int64)
    y0 = torch.clamp(torch.floor(my - ry), 0, tile_h).to(torch.int64)
    y1 = torch.clamp(torch.ceil(my + ry), 0, tile_h).to(torch.int64)
    nx = torch.where(vis, x1 - x0, torch.zeros_like(x0))
    ny = torch.where(vis, y1 - y0, torch.zeros_like(y0))
    tpg = (nx * ny).to(torch.int32)
    n_tiles = tile_w * tile_h
    tile_bits = max(1, math.ceil(math.log2(max(n_tiles, 2))))
    flat = torch.nonzero(tpg.reshape(-1) > 0).reshape(-1)
    keys, vals = [], []
    depth_bits = depths.float().contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    for g in flat.tolist():
        c, i = divmod(g, N)
        ys = torch.arange(int(y0[c, i]), int(y1[c, i]), dtype=torch.int64)
        xs = torch.arange(int(x0[c, i]), int(x1[c, i]), dtype=torch.int64)
        tid = (ys[:, None] * tile_w + xs[None, :]).reshape(-1)
        k = (c << (32 + tile_bits)) | (tid << 32) | int(depth_bits[c, i])
        keys.append(k)
        vals.append(torch.full_like(tid, g))
    if keys:
        keys_t = torch.cat(keys)
        vals_t = torch.cat(vals)
        order = torch.argsort(keys_t, stable=True)
        return tpg, keys_t[order], vals_t[order].to(torch.int32)
    return tpg, torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32)


@torch.no_grad()
def isect_tiles_fast(
    means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_w: int, tile_h: int
) -> Tuple[Tensor, Tensor, Tensor]:
    """Vectorised equivalent of isect_tiles (same outputs) for larger scenes."""
    C, N = depths.shape
    ts = float(tile_size)
    mx, my = means2d[..., 0] / ts, means2d[..., 1] / ts
    rx, ry = radii[..., 0].float() / ts, radii[..., 1].float() / ts
    vis = (radii > 0).all(-1)
    x0 = torch.clamp(torch.floor(mx - rx), 0, tile_w).to(torch.int64)
    x1 = torch.clamp(torch.ceil(mx + rx), 0, tile_w).to(torch.int64)
    y0 = torch.clamp(torch.floor(my - ry), 0, tile_h).to(torch.int64)
    y1 = torch.clamp(torch.ceil(my + ry), 0, tile_h).to(torch.int64)
    nx = torch.where(vis, x1 - x0, torch.zeros_like(x0)).reshape(-1)
    ny = torch.where(vis, y1 - y0, torch.zeros_like(y0)).reshape(-1)
    cnt = nx * ny
    tpg = cnt.to(torch.int32).reshape(C, N)
    n_tiles = tile_w * tile_h
    tile_bits = max(1, math.ceil(math.log2(max(n_tiles, 2))))
    total = int(cnt.sum())
    if total == 0:
        return tpg, torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32)
    g = torch.repeat_interleave(torch.arange(C * N, dtype=torch.int64), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(total, dtype=torch.int64) - start[g]
    nxg = nx[g].clamp_min(1)
    ty = y0.reshape(-1)[g] + local // nxg
    tx = x0.reshape(-1)[g] + local % nxg
    tid = ty * tile_w + tx
    cam = g // N
    depth_bits = depths.float().contiguous().view(torch.int32).to(torch.int64).reshape(-1) & 0xFFFFFFFF
    keys = (cam << (32 + tile_bits)) | (tid << 32) | depth_bits[g]
    order = torch.argsort(keys, stable=True)
    return tpg, keys[order], g[order].to(torch.int32)


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, C: int, tile_w: int, tile_h: int) -> Tensor:
    """offsets[c,ty,tx] = index of the first intersection of that tile."""
    n_tiles = tile_w * tile_h
    tile_bits = max(1, math.ceil(math.log2(max(n_tiles, 2))))
    tile_key = isect_ids >> 32                       # cam << tile_bits | tile
    cam = tile_key >> tile_bits
    tid = tile_key & ((1 << tile_bits) - 1)
    flat_tile = cam * n_tiles + tid
    counts = torch.bincount(flat_tile, minlength=C * n_tiles)
    offsets = torch.cumsum(counts, 0) - counts
    return offsets.to(torch.int32).reshape(C, tile_h, tile_w)


# ---- A.4 compositing --------------------------------------------------------
def rasterize_to_pixels(
    means2d: Tensor,       # [C,N,2]
    conics: Tensor,        # [C,N,3]
    colors: Tensor,        # [C,N,D]
    opacities: Tensor,     # [C,N]
    width: int,
    height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [C,th,tw]
    flatten_ids: Tensor,    # [I]
    backgrounds: Optional[Tensor] = None,   # [C,D]
) -> Tuple[Tensor, Tensor]:
    """Differentiable (autograd) front-to-back compositing, one tile at a time.
    Returns render_colors [C,H,W,D], render_alphas [C,H,W,1]."""
    C, N = means2d.shape[:2]
    D = colors.shape[-1]
    th, tw = isect_offsets.shape[1:]
    n_isects = flatten_ids.numel()
    offs = isect_offsets.reshape(-1).tolist() + [n_isects]
    m2 = means2d.reshape(C * N, 2)
    cn = conics.reshape(C * N, 3)
    cl = colors.reshape(C * N, D)
    op = opacities.reshape(C * N)
    dt = means2d.dtype
    rows_out = []
    alpha_rows = []
    for c in range(C):
        tile_rows_c, tile_rows_a = [], []
        for ty in range(th):
            row_c, row_a = [], []
            for tx in range(tw):
                t = (c * th + ty) * tw + tx
                s, e = offs[t], offs[t + 1]
                y0, x0 = ty * tile_size, tx * tile_size
                hh = min(tile_size, height - y0)
                ww = min(tile_size, width - x0)
                if e <= s:
                    col = torch.zeros(hh, ww, D, dtype=dt)
                    T_fin = torch.ones(hh, ww, dtype=dt)
                else:
                    ids = flatten_ids[s:e].long()
                    py = (torch.arange(hh, dtype=dt) + (y0 + 0.5))[:, None].expand(hh, ww).reshape(-1)
                    px = (torch.arange(ww, dtype=dt) + (x0 + 0.5))[None, :].expand(hh, ww).reshape(-1)
                    mu = m2[ids]
                    co = cn[ids]
                    dx = mu[None, :, 0] - px[:, None]          # [P,L]
                    dy = mu[None, :, 1] - py[:, None]
                    sigma = 0.5 * (co[None, :, 0] * dx * dx + co[None, :, 2] * dy * dy) + co[None, :, 1] * dx * dy
                    alpha = torch.clamp(op[ids][None, :] * torch.exp(-sigma), max=ALPHA_MAX)
                    valid = (sigma >= 0) & (alpha >= ALPHA_THRESHOLD)
                    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
                    one_m = 1.0 - a_eff
                    T_incl = torch.cumprod(one_m, dim=1)                     # T after j
                    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
                    stop = valid & (T_incl <= T_THRESHOLD)
                    stopped = torch.cumsum(stop.to(torch.int32), dim=1) > 0  # j at/after first stop
                    live = ~stopped
                    w = torch.where(live, a_eff * T_excl, torch.zeros_like(a_eff))
                    col = (w @ cl[ids]).reshape(hh, ww, D)
                    # final transmittance: product over live entries only
                    T_fin = torch.prod(torch.where(live, one_m, torch.ones_like(one_m)), dim=1).reshape(hh, ww)
                if backgrounds is not None:
                    col = col + T_fin[..., None] * backgrounds[c][None, None, :]
                row_c.append(col)
                row_a.append(1.0 - T_fin)
            tile_rows_c.append(torch.cat(row_c, dim=1))
            tile_rows_a.append(torch.cat(row_a, dim=1))
        rows_out.append(torch.cat(tile_rows_c, dim=0))
        alpha_rows.append(torch.cat(tile_rows_a, dim=0))
    render_colors = torch.stack(rows_out, 0)
    render_alphas = torch.stack(alpha_rows, 0)[..., None]
    return render_colors, render_alphas


# ---- the boundary -----------------------------------------------------------
def rasterization(
    means: Tensor,
    quats: Tensor,
    scales: Tensor,
    opacities: Tensor,
    colors: Tensor,
    viewmats: Tensor,
    Ks: Tensor,
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = False,
    tile_size: int = TILE_SIZE,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    camera_model: str = "pinhole",
    covars: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Restatement of ``gsplat.rendering.rasterization`` for the kwargs used at
    gs_init_compare/runner.py:341-362 (non-packed, pinhole, single process)."""
    assert camera_model == "pinhole" and not distributed and covars is None
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED")
    C, N = viewmats.shape[0], means.shape[0]
    covs = quat_scale_to_covar(quats, scales)
    antialiased = rasterize_mode == "antialiased"
    radii, means2d, depths, conics, comps = project_gaussians(
        means, covs, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
        opacities, comp_scales_opacity=antialiased,
    )
    opac = opacities[None, :].expand(C, N)
    if antialiased:
        opac = opac * comps
    vis = (radii > 0).all(-1)
    if sh_degree is not None:
        campos = torch.linalg.inv(viewmats)[:, :3, 3]
        dirs = means[None, :, :] - campos[:, None, :]
        shs = colors[None].expand(C, -1, -1, -1) if colors.dim() == 3 else colors
        rgb = eval_sh(sh_degree, dirs, shs)
        rgb = torch.where(vis[..., None], rgb, torch.zeros_like(rgb))
        rgb = torch.clamp_min(rgb + 0.5, 0.0)
    else:
        rgb = colors[None].expand(C, -1, -1) if colors.dim() == 2 else colors
    if render_mode in ("RGB+D", "RGB+ED"):
        feats = torch.cat([rgb, depths[..., None]], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros_like(backgrounds[:, :1])], dim=-1)
    elif render_mode in ("D", "ED"):
        feats = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros_like(backgrounds[:, :1])
    else:
        feats = rgb
    tile_w = math.ceil(width / tile_size)
    tile_h = math.ceil(height / tile_size)
    tpg, isect_ids, flatten_ids = isect_tiles_fast(
        means2d.detach(), radii, depths.detach(), tile_size, tile_w, tile_h
    )
    isect_offsets = isect_offset_encode(isect_ids, C, tile_w, tile_h)
    render_colors, render_alphas = rasterize_to_pixels(
        means2d, conics, feats, opac, width, height, tile_size, isect_offsets, flatten_ids, backgrounds
    )
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat(
            [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1
        )
    meta = {
        "radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
        "opacities": opac, "tile_width": tile_w, "tile_height": tile_h,
        "tiles_per_gauss": tpg, "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets, "width": width, "height": height,
        "tile_size": tile_size, "n_cameras": C, "colors": feats,
    }
    return render_colors, render_alphas, meta
