"""GPU parity: HIP hot path (through the C ABI) vs the CPU oracle.

Tolerances (BASELINE.json north_star): rendered RGB / depth / alpha within
1e-4 absolute of the oracle, parameter gradients within 1e-3 relative
(max |a-b| / max |b| per tensor). Integer outputs (radii, tile lists) are
compared exactly where the fp32 inputs they derive from are identical.
"""
import importlib
import math

import pytest
import torch

from oracle import rasterization_oracle as O
from tests import parity_log, scenes

pytestmark = pytest.mark.gpu

IMG_ATOL = 1e-4
FLIP_ATOL = 5e-3          # one threshold-boundary contribution (see _check)
GRAD_RTOL = 1e-3
PSNR_ATOL_DB = 1e-4       # north_star: "within 1e-4 PSNR" of the reference rasterizer


@pytest.fixture(scope="module")
def R():
    mod = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
    importlib.import_module("3dgs_monocular_depth_init_amd._lib").load()
    return mod


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def _run_both(R, sc, vm, K, W, H, *, sh_degree=3, render_mode="RGB", backgrounds=None,
              rasterize_mode="classic", absgrad=False, split=False, seed=11, tight=False):
    """Run oracle (CPU) and HIP on identical inputs with a random cotangent."""
    names = ["means", "quats", "scales", "opacities", "sh0", "shN"]
    cpu = {k: sc[k].clone().requires_grad_(True) for k in names}
    gpu = {k: sc[k].clone().cuda().requires_grad_(True) for k in names}
    bg_c = backgrounds
    bg_g = backgrounds.cuda() if backgrounds is not None else None
    col_c = torch.cat([cpu["sh0"], cpu["shN"]], 1)
    rc_c, ra_c, meta_c = O.rasterization(
        cpu["means"], cpu["quats"], cpu["scales"], cpu["opacities"], col_c, vm, K, W, H,
        sh_degree=sh_degree, render_mode=render_mode, backgrounds=bg_c,
        rasterize_mode=rasterize_mode)
    col_g = (gpu["sh0"], gpu["shN"]) if split else torch.cat([gpu["sh0"], gpu["shN"]], 1)
    rc_g, ra_g, meta_g = R.rasterization(
        gpu["means"], gpu["quats"], gpu["scales"], gpu["opacities"], col_g, vm.cuda(), K.cuda(),
        W, H, sh_degree=sh_degree, render_mode=render_mode, backgrounds=bg_g, packed=False,
        rasterize_mode=rasterize_mode, absgrad=absgrad, _tight_tiles=tight)
    meta_g["means2d"].retain_grad()
    g = torch.Generator().manual_seed(seed)
    w_c = torch.randn(rc_c.shape, generator=g)
    w_a = torch.randn(ra_c.shape, generator=g)
    ((rc_c * w_c).sum() + (ra_c * w_a).sum()).backward()
    ((rc_g * w_c.cuda()).sum() + (ra_g * w_a.cuda()).sum()).backward()
    torch.cuda.synchronize()
    return cpu, gpu, (rc_c, ra_c, meta_c), (rc_g, ra_g, meta_g)


def _check(cpu, gpu, out_c, out_g, img_atol=IMG_ATOL, grad_rtol=GRAD_RTOL, mean_frac=1e-2,
           flip_frac=1e-4):
    rc_c, ra_c, meta_c = out_c
    rc_g, ra_g, meta_g = out_g
    vis_c = (meta_c["radii"] > 0).all(-1)
    vis_g = (meta_g["radii"].cpu() > 0).all(-1)
    # culling decisions may differ only for pairs sitting on a float boundary
    assert (vis_c != vis_g).sum().item() <= max(2, vis_c.numel() // 100000)
    both = vis_c & vis_g
    assert torch.equal(meta_c["radii"][both], meta_g["radii"].cpu()[both])
    assert _rel(meta_g["means2d"].detach().cpu()[both], meta_c["means2d"].detach()[both]) < 1e-5
    assert _rel(meta_g["conics"].detach().cpu()[both], meta_c["conics"].detach()[both]) < 1e-3
    for name, got, ref in (("render_colors", rc_g, rc_c), ("render_alphas", ra_g, ra_c)):
        got_c, ref_d = got.detach().cpu(), ref.detach()
        err = (got_c - ref_d).abs()
        # A pixel-Gaussian pair whose alpha (or next-T) sits within one ulp of a
        # threshold (1/255, 0.999, 1e-4) may be blended by one implementation and
        # skipped by the other (exp rounding): that moves ONE pixel by at most
        # ~alpha*colour = 4e-3. Such flips are allowed on <= 1e-4 of the pixels;
        # everything else must be within img_atol, and the mean error far below it.
        n_bad = int((err > img_atol).sum())
        flip = FLIP_ATOL * max(1.0, float(ref_d.abs().max()))
        # north_star's own criterion: the PSNR of the render against a fixed target image
        # (the training loss's view of the render) moves by <= 1e-4 dB between HIP and oracle
        target = torch.rand(ref_d.shape, generator=torch.Generator().manual_seed(2)) * float(
            ref_d.abs().max().clamp_min(1e-6))
        peak = float(target.max())
        d_psnr = abs(parity_log.psnr(got_c, target, peak) - parity_log.psnr(ref_d, target, peak))
        parity_log.record("image", output=name, pixels=err.numel(), n_over_atol=n_bad,
                          frac_over_atol=n_bad / err.numel(), max_abs_err=float(err.max()),
                          mean_abs_err=float(err.mean()), psnr_hip_vs_oracle_db=parity_log.psnr(
                              got_c, ref_d, max(1.0, float(ref_d.abs().max()))),
                          delta_psnr_vs_target_db=d_psnr, atol=img_atol, flip_frac_allowed=flip_frac)
        assert n_bad <= max(1, math.ceil(flip_frac * err.numel())), f"{name}: {n_bad} px > {img_atol}"
        assert err.max().item() <= flip, f"{name} max abs err {err.max().item():.3e}"
        assert err.mean().item() <= img_atol * mean_frac, f"{name} mean abs err {err.mean().item():.3e}"
        assert d_psnr <= PSNR_ATOL_DB, f"{name}: PSNR vs target differs by {d_psnr:.2e} dB"
    for k in cpu:
        if cpu[k].grad is None:
            assert gpu[k].grad is None or gpu[k].grad.abs().max().item() == 0.0
            continue
        got, ref = gpu[k].grad.cpu(), cpu[k].grad
        # L2-relative error of the whole tensor, plus the element-wise bound
        # (|a-b| <= rtol * max|b|) on all but the few elements a threshold flip
        # (see above) can touch: a flipped pair adds/removes one O(alpha)
        # contribution to ONE Gaussian's gradients.
        l2 = float((got - ref).norm() / ref.norm().clamp_min(1e-20))
        tol = grad_rtol * float(ref.abs().max())
        n_bad = int(((got - ref).abs() > tol).sum())
        parity_log.record("grad", tensor=k, elements=ref.numel(), l2_rel=l2, max_rel=_rel(got, ref),
                          n_over_rtol=n_bad, frac_over_rtol=n_bad / ref.numel(), rtol=grad_rtol,
                          flip_frac_allowed=flip_frac)
        assert l2 <= grad_rtol, f"grad {k}: L2 rel err {l2:.3e}"
        assert n_bad <= max(1, math.ceil(flip_frac * ref.numel())), f"grad {k}: {n_bad} elements off by > {tol:.2e}"
        assert _rel(got, ref) <= 50 * grad_rtol, f"grad {k}: max rel err {_rel(got, ref):.3e}"


def _tiny(N=600, seed=5, W=70, H=50):
    sc = scenes.make_scene(N, seed, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    vm = torch.eye(4)[None].clone()
    vm[0, 2, 3] = 2.5
    K = torch.tensor([[[80.0, 0, W / 2], [0, 80.0, H / 2], [0, 0, 1]]])
    return sc, vm, K, W, H


def test_tile_lists_match_oracle(R):
    sc, vm, K, W, H = _tiny()
    cpu, gpu, out_c, out_g = _run_both(R, sc, vm, K, W, H)
    mc, mg = out_c[2], out_g[2]
    # same fp32 inputs -> recompute the oracle's lists from the HIP projection outputs
    tw, th = mg["tile_width"], mg["tile_height"]
    tpg, ids, flat = O.isect_tiles_fast(mg["means2d"].detach().cpu(), mg["radii"].cpu(),
                                        mg["depths"].detach().cpu(), 16, tw, th)
    offs = O.isect_offset_encode(ids, 1, tw, th)
    assert torch.equal(mg["flatten_ids"].cpu(), flat)
    assert torch.equal(mg["isect_offsets"].cpu(), offs)


@pytest.mark.parametrize("sh_degree", [0, 1, 2, 3])
def test_tiny_sh_degrees(R, sh_degree):
    sc, vm, K, W, H = _tiny()
    _check(*_run_both(R, sc, vm, K, W, H, sh_degree=sh_degree))


@pytest.mark.parametrize("tight", [False, True])
def test_equal_depths_keep_index_order(R, tight):
    """Every Gaussian on one plane facing the camera: all depths are the SAME float, so every tile's list is one
    long tie that the sort key's index bits must order (gsplat: stable sort by (tile, depth), values in index
    order). With the depth-bin split of round 4 such a bucket has ONE bin per tile, far longer than the in-bin
    rank handles, and takes the bitonic networks kept for it: lists bit-exact against the oracle's stable sort,
    image and gradients as usual."""
    N = 4000
    g = torch.Generator().manual_seed(31)
    sc = scenes.make_scene(N, 31, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    sc["means"][:, 2] = 0.25                      # one plane; the camera below looks along +z from z = -2.5
    vm = torch.eye(4)[None].clone()
    vm[0, 2, 3] = 2.5
    W, H = 112, 80
    K = torch.tensor([[[80.0, 0, W / 2], [0, 80.0, H / 2], [0, 0, 1]]])
    cpu, gpu, out_c, out_g = _run_both(R, sc, vm, K, W, H, tight=tight)
    mg = out_g[2]
    d = mg["depths"].detach().cpu()[0]
    vis = (mg["radii"].cpu()[0] > 0).all(-1)
    assert float(d[vis].min()) == float(d[vis].max())                 # really one depth
    tw, th = mg["tile_width"], mg["tile_height"]
    counts = torch.diff(torch.cat([mg["isect_offsets"].reshape(-1).cpu(),
                                   torch.tensor([mg["flatten_ids"].numel()], dtype=torch.int32)]))
    assert int(counts.max()) > 200                                     # longer than a depth bin may be (160)
    if not tight:
        tpg, ids, flat = O.isect_tiles_fast(mg["means2d"].detach().cpu(), mg["radii"].cpu(), mg["depths"].detach().cpu(), 16, tw, th)
        assert torch.equal(mg["flatten_ids"].cpu(), flat)
        assert torch.equal(mg["isect_offsets"].cpu(), O.isect_offset_encode(ids, 1, tw, th))
    # inside every tile the list is in index order (the tie-break), tight or not
    off = mg["isect_offsets"].reshape(-1).long().cpu()
    fl = mg["flatten_ids"].long().cpu()
    tile_of = torch.bucketize(torch.arange(fl.numel()), off, right=True)
    same = tile_of[1:] == tile_of[:-1]
    assert (fl[1:][same] > fl[:-1][same]).all()
    _check(cpu, gpu, out_c, out_g)


def test_tiny_split_sh_layout(R):
    sc, vm, K, W, H = _tiny()
    _check(*_run_both(R, sc, vm, K, W, H, split=True))


@pytest.mark.parametrize("N,sh_degree", [(1001, 3), (1002, 3), (1003, 3), (1001, 2)])
def test_multi_camera_rows_not_16_byte_aligned(R, N, sh_degree):
    """Several cameras and N % 4 != 0: the projection forward's waves are formed over the flat
    (camera, Gaussian) index, so for cameras >= 1 a wave's first shN row is 64 m - c N -- not a multiple
    of 4 rows, i.e. its 11.5 KB block does not start on a 16-byte boundary and must NOT go through the
    16-byte LDS-DMA slab prefetch of degree 3 (such waves take the band-wise dword reads).
    Every earlier multi-camera test had N % 4 == 0. Both SH layouts, against the oracle."""
    sc = scenes.make_scene(N, 9, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    W, H = 96, 64
    vm, K = scenes.cameras([0, 40, 77], width=W, height=H, f=90.0, dist=2.5)
    for split in (True, False):
        _check(*_run_both(R, sc, vm, K, W, H, sh_degree=sh_degree, split=split))


def test_tiny_rgb_ed_with_background(R):
    sc, vm, K, W, H = _tiny()
    bg = torch.tensor([[0.2, 0.5, 0.9]])
    _check(*_run_both(R, sc, vm, K, W, H, render_mode="RGB+ED", backgrounds=bg), img_atol=5e-4)


@pytest.mark.parametrize("render_mode", ["D", "ED", "RGB+D"])
def test_tiny_depth_render_modes(R, render_mode):
    """The depth-carrying render modes of gsplat's rasterization(): accumulated depth (D), expected
    depth (ED = D / alpha) alone and beside the colours; gradients flow to the means through the depth."""
    sc, vm, K, W, H = _tiny()
    _check(*_run_both(R, sc, vm, K, W, H, render_mode=render_mode), img_atol=5e-4)


def test_tiny_precomputed_colors_and_two_camera_antialiased(R):
    """colors given per Gaussian ([N,3], sh_degree=None) instead of SH coefficients, two cameras,
    antialiased compensation: the non-SH input path (`gsr_pack_records`) against the oracle."""
    sc = scenes.make_scene(700, 12, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    vm, K = scenes.cameras([3, 40], width=80, height=56, f=85.0, dist=2.5)
    col = torch.rand(700, 3, generator=torch.Generator().manual_seed(4))
    names = ["means", "quats", "scales", "opacities"]
    cpu = {k: sc[k].clone().requires_grad_(True) for k in names}
    gpu = {k: sc[k].clone().cuda().requires_grad_(True) for k in names}
    cpu["colors"], gpu["colors"] = col.clone().requires_grad_(True), col.clone().cuda().requires_grad_(True)
    rc_c, ra_c, meta_c = O.rasterization(cpu["means"], cpu["quats"], cpu["scales"], cpu["opacities"], cpu["colors"],
                                         vm, K, 80, 56, sh_degree=None, rasterize_mode="antialiased")
    rc_g, ra_g, meta_g = R.rasterization(gpu["means"], gpu["quats"], gpu["scales"], gpu["opacities"], gpu["colors"],
                                         vm.cuda(), K.cuda(), 80, 56, sh_degree=None, packed=False,
                                         rasterize_mode="antialiased")
    g = torch.Generator().manual_seed(21)
    w_c, w_a = torch.randn(rc_c.shape, generator=g), torch.randn(ra_c.shape, generator=g)
    ((rc_c * w_c).sum() + (ra_c * w_a).sum()).backward()
    ((rc_g * w_c.cuda()).sum() + (ra_g * w_a.cuda()).sum()).backward()
    torch.cuda.synchronize()
    _check(cpu, gpu, (rc_c, ra_c, meta_c), (rc_g, ra_g, meta_g))


def test_tiny_antialiased(R):
    sc, vm, K, W, H = _tiny()
    _check(*_run_both(R, sc, vm, K, W, H, rasterize_mode="antialiased"))


def test_two_cameras(R):
    sc = scenes.make_scene(800, 9, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    vm, K = scenes.cameras([0, 30], width=96, height=64, f=90.0, dist=2.5)
    _check(*_run_both(R, sc, vm, K, 96, 64))


def test_absgrad_and_means2d_grad(R):
    sc, vm, K, W, H = _tiny()
    cpu, gpu, out_c, out_g = _run_both(R, sc, vm, K, W, H, absgrad=True)
    m2d = out_g[2]["means2d"]
    assert m2d.grad is not None and m2d.grad.shape == m2d.shape
    assert hasattr(m2d, "absgrad") and (m2d.absgrad >= m2d.grad.abs() - 1e-6).all()
    _check(cpu, gpu, out_c, out_g)


def test_config_c1(R):
    """BASELINE config c1: 10k Gaussians, 1 camera, 256x256."""
    sc, vm, K, W, H = scenes.config_c1()
    _check(*_run_both(R, sc, vm, K, W, H))


def test_empty_and_offscreen(R):
    sc, vm, K, W, H = _tiny(N=50)
    sc["means"][:, 2] = -10.0            # everything behind the camera
    names = ["means", "quats", "scales", "opacities"]
    g = {k: sc[k].cuda() for k in names}
    col = torch.cat([sc["sh0"], sc["shN"]], 1).cuda()
    rc, ra, meta = R.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], col,
                                   vm.cuda(), K.cuda(), W, H, sh_degree=3, packed=False)
    assert rc.abs().max().item() == 0 and ra.abs().max().item() == 0
    assert meta["flatten_ids"].numel() == 0 and (meta["radii"] == 0).all()


def test_long_tile_lists(R):
    """All Gaussians in one tile: a single tile list longer than the LDS sorter (8192), sorted in two groups of
    depth bins."""
    N = 9000
    g = torch.Generator().manual_seed(1)
    sc = scenes.make_scene(N, 2, box=(0.02, 0.02, 0.5), scale_mean=0.002)
    sc["opacities"] = torch.full((N,), 0.02)
    vm = torch.eye(4)[None].clone()
    vm[0, 2, 3] = 2.0
    K = torch.tensor([[[60.0, 0, 24], [0, 60.0, 24], [0, 0, 1]]])
    cpu, gpu, out_c, out_g = _run_both(R, sc, vm, K, 48, 48, sh_degree=1)
    mg = out_g[2]
    tpg, ids, flat = O.isect_tiles_fast(mg["means2d"].detach().cpu(), mg["radii"].cpu(),
                                        mg["depths"].detach().cpu(), 16, 3, 3)
    assert torch.equal(mg["flatten_ids"].cpu(), flat)
    counts = torch.diff(torch.cat([mg["isect_offsets"].reshape(-1).cpu(),
                                   torch.tensor([flat.numel()], dtype=torch.int32)]))
    assert counts.max().item() > 8192
    _check(cpu, gpu, out_c, out_g)


@pytest.mark.parametrize("case", ["spread", "equal_depths", "one_big_tile", "equal_depths_one_big_tile"])
def test_long_buckets_of_short_tile_lists(R, case):
    """A bucket (8 adjacent tiles, the unit the sort kernel works on) holding more pairs than the LDS sorter
    takes (8192): the dense-scene case (2 M Gaussians seeded from depth maps at 1080p: 12 000 pairs per bucket,
    2 000 per tile), sorted group by group of consecutive (tile, depth) bins (isect_bucket.hip). With equal depths
    every tile is ONE bin and the groups take compare-exchange networks instead of the in-bin ranks; with equal
    depths and one tile of more than 4096 pairs no grouping is left and the bucket takes the global-memory network.
    (test_long_tile_lists: one tile list alone longer than the sorter.) The lists must equal the oracle's full sort,
    image and gradients the oracle's."""
    N = 18000
    sc = scenes.make_scene(N, 6, box=(1.8, 0.2, 0.3), scale_mean=0.002)
    sc["opacities"] = torch.full((N,), 0.02)
    if case.startswith("equal_depths"):
        sc["means"][:, 2] = 0.125
    if case.endswith("one_big_tile"):                      # 5 000 of them inside the third tile of the middle row
        sc["means"][:5000, 0] = -0.8 + 0.15 * (torch.rand(5000, generator=torch.Generator().manual_seed(8)) - 0.5)
    vm = torch.eye(4)[None].clone()
    vm[0, 2, 3] = 2.0
    W, H = 128, 48
    K = torch.tensor([[[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]]])
    cpu, gpu, out_c, out_g = _run_both(R, sc, vm, K, W, H, sh_degree=1)
    mg = out_g[2]
    tpg, ids, flat = O.isect_tiles_fast(mg["means2d"].detach().cpu(), mg["radii"].cpu(),
                                        mg["depths"].detach().cpu(), 16, W // 16, H // 16)
    assert torch.equal(mg["flatten_ids"].cpu(), flat)
    counts = torch.diff(torch.cat([mg["isect_offsets"].reshape(-1).cpu(),
                                   torch.tensor([flat.numel()], dtype=torch.int32)])).view(H // 16, W // 16)
    assert counts.max().item() <= 8192 and counts.sum(1).max().item() > 2 * 8192      # three groups at least
    assert (counts.max().item() > 4096) == case.endswith("one_big_tile")
    # (a 128 x 48 image: ONE threshold-boundary pixel is three of 18 432 values, 1.6e-4 of them)
    _check(cpu, gpu, out_c, out_g, flip_frac=2e-4)


@pytest.mark.parametrize("case", ["short_buckets", "long_bucket", "short_buckets_outliers"])
def test_clustered_depths_take_the_equalised_bins(R, case):
    """Gaussians on two thin shells (depth 1.4 and 2.6, 0.001 thick): linear depth bins over the bucket's [min, max]
    would hold a whole shell each, so the sort kernel re-draws its bins from the tiles' own depth distribution
    (isect_bucket.hip, 'Equalised bins'). The mapping must stay monotone in depth: the lists equal the oracle's full
    sort. Short buckets (keys parked in LDS), one long bucket (keys streamed, three groups), and shells plus outliers
    that stretch the range 20x."""
    g = torch.Generator().manual_seed(17)
    if case == "long_bucket":
        N, W, H, box = 18000, 128, 16, (1.8, 0.2, 0.3)
    else:
        N, W, H, box = 30000, 256, 64, (3.8, 0.9, 0.3)
    sc = scenes.make_scene(N, 12, box=box, scale_mean=0.002)
    shell = torch.where(torch.rand(N, generator=g) < 0.5, -0.6, 0.6)
    sc["means"][:, 2] = shell + 0.001 * torch.randn(N, generator=g)
    if case.endswith("outliers"):
        sc["means"][:60, 2] = -1.5 + 25.0 * torch.rand(60, generator=g)
    vm = torch.eye(4)[None].clone()
    vm[0, 2, 3] = 2.0
    K = torch.tensor([[[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]]])
    dev = {k: v.cuda() for k, v in sc.items()}
    with torch.no_grad():
        _, _, mg = R.rasterization(dev["means"], dev["quats"], dev["scales"], dev["opacities"],
                                   torch.cat([dev["sh0"], dev["shN"]], 1), vm.cuda(), K.cuda(), W, H, sh_degree=1, packed=False)
    tpg, ids, flat = O.isect_tiles_fast(mg["means2d"].cpu(), mg["radii"].cpu(), mg["depths"].cpu(), 16, W // 16, H // 16)
    assert torch.equal(mg["flatten_ids"].cpu(), flat)
    counts = torch.diff(torch.cat([mg["isect_offsets"].reshape(-1).cpu(),
                                   torch.tensor([flat.numel()], dtype=torch.int32)])).view(H // 16, W // 16)
    per_bucket = counts.view(H // 16, -1, 8).sum(-1)
    if case == "long_bucket":
        assert per_bucket.max().item() > 2 * 8192 and counts.max().item() <= 8192
    else:
        assert per_bucket.max().item() <= 8192 and counts.float().mean().item() > 120      # two bins of > 60 keys per tile


@pytest.mark.parametrize("tight", [False, True])
def test_needle_gaussians_tight_lists_vs_oracle(R, tight):
    """Strongly anisotropic, randomly rotated Gaussians (axis ratio 100-300: 2-D conic eigenvalues orders of
    magnitude apart) under tight lists: the exact pair test's edge constants kx = det / 2c, ky = det / 2a come
    from a compensated determinant (raster_common.h make_pair_conic); with the plain a - b^2/c a marginally
    visible pair or quadrant could be dropped. Image and gradients against the oracle, and the
    tight lists against the exact masks."""
    N = 900
    g = torch.Generator().manual_seed(21)
    sc = scenes.make_scene(N, 21, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    long_axis = torch.randint(0, 3, (N,), generator=g)
    scales = torch.full((N, 3), 0.0012) * (0.5 + torch.rand(N, 3, generator=g))
    scales[torch.arange(N), long_axis] = 0.08 + 0.25 * torch.rand(N, generator=g)
    sc["scales"] = scales
    sc["quats"] = torch.randn(N, 4, generator=g)
    sc["opacities"] = 0.3 + 0.69 * torch.rand(N, generator=g)
    W, H = 128, 96
    vm, K = scenes.cameras([5], width=W, height=H, f=120.0, dist=2.5)
    cpu, gpu, out_c, out_g = _run_both(R, sc, vm, K, W, H, tight=tight)
    _check(cpu, gpu, out_c, out_g)
    a, b, c = out_g[2]["conics"].detach()[0].unbind(-1)
    ratio = ((a + c) / 2 + ((a - c) ** 2 / 4 + b * b).sqrt()) / ((a + c) / 2 - ((a - c) ** 2 / 4 + b * b).sqrt()).clamp_min(1e-12)
    assert float(ratio[(out_g[2]["radii"][0] > 0).all(-1)].median()) > 30.0        # needles on screen, too


@pytest.mark.parametrize("antialiased", [False, True])
@pytest.mark.parametrize("cams", [[0], [0, 40]])
def test_runner_rasterize_splats_fused_activations(R, antialiased, cams):
    """runner.rasterize_splats (A1): exp / sigmoid and their backward run inside the
    projection kernels; gradients w.r.t. the RAW parameters must match autograd
    through torch.exp / torch.sigmoid + the oracle."""
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    sc = scenes.make_scene(700, 4, box=(1.0, 0.7, 0.4), scale_mean=0.03)
    W, H = 96, 64
    vm, K = scenes.cameras(cams, width=W, height=H, f=90.0, dist=2.5)
    raw = dict(means=sc["means"], quats=sc["quats"], scales=torch.log(sc["scales"]),
               opacities=torch.logit(sc["opacities"]), sh0=sc["sh0"], shN=sc["shN"])
    cpu = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    gpu = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().cuda()) for k, v in raw.items()})
    mode = "antialiased" if antialiased else "classic"
    rc_c, ra_c, _ = O.rasterization(
        cpu["means"], cpu["quats"], torch.exp(cpu["scales"]), torch.sigmoid(cpu["opacities"]),
        torch.cat([cpu["sh0"], cpu["shN"]], 1), vm, K, W, H, sh_degree=3, rasterize_mode=mode)
    cfg = runner.RasterConfig(antialiased=antialiased)
    rc_g, ra_g, info = runner.rasterize_splats(gpu, torch.linalg.inv(vm).cuda(), K.cuda(), W, H, cfg,
                                               sh_degree=3)
    g = torch.Generator().manual_seed(3)
    w_c, w_a = torch.randn(rc_c.shape, generator=g), torch.randn(ra_c.shape, generator=g)
    ((rc_c * w_c).sum() + (ra_c * w_a).sum()).backward()
    ((rc_g * w_c.cuda()).sum() + (ra_g * w_a.cuda()).sum()).backward()
    torch.cuda.synchronize()
    assert (rc_g.detach().cpu() - rc_c.detach()).abs().max() <= 5e-4
    for k in raw:
        got, ref = gpu[k].grad.cpu(), cpu[k].grad
        l2 = float((got - ref).norm() / ref.norm().clamp_min(1e-20))
        assert l2 <= GRAD_RTOL, f"grad {k}: L2 rel err {l2:.3e}"


def test_inverse4x4_matches_torch(R):
    vm, _ = scenes.cameras([0, 13, 77])
    inv, campos = R.inverse4x4(vm.cuda())
    ref = torch.linalg.inv(vm)
    assert torch.allclose(inv.cpu(), ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(campos.cpu(), ref[:, :3, 3], rtol=1e-5, atol=1e-6)


def test_tree_reduce8(R):
    """Lane semantics of the v_permlane32_swap / v_permlane16_swap / DPP halving
    tree the compositing backward reduces with."""
    lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 64, generator=g, dtype=torch.float32)
    xin = x.cuda().contiguous()
    out = torch.empty(192, device="cuda")
    idx = torch.empty(64, dtype=torch.int32, device="cuda")
    lib.call("gsr_debug_tree_reduce8", xin.data_ptr(), out.data_ptr(), idx.data_ptr(),
             torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tot = x.double().sum(1)
    idx = idx.cpu().long()
    assert sorted(set(idx.tolist())) == list(range(8))
    assert torch.allclose(out[:64].cpu().double(), tot[idx], atol=1e-4)
    assert torch.allclose(out[64:128].cpu().double(), tot[0].expand(64), atol=1e-4)
    assert float(out[191]) == pytest.approx(float(tot[1]), abs=1e-4)      # wave_sum_lane63


def test_bucketed_and_atomic_tile_lists_agree(R):
    """isect_bucket.hip (LDS histograms + per-bucket LDS sort) and isect.hip (per-tile
    atomics + per-tile sort) must produce identical offsets and lists."""
    sc = scenes.make_scene(20000, 12, box=(1.2, 0.8, 0.4), scale_mean=0.02)
    W, H = 300, 200                                   # 19 x 13 tiles: last bucket of a row is partial
    vm, K = scenes.cameras([0, 25, 50], width=W, height=H, f=250.0, dist=2.5)
    g = {k: sc[k].cuda() for k in ("means", "quats", "scales", "opacities")}
    col = torch.cat([sc["sh0"], sc["shN"]], 1).cuda()
    _, _, meta = R.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], col, vm.cuda(),
                                 K.cuda(), W, H, sh_degree=1, packed=False)
    tw, th = meta["tile_width"], meta["tile_height"]
    assert R.bucket_layout_ok(3, 20000, tw, th)
    m2d, radii, depths = meta["means2d"].detach(), meta["radii"], meta["depths"].detach()
    con, opa = meta["conics"].detach(), meta["opacities"].detach()
    off_b, ord_b, ids_b, _, _, pairs_b = R.isect_tiles_sorted(m2d, radii, depths, tw, th, conics=con,
                                                              opacities=opa)
    off_a, ord_a, ids_a, _, tpg, pairs_a = R.isect_tiles_sorted(m2d, radii, depths, tw, th,
                                                                want_tiles_per_gauss=True, conics=con,
                                                                opacities=opa)
    assert torch.equal(off_a, off_b) and torch.equal(ids_a, ids_b)
    # pair words: the emit pass of the bucketed builder and gsr_pair_masks evaluate the same test
    assert torch.equal(pairs_a, pairs_b)
    assert torch.equal(pairs_b & 0x07FFFFFF, ids_b)
    assert int(tpg.sum()) == ids_a.numel() == int(off_a[-1])
    n_tiles = 3 * tw * th
    for o in (ord_a, ord_b):                          # both work orders are permutations of the tiles
        assert sorted(o.cpu().tolist()) == list(range(n_tiles))
    lens = (off_a[1:] - off_a[:-1])[ord_a.long()]     # isect.hip: longest tile first (classes of 32)
    assert (((lens[:-1] + 31) // 32) >= ((lens[1:] + 31) // 32)).all()
    if not R.EXACT_TILE_ORDER:                        # bucketed: longest BUCKET first (classes of 128)
        bw = (tw + 7) // 8
        t = ord_b.long()
        bucket = (t // tw) * bw + (t % tw) // 8
        blen = torch.zeros(3 * th * bw, dtype=torch.long, device=t.device).index_add_(
            0, ((torch.arange(n_tiles, device=t.device) // tw) * bw + (torch.arange(n_tiles, device=t.device) % tw) // 8),
            (off_a[1:] - off_a[:-1]).long())
        cls = (blen[bucket] + 127) // 128
        assert (cls[:-1] >= cls[1:]).all()
    # and against the oracle's stable global sort
    _, ids, flat = O.isect_tiles_fast(m2d.cpu(), radii.cpu(), depths.cpu(), 16, tw, th)
    assert torch.equal(ids_b.cpu(), flat)
    assert torch.equal(off_b[:-1].cpu().view(3, th, tw), O.isect_offset_encode(ids, 3, tw, th))


@pytest.mark.parametrize("tight", [False, True])
def test_deferred_sync_capacity_overflow_is_rebuilt(R, tight):
    """The tile-list buffers are sized from the previous frame; a frame that outgrows
    the guess must be rebuilt transparently and give the same result. (The truncated lists of
    the overflowing frame are composited before the overflow is known: they must be
    self-consistent -- every listed pair word initialised -- or the kernels follow garbage ids.
    The buffers are poisoned here so that a gap would fault.)"""
    sc, vm, K, W, H = _tiny(N=3000)
    g = {k: sc[k].cuda() for k in ("means", "quats", "scales", "opacities")}
    col = torch.cat([sc["sh0"], sc["shN"]], 1).cuda()
    args = (g["means"], g["quats"], g["scales"], g["opacities"], col, vm.cuda(), K.cuda(), W, H)
    dev = g["means"].device.index
    R._IsectState.capacity.pop(dev, None)
    kw = dict(sh_degree=2, packed=False, _tight_tiles=tight)
    rc0, ra0, m0 = R.rasterization(*args, **kw)                             # blocking first frame
    n = m0["flatten_ids"].numel()
    assert R._IsectState.capacity[dev] >= n
    rc1, ra1, m1 = R.rasterization(*args, **kw)                             # deferred, fits
    for frac in (3, 7):
        R._IsectState.capacity[dev] = max(1, n // frac)                     # force an overflow
        # poison what the caching allocator will hand out next: 0x7f7f7f7f as an id is far outside
        # the record table
        junk = [torch.full((max(1, n // frac),), 0x7F7F7F7F, dtype=torch.int32, device="cuda") for _ in range(6)]
        del junk
        rc2, ra2, m2 = R.rasterization(*args, **kw)
        torch.cuda.synchronize()
    for rc, m in ((rc1, m1), (rc2, m2)):
        assert torch.equal(rc, rc0)
        assert torch.equal(m["flatten_ids"], m0["flatten_ids"])
        assert torch.equal(m["isect_offsets"], m0["isect_offsets"])
    assert R._IsectState.capacity[dev] >= n                                 # re-learnt


def _torch_min_sigma_rect(a, b, c, mx, my, x0, x1, y0, y1):
    """fp32 torch restatement of csrc/raster_common.h min_sigma_rect (test infrastructure)."""
    dxhi, dyhi = mx - x0, my - y0
    dxlo, dylo = dxhi - (x1 - x0), dyhi - (y1 - y0)
    inside = (dxlo <= 0) & (dxhi >= 0) & (dylo <= 0) & (dyhi >= 0)
    sig = lambda dx, dy: 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
    cl = lambda v, lo, hi: torch.minimum(torch.maximum(v, lo), hi)
    m = sig(dxlo, cl(-b * dxlo / c, dylo, dyhi))
    m = torch.minimum(m, sig(dxhi, cl(-b * dxhi / c, dylo, dyhi)))
    m = torch.minimum(m, sig(cl(-b * dylo / a, dxlo, dxhi), dylo))
    m = torch.minimum(m, sig(cl(-b * dyhi / a, dxlo, dxhi), dyhi))
    return torch.where(inside, torch.zeros_like(m), m)


def _pair_tiles(meta):
    """tile index of every entry of meta's lists."""
    offs = meta["isect_offsets"].reshape(-1).long()
    I = meta["flatten_ids"].numel()
    full = torch.cat([offs, torch.tensor([I], device=offs.device)])
    return torch.repeat_interleave(torch.arange(offs.numel(), device=offs.device), full[1:] - full[:-1])


@pytest.mark.parametrize("n_cams", [1, 2])
def test_pair_masks_are_exact_and_conservative(R, n_cams):
    """pair_ids = flatten id | quadrant mask << 28. Every (pixel, Gaussian) contribution the
    compositing would blend (alpha >= 1/255, evaluated densely in fp64 here) must lie in a
    quadrant whose bit is set; and the mask must agree with the fp32 restatement of the kernel's
    ellipse-vs-rectangle test except where that test's value sits within rounding of its threshold."""
    sc = scenes.make_scene(4000, 21, box=(1.2, 0.8, 0.4), scale_mean=0.02)
    W, H = 200, 120
    vm, K = scenes.cameras(list(range(0, 25 * n_cams, 25)), width=W, height=H, f=200.0, dist=2.5)
    g = {k: sc[k].cuda() for k in ("means", "quats", "scales", "opacities")}
    col = torch.cat([sc["sh0"], sc["shN"]], 1).cuda()
    _, _, meta = R.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], col, vm.cuda(),
                                 K.cuda(), W, H, sh_degree=1, packed=False)
    ids = meta["flatten_ids"].long()
    pairs = meta["pair_ids"]
    assert torch.equal((pairs & 0x07FFFFFF).long(), ids)
    mask = (pairs >> 28) & 15
    tile = _pair_tiles(meta)
    tw, th = meta["tile_width"], meta["tile_height"]
    tin = tile % (tw * th)
    tx0, ty0 = (tin % tw).float() * 16, (tin // tw).float() * 16
    m2 = meta["means2d"].detach().reshape(-1, 2)[ids]
    con = meta["conics"].detach().reshape(-1, 3)[ids]
    op = meta["opacities"].detach().reshape(-1)[ids % 4000]
    tau = torch.log(op * 255.0)
    tau_m = tau + 1e-4 * (1 + tau.abs())
    n_soft = 0
    for q in range(4):
        x0, y0 = tx0 + 8.0 * (q & 1) + 0.5, ty0 + 8.0 * (q >> 1) + 0.5
        ms = _torch_min_sigma_rect(con[:, 0], con[:, 1], con[:, 2], m2[:, 0], m2[:, 1], x0, x0 + 7, y0, y0 + 7)
        want = ms <= tau_m
        got = ((mask >> q) & 1).bool()
        soft = (ms - tau_m).abs() <= 1e-4 * (1 + tau_m.abs())      # within rounding of the threshold
        assert bool(((want == got) | soft).all())
        n_soft += int((want != got).sum())
        # dense fp64 check of conservativeness: max alpha over the quadrant's 64 pixel centres
        xs = torch.arange(8, device=m2.device, dtype=torch.float64)
        dx = m2[:, 0, None, None].double() - (x0.double()[:, None, None] + xs[None, None, :])
        dy = m2[:, 1, None, None].double() - (y0.double()[:, None, None] + xs[None, :, None])
        a, b, c = (con[:, k, None, None].double() for k in range(3))
        sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
        alpha = op.double()[:, None, None] * torch.exp(-sigma)
        reach = ((alpha >= 1.0 / 255.0) & (sigma >= 0)).flatten(1).any(1)
        assert bool((got | ~reach).all()), "a quadrant with alpha >= 1/255 is masked out"
    assert n_soft <= max(2, ids.numel() // 10000)


def test_tight_tile_lists_same_render_and_gradients(R):
    """`_tight_tiles=True` (runner.rasterize_splats) drops exactly the pairs whose mask is 0; the
    compositing kernels skip those anyway, so image and alpha are bit-identical, the gradients equal
    up to the order of the float atomics, and the lists are the default lists minus the dead pairs."""
    sc, vm, K, W, H = _tiny(N=3000)
    res = {}
    for tight in (False, True):
        cpu, gpu, out_c, out_g = _run_both(R, sc, vm, K, W, H, tight=tight, absgrad=True)
        if tight:
            _check(cpu, gpu, out_c, out_g)              # and the oracle agrees with the tight path
        res[tight] = (gpu, out_g)
    (g0, (rc0, ra0, m0)), (g1, (rc1, ra1, m1)) = res[False], res[True]
    assert torch.equal(rc0, rc1) and torch.equal(ra0, ra1)
    live = ((m0["pair_ids"] >> 28) & 15) != 0
    assert int((~live).sum()) > 0, "scene has no dead pair: the test would be vacuous"
    assert torch.equal(m0["flatten_ids"][live], m1["flatten_ids"])
    assert torch.equal(m0["pair_ids"][live], m1["pair_ids"])
    t0 = _pair_tiles(m0)[live]
    assert torch.equal(t0, _pair_tiles(m1))
    for k in g0:
        a, b = g0[k].grad, g1[k].grad
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12, k
    ab0, ab1 = m0["means2d"].absgrad, m1["means2d"].absgrad
    assert float((ab0 - ab1).abs().max()) <= 1e-5 * float(ab0.abs().max()) + 1e-12


def test_tight_lists_long_rectangles(R):
    """Rectangles of more than 16 tiles take the emit pass's own mask evaluation (no stored word)."""
    sc = scenes.make_scene(300, 9, box=(0.8, 0.5, 0.3), scale_mean=0.15, scale_std=0.6)
    W, H = 160, 128
    vm, K = scenes.cameras([0], width=W, height=H, f=150.0, dist=2.5)
    g = {k: sc[k].cuda() for k in ("means", "quats", "scales", "opacities")}
    col = torch.cat([sc["sh0"], sc["shN"]], 1).cuda()
    args = (g["means"], g["quats"], g["scales"], g["opacities"], col, vm.cuda(), K.cuda(), W, H)
    rc0, ra0, m0 = R.rasterization(*args, sh_degree=1, packed=False)
    rc1, ra1, m1 = R.rasterization(*args, sh_degree=1, packed=False, _tight_tiles=True)
    rad = m0["radii"].reshape(-1, 2).float()
    assert int(((2 * rad[:, 0] / 16 + 1) * (2 * rad[:, 1] / 16 + 1) > 16).sum()) > 10, "no large rectangle"
    live = ((m0["pair_ids"] >> 28) & 15) != 0
    assert torch.equal(m0["pair_ids"][live], m1["pair_ids"])
    assert torch.equal(rc0, rc1) and torch.equal(ra0, ra1)


def test_bucketed_lists_more_than_one_trip_per_workgroup(R):
    """More than 2^20 (camera, Gaussian) pairs: every workgroup of the bucketed builder's count and
    emit passes walks its Gaussians in more than one trip (the first trip's values are loaded at
    the kernel top, the later ones where they are used). Lists and pair words must equal the
    per-tile builder's, and the tight lists must be those lists without the mask-0 pairs."""
    C, N, W, H = 2, 600_000, 640, 368
    tw, th = W // 16, H // 16
    gen = torch.Generator().manual_seed(5)
    m2d = torch.rand(C, N, 2, generator=gen) * torch.tensor([W + 40.0, H + 40.0]) - 20.0
    radii = torch.randint(0, 14, (C, N, 2), generator=gen, dtype=torch.int32)
    radii[:, ::97] = torch.randint(20, 90, radii[:, ::97].shape, generator=gen, dtype=torch.int32)
    radii[:, ::5] = 0                                                        # culled Gaussians
    depths = torch.rand(C, N, generator=gen) * 9.0 + 0.2
    s = (radii.float().clamp(min=1.0) / 3.0) ** 2                            # conic ~ 3-sigma extent
    con = torch.stack([1.0 / s[..., 0], (torch.rand(C, N, generator=gen) - 0.5) * 0.6 / (s[..., 0] * s[..., 1]).sqrt(),
                       1.0 / s[..., 1]], -1)
    opa = torch.rand(N, generator=gen) * 0.98 + 0.01
    opa[::11] = 0.9995                                                       # clamp flag set
    m2d, radii, depths, con, opa = (t.cuda().contiguous() for t in (m2d, radii, depths, con, opa))
    assert C * N > (1 << 20) and R.bucket_layout_ok(C, N, tw, th)
    off_a, _, ids_a, _, _, pairs_a = R.isect_tiles_sorted(m2d, radii, depths, tw, th, want_tiles_per_gauss=True,
                                                          conics=con, opacities=opa)
    off_b, _, ids_b, _, _, pairs_b = R.isect_tiles_sorted(m2d, radii, depths, tw, th, conics=con, opacities=opa)
    assert torch.equal(off_a, off_b) and torch.equal(ids_a, ids_b) and torch.equal(pairs_a, pairs_b)
    off_t, _, ids_t, _, _, pairs_t = R.isect_tiles_sorted(m2d, radii, depths, tw, th, conics=con, opacities=opa,
                                                          tight=True)
    keep = (pairs_a >> 28) & 15 != 0
    assert 0.3 < float(keep.float().mean()) < 0.98
    assert torch.equal(pairs_t, pairs_a[keep]) and torch.equal(ids_t, ids_a[keep])
    tile_of = torch.repeat_interleave(torch.arange(C * tw * th, device=off_a.device),
                                      (off_a[1:] - off_a[:-1]).long())
    cnt = torch.zeros(C * tw * th, dtype=torch.long, device=off_a.device).index_add_(0, tile_of[keep], torch.ones_like(tile_of[keep]))
    assert torch.equal((off_t[1:] - off_t[:-1]).long(), cnt)
