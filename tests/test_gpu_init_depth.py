"""GPU parity of the monocular-depth init kernels (rows B1-B9) against the
golden vectors of the reference's own modules (tests/golden/init_golden.npz)
and the CPU oracle (oracle/init_oracle.py).

Bars: masks / integer coordinates bit-exact; scale/shift within 1e-5 relative
(the kernel sums in fp64, the reference in fp32); aligned depth within 1e-5
relative; unprojected points within 1e-5 relative to the scene extent.
"""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import init_oracle as IO

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).resolve().parent / "golden" / "init_golden.npz")
P_ = "3dgs_monocular_depth_init_amd."


def mod(name):
    return importlib.import_module(P_ + name)


def _t(name):
    return torch.from_numpy(G[name])


def _bits(name, n):
    return torch.from_numpy(np.unpackbits(G[name])[:n].astype(bool))


def _pd(depth, mask):
    return mod("depth_prediction.predictors.depth_predictor_interface").PredictedDepth(
        depth=depth.cuda(), mask=mask.cuda())


@pytest.mark.parametrize("i", [0, 1, 2])
def test_lstsq_vs_reference_golden(i):
    L = mod("depth_alignment.alignment.lstsqrs")
    depth, mask = _t(f"lsq{i}_depth"), _t(f"lsq{i}_mask")
    coords, gt = _t(f"lsq{i}_coords"), _t(f"lsq{i}_gt")
    res = L.DepthAlignmentLstSqrs.align(_pd(depth, mask), coords.cuda(), gt.cuda())
    d = L.gather_depth(depth.cuda(), coords.cuda())
    assert torch.equal(d.cpu(), depth[coords[1], coords[0]])
    s, t = L.align_depth_least_squares(torch.vstack([d, torch.ones_like(d)]), gt.cuda())
    ref = G[f"lsq{i}_scale_shift"]
    assert float(s) == pytest.approx(ref[0], rel=1e-5)
    assert float(t) == pytest.approx(ref[1], abs=2e-4)      # fp64 sums here, fp32 in the reference
    assert torch.allclose(res.aligned_depth.cpu(), _t(f"lsq{i}_aligned"), rtol=1e-5, atol=1e-5)
    assert torch.equal(res.mask.cpu(), mask)


@pytest.mark.parametrize("i", [0, 1])
def test_masks_vs_reference_golden(i):
    S = mod("depth_subsampling")
    depth, mask, coords = _t(f"msk{i}_depth"), _t(f"msk{i}_mask"), _t(f"msk{i}_coords")
    H, W = depth.shape
    rgb = torch.zeros(H, W, 3).cuda()
    for k in (3, 10):
        m = S.StaticDepthSubsampler(k).get_mask(rgb, depth.cuda(), mask.cuda())
        assert m.dtype == torch.bool and torch.equal(m.cpu(), _bits(f"msk{i}_static{k}", H * W))
    m = S.AdaptiveDepthSubsampler(S.AdaptiveSubsamplingConfig()).get_mask(rgb, depth.cuda(), mask.cuda())
    assert torch.equal(m.cpu(), _bits(f"msk{i}_adaptive", H * W))
    A = mod("depth_subsampling.adaptive_subsampling")
    lo, hi = A.iqr_outlier_bounds(depth.cuda()[mask.cuda()])
    assert [float(lo), float(hi)] == pytest.approx(list(G[f"msk{i}_iqr"]), rel=1e-6)
    nps, thr = (int(v) for v in G[f"msk{i}_sfmcfg"])
    m = S.num_sfm_points_mask(coords.cuda(), (H, W), S.NumSfMPointsMaskConfig(nps, thr))
    assert torch.equal(m.cpu().reshape(-1), _bits(f"msk{i}_sfmmask", H * W))


def _scene(H=270, W=480, n_sfm=3000, seed=3):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    t = torch.clamp((xx + 0.5 * yy) / 1.5, 0, 1)
    true_depth = 2.0 + 6.0 * (t * t * (3 - 2 * t))
    pred = ((true_depth - 0.4) / 1.7 + 0.01 * torch.randn(H, W, generator=g)).float()
    mask = torch.rand(H, W, generator=g) > 0.03
    K = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1.0]])
    th = 0.3
    c2w = torch.eye(4)
    c2w[:3, :3] = torch.tensor([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    c2w[:3, 3] = torch.tensor([0.3, -0.2, 0.5])
    # SfM points: unproject random pixels at the TRUE depth (+noise, 20% gross outliers, some off-image)
    xs = torch.rand(n_sfm, generator=g) * (W * 1.2) - 0.1 * W
    ys = torch.rand(n_sfm, generator=g) * (H * 1.2) - 0.1 * H
    xi, yi = xs.clamp(0, W - 1).long(), ys.clamp(0, H - 1).long()
    z = true_depth[yi, xi] + 0.02 * torch.randn(n_sfm, generator=g)
    outl = torch.rand(n_sfm, generator=g) < 0.2
    z = torch.where(outl, z * (0.3 + 2.7 * torch.rand(n_sfm, generator=g)), z)
    cam = torch.stack([(xs - K[0, 2]) / K[0, 0] * z, (ys - K[1, 2]) / K[1, 1] * z, z], 1)
    world = (c2w[:3, :3] @ cam.T).T + c2w[:3, 3]
    rgb = torch.rand(H, W, 3, generator=g)
    return pred, mask, K, c2w, world.float(), rgb


def test_project_and_filter_sfm_pts_vs_oracle():
    PF = mod("depth_prediction.points_from_depth")
    pred, mask, K, c2w, world, rgb = _scene()
    H, W = pred.shape
    R, C = c2w[:3, :3].T, c2w[:3, 3]
    P = (K @ R @ torch.hstack([torch.eye(3), -C[:, None]])).float()
    co, de = IO.project_and_filter_sfm_pts(world, P, (W, H), mask)
    cg, dg = PF.project_and_filter_sfm_pts(None, world.cuda(), P.cuda(), (W, H), _pd(pred, mask))
    assert cg.dtype == torch.int64 and cg.shape == co.shape
    assert torch.equal(cg.cpu(), co)
    assert torch.allclose(dg.cpu(), de, rtol=1e-6, atol=1e-6)
    # < 1/4 in bounds -> the reference's exception
    with pytest.raises(mod("depth_alignment").LowDepthAlignmentConfidenceError):
        PF.project_and_filter_sfm_pts(None, (world + 100.0).cuda(), P.cuda(), (W, H), _pd(pred, mask))


@pytest.mark.parametrize("seed", [42, 7, 123, 2024])
@pytest.mark.parametrize("loss", ["ransac", "msac"])
def test_ransac_vs_oracle(loss, seed):
    Rm = mod("depth_alignment.alignment.ransacs")
    cfgm = mod("depth_alignment.config")
    pred, mask, K, c2w, world, rgb = _scene()
    H, W = pred.shape
    R, C = c2w[:3, :3].T, c2w[:3, 3]
    P = (K @ R @ torch.hstack([torch.eye(3), -C[:, None]])).float()
    co, de = IO.project_and_filter_sfm_pts(world, P, (W, H), mask)
    torch.manual_seed(seed)
    s_o, t_o, aligned_o, it_o, inl_o = IO.ransac_align(pred, co, de, loss, IO.RansacConfig())
    torch.manual_seed(seed)
    res, st = Rm._align_depth_ransac_generic(_pd(pred, mask), co.cuda(), de.cuda(), loss,
                                             cfgm.RansacConfig(), return_stats=True)
    rng_after = torch.get_rng_state()
    # the accept rule replays the reference's decisions: with identical sample
    # streams both converge to the same consensus set
    assert st["scale"] == pytest.approx(float(s_o), rel=1e-3)
    assert st["shift"] == pytest.approx(float(t_o), rel=1e-2, abs=1e-3)
    assert abs(st["inliers"] - inl_o) <= max(2, 0.01 * inl_o)
    assert torch.allclose(res.aligned_depth.cpu(), aligned_o, rtol=2e-3, atol=2e-3)
    # the global RNG is left exactly where `iterations + 1` reference draws leave it
    torch.manual_seed(seed)
    for _ in range(st["iterations"] + 1):
        torch.randperm(co.shape[1])
    assert torch.equal(rng_after, torch.get_rng_state())
    assert st["iterations"] == it_o


@pytest.mark.parametrize("i", range(8))
def test_ransac_vs_reference_golden(i):
    """The HIP RANSAC/MSAC against a recorded run of the REFERENCE's own ransacs.py
    (tests/golden/ransac_golden.npz): same RNG stream, same iteration at which the
    adaptive stop fires, same consensus set; scale/shift agree to fp32 LSQ accuracy
    (normal equations in fp64 here, fp32 einsum + pinv in the reference)."""
    from tests.test_init_oracle_golden import ransac_case
    Rm = mod("depth_alignment.alignment.ransacs")
    cfgm = mod("depth_alignment.config")
    c = ransac_case(i)
    torch.manual_seed(c["seed"])
    res, st = Rm._align_depth_ransac_generic(_pd(c["depth"], c["mask"]), c["coords"].cuda(), c["gt"].cuda(),
                                             c["loss"], cfgm.RansacConfig(**c["cfg"]), return_stats=True)
    assert st["iterations"] == c["iterations"]
    assert abs(st["inliers"] - c["inliers"]) <= max(1, 0.005 * c["inliers"])
    assert st["scale"] == pytest.approx(c["scale_shift"][0], rel=1e-3)
    assert st["shift"] == pytest.approx(c["scale_shift"][1], rel=1e-2, abs=2e-3)
    assert torch.allclose(res.aligned_depth.cpu(), c["aligned"], rtol=2e-3, atol=2e-3)


def test_ransac_scoring_matches_oracle_for_given_hypotheses():
    """Bit-level check of the scoring kernels on fixed hypotheses (decisions
    depend only on these numbers)."""
    lib = mod("_lib")
    g = torch.Generator().manual_seed(0)
    M, T = 3001, 37
    d = torch.rand(M, generator=g) * 5 + 1
    gt = 1.7 * d + 0.4 + 0.05 * torch.randn(M, generator=g)
    hyp = torch.stack([1.7 + 0.2 * torch.randn(T, generator=g), 0.4 + 0.2 * torch.randn(T, generator=g)], 1)
    dc, gc, hc = d.cuda(), gt.cuda(), hyp.cuda().contiguous()
    o_r = torch.empty(T, dtype=torch.int32, device="cuda")
    o_i = torch.empty_like(o_r)
    o_m = torch.empty(T, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib.call("gsr_ransac_score", T, M, hc.data_ptr(), dc.data_ptr(), gc.data_ptr(), 0.01,
             o_r.data_ptr(), o_m.data_ptr(), o_i.data_ptr(), st)
    for k in range(T):
        dists = (hyp[k, 0] * d + hyp[k, 1] - gt) ** 2
        assert int(o_r[k]) == int(IO.ransac_loss(dists, 0.01))
        assert int(o_i[k]) == int((dists < 0.01).sum())
        assert float(o_m[k]) == pytest.approx(float(IO.msac_loss(dists, 0.01)), rel=1e-5)


@pytest.mark.parametrize("subsample", [10, "adaptive"])
@pytest.mark.parametrize("aligner", ["lstsqrs", "ransac"])
def test_get_pts_from_depth_vs_oracle_chain(subsample, aligner):
    PF = mod("depth_prediction.points_from_depth")
    cfgm = mod("config")
    dac = mod("depth_alignment.config")
    types = mod("types")
    pred, mask, K, c2w, world, rgb = _scene()
    H, W = pred.shape
    cfg = cfgm.Config()
    cfg.mdi.subsample_factor = subsample
    cfg.mdi.alignment.aligner = dac.DepthAlignmentStrategyEnum(aligner)
    cfg.mdi.depth_grad_mask_thresh = 0.5
    image = types.InputImage(data=rgb, name="img0", cam2world=c2w, K=K)
    torch.manual_seed(42)
    pts, fmask, P, rgbs = PF.get_pts_from_depth(_pd(pred, mask), image, world, cfg, "cuda",
                                               return_rgb=True)
    # oracle chain on the CPU with the same RNG stream
    R, C = c2w[:3, :3].T, c2w[:3, 3]
    P_o = (K @ R @ torch.hstack([torch.eye(3), -C[:, None]])).float()
    co, de = IO.project_and_filter_sfm_pts(world, P_o, (W, H), mask)
    torch.manual_seed(42)
    if aligner == "lstsqrs":
        _, _, aligned = IO.lstsq_align(pred, co, de)
    else:
        _, _, aligned, _, _ = IO.ransac_align(pred, co, de, "ransac", IO.RansacConfig())
    out_depth, omask = IO.pipeline_align_noseg(aligned, mask, mask)
    # masks are computed from the DEVICE's aligned depth so that ulp-level
    # scale/shift differences cannot move a pixel across a stride boundary
    ad = PF.DepthAlignmentPipeline.from_config(cfg)   # noqa: F841 (import check)
    sub = (IO.static_mask((H, W), subsample, omask) if subsample != "adaptive"
           else IO.adaptive_mask((H, W, 3), out_depth.clone(), omask))
    pts_o, mask_o = IO.assemble_mask_and_unproject(out_depth, omask, sub, K, c2w, co,
                                                   depth_grad_mask_thresh=0.5)
    differing = int((fmask != mask_o).sum())
    assert differing <= max(1, int(1e-4 * H * W)), f"{differing} mask pixels differ"
    # points on EVERY pixel both sides kept (never skipped)
    common = fmask & mask_o
    rows_h = (torch.cumsum(fmask.long(), 0) - 1)[common]
    rows_o = (torch.cumsum(mask_o.long(), 0) - 1)[common]
    extent = pts_o.abs().max()
    assert (pts.cpu()[rows_h] - pts_o[rows_o]).abs().max() <= 3e-3 * extent      # scale/shift differ ~1e-3 (ransac)
    assert torch.equal(rgbs.cpu()[rows_h], rgb.view(-1, 3)[mask_o][rows_o])
    assert pts.shape[0] == int(fmask.sum()) and pts.shape[0] > 500


def test_unprojection_exact_inputs_vs_oracle():
    """Same aligned depth and masks on both sides: unprojection within 1e-5."""
    PF = mod("depth_prediction.points_from_depth")
    pred, mask, K, c2w, world, rgb = _scene(H=123, W=217)
    H, W = pred.shape
    aligned = (pred * 1.7 + 0.4)
    aligned[5, 7] = -1.0                                    # negative depth is dropped
    sub = IO.static_mask((H, W), 4, mask)
    pts_o, mask_o = IO.assemble_mask_and_unproject(aligned, mask, sub, K, c2w, None,
                                                   use_num_sfm_points_mask=False)
    pts, rgbs, fmask = PF.unproject_masked(aligned.cuda(), mask.cuda(), sub.cuda(), None,
                                           rgb.cuda(), K, c2w)
    assert torch.equal(fmask.cpu(), mask_o)
    assert torch.allclose(pts.cpu(), pts_o, rtol=1e-5, atol=1e-5)
    assert torch.equal(rgbs.cpu(), rgb.view(-1, 3)[mask_o])
    g = PF.depth_gradient_mask(aligned.cuda(), 0.3)
    assert torch.equal(g.cpu(), IO.depth_gradient_mask(aligned, 0.3))


def test_c3_full_size_properties():
    """BASELINE config c3 shape: 1080x1920, static k=10 -> 108*192 = 20 736 seeds
    when every pixel is valid; compaction preserves pixel order."""
    PF = mod("depth_prediction.points_from_depth")
    S = mod("depth_subsampling")
    H, W = 1080, 1920
    g = torch.Generator().manual_seed(3)
    depth = (2 + 6 * torch.rand(H, W, generator=g)).cuda()
    mask = torch.ones(H, W, dtype=torch.bool).cuda()
    sub = S.StaticDepthSubsampler(10).get_mask(None, depth, mask)
    assert int(sub.sum()) == 108 * 192
    K = torch.tensor([[1200.0, 0, 960], [0, 1200.0, 540], [0, 0, 1]])
    pts, _, fmask = PF.unproject_masked(depth, mask, sub, None, None, K, torch.eye(4))
    assert pts.shape == (20736, 3) and torch.equal(fmask, sub)
    # identity camera: z == depth at the kept pixels, in row-major pixel order
    assert torch.allclose(pts[:, 2], depth.view(-1)[sub], rtol=1e-6)
    # applying the mask twice changes nothing (idempotence)
    sub2 = S.StaticDepthSubsampler(10).get_mask(None, depth, sub.view(H, W))
    assert torch.equal(sub2, sub)


class _FakeMetric3dNet:
    """Stand-in for the torch.hub network (remote weights): a deterministic
    function of the pre-processed input so the whole predictor can be compared."""

    def inference(self, data):
        x = data["input"]                                    # [1,3,616,1064]
        depth = (x.mean(1, keepdim=True) * 0.7 + 3.0)
        conf = torch.sigmoid(x[:, :1])
        normal = torch.cat([x, x[:, :1] * 0.5], 1)
        return depth, conf, {"prediction_normal": normal}


@pytest.mark.parametrize("size", [(270, 480), (480, 360), (1080, 1920)])
def test_metric3d_pre_post_processing_vs_oracle(size):
    M = mod("depth_prediction.predictors.metric3d")
    ifc = mod("depth_prediction.predictors.depth_predictor_interface")
    H, W = size
    g = torch.Generator().manual_seed(4)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    img = (torch.stack([xx, yy, 1 - xx], -1) * 0.8 + 0.2 * torch.rand(H, W, 3, generator=g)).clamp(0, 1)
    net_in, pad_info, scale = M.preprocess(img.cuda())
    ref_in, ref_pad, ref_scale = IO.metric3d_preprocess(img)
    assert pad_info == ref_pad and scale == ref_scale and net_in.shape == (1, 3, 616, 1064)
    diff = (net_in.cpu() - ref_in).abs()
    one_lsb = 1.0 / 57.0
    assert diff.max() <= one_lsb * 1.01                      # at most one uint8 step (rounding ties)
    assert (diff > 1e-5).float().mean() < 2e-3
    K = torch.tensor([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1.0]])
    pred = M.Metric3d(None, "cuda", model=_FakeMetric3dNet()).predict_depth(img.cuda(), ifc.CameraIntrinsics(K))
    d_ref, c_ref, o_ref = _FakeMetric3dNet().inference({"input": net_in.cpu()})
    depth_ref = IO.metric3d_postprocess_depth(d_ref.squeeze(), pad_info, (H, W), float(K[0, 0]), scale)
    assert pred.depth.shape == (H, W) and pred.mask.all() and pred.mask.dtype == torch.bool
    assert torch.allclose(pred.depth.cpu(), depth_ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(pred.depth_confidence.cpu(),
                          IO.metric3d_to_og_size(c_ref.squeeze(), pad_info, (H, W)), rtol=1e-5, atol=1e-5)
    n_ref = torch.stack([IO.metric3d_to_og_size(o_ref["prediction_normal"][0, k], pad_info, (H, W))
                         for k in range(3)], -1)
    assert pred.normal.shape == (H, W, 3) and torch.allclose(pred.normal.cpu(), n_ref, rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        M.Metric3d(None, "cuda")                              # no silent torch.hub fetch


# --------------------------------------------------------------------------------------------- #
# B1 / B4 / B8 / B9 against outputs of the REFERENCE's own points_from_depth.py / pipeline.py
# (tests/golden/points_golden.npz, generated by tests/golden/make_points_golden.py)
# --------------------------------------------------------------------------------------------- #
from tests import points_golden as PG  # noqa: E402


@pytest.mark.parametrize("i", range(int(PG.G["b1_n"])))
def test_b1_project_sfm_vs_reference_golden(i):
    PF = mod("depth_prediction.points_from_depth")
    sc = PG.scene(f"b1_{i}")
    H, W = sc["depth"].shape
    cg, dg = PF.project_and_filter_sfm_pts(None, sc["sfm"].cuda(), sc["P"].cuda(), (W, H),
                                           _pd(sc["depth"], sc["mask"]))
    assert cg.dtype == torch.int64 and torch.equal(cg.cpu(), PG.t(f"b1_{i}_coords"))     # bit-exact
    assert torch.allclose(dg.cpu(), PG.t(f"b1_{i}_depths"), rtol=1e-6, atol=1e-6)


def test_b1_low_confidence_branch_vs_reference_golden():
    PF = mod("depth_prediction.points_from_depth")
    sc = PG.scene("b1_err")               # the reference raised on exactly this input
    with pytest.raises(mod("depth_alignment").LowDepthAlignmentConfidenceError):
        PF.project_and_filter_sfm_pts(None, sc["sfm"].cuda(), sc["P"].cuda(), (96, 64),
                                      _pd(sc["depth"], sc["mask"]))


@pytest.mark.parametrize("i", range(int(PG.G["b8_n"])))
def test_b8_depth_gradient_mask_vs_reference_golden(i):
    PF = mod("depth_prediction.points_from_depth")
    d = PG.scene(f"b8_{i}")["depth"]
    m = PF.depth_gradient_mask(d.cuda(), float(PG.G[f"b8_{i}_thr"]))
    assert torch.equal(m.cpu(), PG.t(f"b8_{i}_gradmask"))                                    # bit-exact


def _hip_cfg(i):
    cfgm, dac = mod("config"), mod("depth_alignment.config")
    aligner, factor, grad_thr, nsfm = PG.b9_cfg(i)
    cfg = cfgm.Config()
    cfg.mdi.alignment.aligner = dac.DepthAlignmentStrategyEnum[aligner]
    cfg.mdi.subsample_factor = factor
    cfg.mdi.depth_grad_mask_thresh = grad_thr
    cfg.mdi.use_num_sfm_points_mask = nsfm
    return cfg, aligner


class _GoldenParser:
    """The attributes get_pts_from_depth reads from the reference's Parser
    (points_from_depth.py:233-237)."""

    def __init__(self, name, pts):
        self.points = pts.numpy()
        self.point_indices = {name: np.arange(pts.shape[0])}


@pytest.mark.parametrize("i", range(int(PG.G["b9_n"])))
def test_b4_pipeline_noseg_vs_reference_golden(i):
    """DepthAlignmentPipeline.align (no segmenter) against the reference's aligned depth map
    and result mask. lstsqrs: fp32 LSQ accuracy; ransac/msac: the tolerance of the pinned
    RANSAC test (consensus set reproduced, LSQ in fp64 here vs fp32 einsum + pinv there)."""
    PF = mod("depth_prediction.points_from_depth")
    types = mod("types")
    sc = PG.scene(f"b9_{i}")
    cfg, aligner = _hip_cfg(i)
    H, W = sc["depth"].shape
    pd = _pd(sc["depth"], sc["mask"])
    co, de = PF.project_and_filter_sfm_pts(None, sc["sfm"].cuda(), sc["P"].cuda(), (W, H), pd)
    image = types.InputImage(data=sc["rgb"], name="img0", cam2world=sc["c2w"], K=sc["K"])
    torch.manual_seed(int(PG.G[f"b9_{i}_rng_seed"]))
    res = PF.DepthAlignmentPipeline.from_config(cfg).align(image, pd, co, de, cfg, None)
    tol = 1e-5 if aligner == "lstsqrs" else 2e-3
    assert torch.allclose(res.aligned_depth.cpu(), PG.t(f"b9_{i}_aligned"), rtol=tol, atol=tol)
    assert torch.equal(res.mask.flatten().cpu(), PG.bits(f"b9_{i}_align_mask", H * W))    # bit-exact


@pytest.mark.parametrize("i", range(int(PG.G["b9_n"])))
def test_b9_masks_and_unprojection_on_the_reference_aligned_depth(i):
    """Integer work given IDENTICAL inputs: every mask kernel and the compaction are fed the
    reference's own aligned depth map and result mask (stored in the fixture), so the final
    mask must be torch.equal to the reference's and the point comparison is never skipped."""
    PF = mod("depth_prediction.points_from_depth")
    S = mod("depth_subsampling")
    sc = PG.scene(f"b9_{i}")
    _, factor, grad_thr, nsfm = PG.b9_cfg(i)
    out_depth = PG.t(f"b9_{i}_aligned")
    H, W = out_depth.shape
    omask = PG.bits(f"b9_{i}_align_mask", H * W).view(H, W)
    co, _ = PF.project_and_filter_sfm_pts(None, sc["sfm"].cuda(), sc["P"].cuda(), (W, H),
                                          _pd(sc["depth"], sc["mask"]))      # bit-exact (test_b1_*)
    co = co.cpu()
    cfg, _ = _hip_cfg(i)
    depth_g, mask_g = out_depth.cuda(), omask.cuda()
    sub = PF.get_subsampler(cfg).get_mask(sc["rgb"].cuda(), depth_g, mask_g)
    extra = None
    if grad_thr is not None:
        extra = PF.depth_gradient_mask(depth_g, grad_thr).flatten()
    if nsfm:
        m = S.num_sfm_points_mask(co.cuda(), (H, W), cfg.mdi.num_sfm_points_mask).flatten()
        extra = m if extra is None else (extra & m)
    pts, rgbs, fmask = PF.unproject_masked(depth_g, mask_g, sub, extra, sc["rgb"].cuda(), sc["K"], sc["c2w"])
    ref_mask = PG.bits(f"b9_{i}_final_mask", H * W)
    assert torch.equal(fmask.cpu(), ref_mask)                                            # bit-exact
    ref_pts = PG.t(f"b9_{i}_pts")
    assert pts.shape == ref_pts.shape
    extent = float(ref_pts.abs().max())
    assert float((pts.cpu() - ref_pts).abs().max()) <= 1e-5 * extent
    assert torch.equal(rgbs.cpu(), sc["rgb"].view(-1, 3)[ref_mask])


@pytest.mark.parametrize("i", range(int(PG.G["b9_n"])))
def test_b9_get_pts_from_depth_vs_reference_golden(i):
    """The whole chain through the reference's own signature
    `get_pts_from_depth(predicted_depth, image, parser, config, device, debug_export_dir)`.
    The aligned depth differs from the reference's in the last bits (fp64 vs fp32 LSQ sums), so
    a pixel sitting exactly on a stride / threshold boundary may change sides: the masks must
    agree on all but <= 1e-4 of the pixels and the points are compared on EVERY common pixel."""
    PF = mod("depth_prediction.points_from_depth")
    types = mod("types")
    sc = PG.scene(f"b9_{i}")
    cfg, aligner = _hip_cfg(i)
    H, W = sc["depth"].shape
    image = types.InputImage(data=sc["rgb"], name="img0", cam2world=sc["c2w"], K=sc["K"])
    torch.manual_seed(int(PG.G[f"b9_{i}_rng_seed"]))
    pts, fmask, P = PF.get_pts_from_depth(_pd(sc["depth"], sc["mask"]), image,
                                          _GoldenParser("img0", sc["sfm"]), cfg, "cuda", None)
    assert torch.allclose(P.cpu(), PG.t(f"b9_{i}_P"), rtol=1e-6, atol=1e-6)
    ref_mask, ref_pts = PG.bits(f"b9_{i}_final_mask", H * W), PG.t(f"b9_{i}_pts")
    differing = int((fmask != ref_mask).sum())
    from tests import parity_log
    parity_log.record("init_chain", case=i, aligner=aligner, pixels=H * W, mask_pixels_differing=differing,
                      points=int(fmask.sum()), points_reference=int(ref_mask.sum()))
    assert differing <= max(1, int(1e-4 * H * W)), f"{differing} mask pixels differ"
    common = fmask & ref_mask
    rows_h = (torch.cumsum(fmask.long(), 0) - 1)[common]
    rows_r = (torch.cumsum(ref_mask.long(), 0) - 1)[common]
    assert rows_h.numel() > 100
    tol = (1e-5 if aligner == "lstsqrs" else 3e-3) * float(ref_pts.abs().max())
    assert float((pts.cpu()[rows_h] - ref_pts[rows_r]).abs().max()) <= tol
