"""Pin oracle/init_oracle.py rows B1 / B4 / B8 / B9 against outputs of the reference's own
depth_prediction/points_from_depth.py and depth_alignment/pipeline.py
(tests/golden/make_points_golden.py; runs without /root/reference)."""
import pytest
import torch

from oracle import init_oracle as IO
from tests import points_golden as PG

G = PG.G


@pytest.mark.parametrize("i", range(int(G["b1_n"])))
def test_b1_project_and_filter(i):
    sc = PG.scene(f"b1_{i}")
    H, W = sc["depth"].shape
    coords, depths = IO.project_and_filter_sfm_pts(sc["sfm"], sc["P"], (W, H), sc["mask"])
    assert torch.equal(coords, PG.t(f"b1_{i}_coords"))
    assert torch.equal(depths, PG.t(f"b1_{i}_depths"))


def test_b1_low_confidence_error_branch():
    sc = PG.scene("b1_err")
    with pytest.raises(IO.LowDepthAlignmentConfidenceError):
        IO.project_and_filter_sfm_pts(sc["sfm"], sc["P"], (96, 64), sc["mask"])


@pytest.mark.parametrize("i", range(int(G["b8_n"])))
def test_b8_depth_gradient_mask(i):
    d = PG.scene(f"b8_{i}")["depth"]
    assert torch.equal(IO.depth_gradient_mask(d, float(G[f"b8_{i}_thr"])), PG.t(f"b8_{i}_gradmask"))


def _oracle_chain(i):
    sc = PG.scene(f"b9_{i}")
    aligner, factor, grad_thr, nsfm = PG.b9_cfg(i)
    H, W = sc["depth"].shape
    co, de = IO.project_and_filter_sfm_pts(sc["sfm"], sc["P"], (W, H), sc["mask"])
    torch.manual_seed(int(G[f"b9_{i}_rng_seed"]))
    if aligner == "lstsqrs":
        _, _, aligned = IO.lstsq_align(sc["depth"], co, de)
    else:
        _, _, aligned, _, _ = IO.ransac_align(sc["depth"], co, de, aligner, IO.RansacConfig())
    out_depth, omask = IO.pipeline_align_noseg(aligned, sc["mask"], sc["mask"])
    return sc, co, out_depth, omask, (factor, grad_thr, nsfm)


@pytest.mark.parametrize("i", range(int(G["b9_n"])))
def test_b4_pipeline_noseg(i):
    sc, co, out_depth, omask, _ = _oracle_chain(i)
    H, W = out_depth.shape
    # scale / shift come out of fp32 sums whose order depends on the CPU's vector width: equal on
    # the machine that generated the fixture, within an ulp or two of the map elsewhere
    assert torch.allclose(out_depth, PG.t(f"b9_{i}_aligned"), rtol=2e-6, atol=0)
    assert torch.equal(omask.flatten(), PG.bits(f"b9_{i}_align_mask", H * W))


@pytest.mark.parametrize("i", range(int(G["b9_n"])))
def test_b9_get_pts_from_depth_chain(i):
    """Masks + compaction + unprojection of the oracle on the REFERENCE's aligned depth map
    (identical inputs -> identical integer work -> bit-exact masks and points)."""
    sc, co, _, omask, (factor, grad_thr, nsfm) = _oracle_chain(i)
    out_depth = PG.t(f"b9_{i}_aligned")
    H, W = out_depth.shape
    sub = (IO.static_mask((H, W), factor, omask) if factor != "adaptive"
           else IO.adaptive_mask((H, W, 3), out_depth.clone(), omask))
    pts, mask = IO.assemble_mask_and_unproject(out_depth, omask, sub, sc["K"], sc["c2w"], co,
                                               depth_grad_mask_thresh=grad_thr,
                                               use_num_sfm_points_mask=nsfm)
    assert torch.equal(mask, PG.bits(f"b9_{i}_final_mask", H * W))
    assert torch.equal(pts, PG.t(f"b9_{i}_pts"))
