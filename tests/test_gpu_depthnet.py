"""GPU parity of the MFMA depth network (csrc/depthnet.hip).

Two kinds of checks:
  * kernels against a plain PyTorch fp32 reference of the same op on the same fp16 operands
    (GEMM + epilogues, attention, LayerNorm, im2col convolution, resize, pooling) at the real
    Metric3D shapes (3349 tokens, 1024-d, 16 heads);
  * the assembled encoder / decoder against activations recorded from the REFERENCE's own
    vendored modules (tests/golden/make_depthnet_golden.py) with deterministic weights.
Tolerances (written per test): operands are fp16 with fp32 accumulation, so a GEMM is exact up
to the fp16 rounding of its output (2^-11 relative); whole-network activations are compared at
2e-2 of the reference's maximum and 4e-3 of its mean magnitude. Real weights are a remote
download: "structurally pinned, random weights".
"""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.golden import dn_weights as DW

pytestmark = pytest.mark.gpu
P_ = "3dgs_monocular_depth_init_amd."
G = np.load(Path(__file__).resolve().parent / "golden" / "depthnet_golden.npz")


def mod(name):
    return importlib.import_module(P_ + name)


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.fixture(scope="module")
def N():
    return mod("depth_prediction.predictors.metric3d_net")


def _gemm(N, A, W, bias=None, act=0, gamma=None, residual=None, out32=True):
    lib = mod("_lib")
    M, K = A.shape
    n = W.shape[0]
    o16 = torch.zeros(M, n, dtype=torch.float16, device="cuda")
    o32 = torch.zeros(M, n, dtype=torch.float32, device="cuda") if out32 else None
    lib.call("gsr_dn_gemm", M, n, K, A.data_ptr(), K, W.data_ptr(), lib.ptr(bias), act, lib.ptr(gamma),
             lib.ptr(residual), n, None, 0, o16.data_ptr(), n, lib.ptr(o32), n, 0, _st())
    return o16, o32


# (3349, 4096, 1024), (3900, 3900, 576) and (40964, 256, 2304) take the 256x256 eight-phase core (224 / 256 /
# 161 workgroups, ragged edges in M and N, odd and even K-tile counts), the others the 128-row tiles in
# both widths
@pytest.mark.parametrize("M,n,K", [(3349, 1152, 384), (3349, 1024, 4096), (1, 384, 1920), (200, 258, 512),
                                   (40964, 256, 2304), (130, 6, 448), (3349, 4096, 1024), (3900, 3900, 576)])
def test_gemm_vs_torch(N, M, n, K):
    g = torch.Generator().manual_seed(M + n + K)
    A = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    W = (torch.randn(n, K, generator=g) / K ** 0.5).half().cuda()
    bias = torch.randn(n, generator=g).cuda()
    ref = A.float() @ W.float().T + bias
    o16, o32 = _gemm(N, A, W, bias)
    scale = float(ref.abs().max())
    assert float((o32 - ref).abs().max()) <= 2e-5 * scale * max(1.0, K / 512) ** 0.5 + 1e-5
    assert float((o16.float() - ref).abs().max()) <= 1.5e-3 * scale          # fp16 output rounding


def test_gemm_epilogues_vs_torch(N):
    g = torch.Generator().manual_seed(7)
    M, n, K = 300, 192, 256
    A = torch.randn(M, K, generator=g).half().cuda()
    W = (torch.randn(n, K, generator=g) / 16).half().cuda()
    bias, gamma = torch.randn(n, generator=g).cuda(), torch.rand(n, generator=g).cuda()
    res = torch.randn(M, n, generator=g).cuda()
    z = A.float() @ W.float().T + bias
    for act, fn in ((1, lambda t: F.gelu(t)), (2, torch.relu), (3, torch.sigmoid), (4, torch.tanh)):
        _, o32 = _gemm(N, A, W, bias, act=act)
        assert torch.allclose(o32, fn(z), rtol=2e-5, atol=2e-5), act
    r = res.clone()
    lib = mod("_lib")
    lib.call("gsr_dn_gemm", M, n, K, A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), 0, gamma.data_ptr(),
             r.data_ptr(), n, None, 0, None, 0, r.data_ptr(), n, 0, _st())    # x += gamma * (A W^T + b), in place
    assert torch.allclose(r, res + gamma * z, rtol=2e-5, atol=2e-5)


def test_gemm_eight_phase_core_epilogues_vs_torch(N):
    """The 256x256 core (grid of 12 x 12 tiles) with every epilogue input: rows through LDS where a wave's
    64 columns are inside N, element-wise at the ragged right edge; GELU, LayerScale, fp32 residual in
    place, fp16 residual, fp16 output with zero channels up to the padded width; 5 K-tiles (odd)."""
    lib = mod("_lib")
    g = torch.Generator().manual_seed(11)
    M, n, K = 3000, 2968, 320
    A = torch.randn(M, K, generator=g).half().cuda()
    W = (torch.randn(n, K, generator=g) / K ** 0.5).half().cuda()
    bias, gamma = torch.randn(n, generator=g).cuda(), torch.rand(n, generator=g).cuda()
    res = torch.randn(M, n, generator=g).cuda()
    z = A.float() @ W.float().T + bias
    o16, o32 = _gemm(N, A, W, bias, act=1)
    assert torch.allclose(o32, F.gelu(z), rtol=2e-5, atol=3e-5)
    assert float((o16.float() - F.gelu(z)).abs().max()) <= 1.5e-3 * float(z.abs().max())
    r = res.clone()
    lib.call("gsr_dn_gemm", M, n, K, A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), 0, gamma.data_ptr(),
             r.data_ptr(), n, None, 0, None, 0, r.data_ptr(), n, 0, _st())    # x += gamma * (A W^T + b), in place
    assert torch.allclose(r, res + gamma * z, rtol=2e-5, atol=3e-5)
    ld = 3008                                                                 # the map's stored width (n -> 64)
    r16 = torch.randn(M, ld, generator=g).half().cuda()
    out = torch.full((M, ld), 7.0, dtype=torch.float16, device="cuda")
    lib.call("gsr_dn_gemm", M, n, K, A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), 2, None, None, 0,
             r16.data_ptr(), ld, out.data_ptr(), ld, None, 0, ld, _st())
    ref = torch.relu(z) + r16[:, :n].float()
    assert float((out[:, :n].float() - ref).abs().max()) <= 1.5e-3 * float(ref.abs().max())
    assert float(out[:, n:].abs().max()) == 0.0                               # zero channels written


def test_conv3_head_vs_torch(N):
    """gsr_dn_conv3_head: flow[:, o:o+n] += bias + conv3x3(map)[:, :n] for the flow head's 2- and
    4-channel outputs, against F.conv2d in fp32 on the same fp16 operands."""
    lib = mod("_lib")
    g = torch.Generator().manual_seed(3)
    H, Wd, C, ld = 37, 53, 128, 256
    full = torch.randn(H * Wd, ld, generator=g).half().cuda()
    for c0, o0, n in ((0, 0, 2), (128, 2, 4)):
        w = (torch.randn(n, C, 3, 3, generator=g) / (9 * C) ** 0.5).half()
        b = torch.randn(n, generator=g).cuda()
        rows = w.permute(0, 2, 3, 1).reshape(n, 9 * C).contiguous().cuda()     # column = (ky * 3 + kx) * C + c
        flow = torch.randn(H * Wd, 6, generator=g).cuda()
        before = flow.clone()
        part = full[:, c0:]
        lib.call("gsr_dn_conv3_head", H, Wd, C, part.data_ptr(), ld, n, rows.data_ptr(), 9 * C, b.data_ptr(),
                 flow[:, o0:].data_ptr(), 6, _st())
        x = full[:, c0:c0 + C].float().reshape(1, H, Wd, C).permute(0, 3, 1, 2)
        ref = F.conv2d(x, w.float().cuda(), b, padding=1)[0].permute(1, 2, 0).reshape(H * Wd, n)
        assert torch.allclose(flow[:, o0:o0 + n], before[:, o0:o0 + n] + ref, rtol=1e-4, atol=1e-4)
        keep = [c for c in range(6) if not (o0 <= c < o0 + n)]
        assert torch.equal(flow[:, keep], before[:, keep])                      # the other columns untouched


# key split of the workgroup (gsr_dn_attention): (101, 2) and (1500, 64) run unsplit, (3349, 16) in two groups,
# (3349, 6), (777, 3) and (150, 1) in four (25 and 5 key tiles: groups with a surplus tile at the end)
@pytest.mark.parametrize("n_tok,heads", [(101, 2), (3349, 6), (3349, 16), (777, 3), (150, 1), (1500, 64)])
def test_attention_vs_torch(N, n_tok, heads):
    lib = mod("_lib")
    D = heads * 64
    g = torch.Generator().manual_seed(n_tok + heads)
    qkv = (torch.randn(n_tok, 3 * D, generator=g) * 1.5).half().cuda()
    n_pad = (n_tok + 63) // 64 * 64
    vt = torch.empty(heads * 64 * n_pad, dtype=torch.float16, device="cuda")
    out = torch.zeros(n_tok, D, dtype=torch.float16, device="cuda")
    lib.call("gsr_dn_attention", n_tok, n_pad, heads, qkv.data_ptr(), 3 * D, vt.data_ptr(), 0.125, out.data_ptr(),
             D, _st())
    q, k, v = (t.reshape(n_tok, heads, 64).permute(1, 0, 2).float() for t in qkv.split(D, dim=1))
    ref = (torch.softmax(q @ k.transpose(1, 2) * 0.125, -1) @ v).permute(1, 0, 2).reshape(n_tok, D)
    err = (out.float() - ref).abs()
    # P is rounded to fp16 before the PV product: ~5e-4 relative per weight, averaged over keys
    assert float(err.max()) <= 4e-3 * float(ref.abs().max()) and float(err.mean()) <= 4e-4 * float(ref.abs().mean()) + 1e-5


def test_layernorm_conv_resize_pool_vs_torch(N):
    lib = mod("_lib")
    g = torch.Generator().manual_seed(3)
    # LayerNorm fp32 in
    x = torch.randn(77, 384, generator=g).cuda() * 3 + 1
    w, b = torch.rand(384, generator=g).cuda() + 0.5, torch.randn(384, generator=g).cuda()
    o = torch.zeros(77, 384, dtype=torch.float16, device="cuda")
    lib.call("gsr_dn_layernorm", 77, 384, x.data_ptr(), 384, 0, w.data_ptr(), b.data_ptr(), 1e-6, o.data_ptr(), 384,
             None, 0, 0, _st())
    assert torch.allclose(o.float(), F.layer_norm(x, (384,), w, b, 1e-6), rtol=2e-3, atol=2e-3)
    # 3x3 convolution through im2col + GEMM, stride 1 and C not a multiple of 8
    net = N.Metric3DNet.__new__(N.Metric3DNet)
    net.dev, net._scratch = torch.device("cuda"), {}
    # (without input ReLU: C % 64 == 0 takes the implicit-GEMM path as is, 32 <= C with the map's zero
    # channels up to the next multiple of 64 and zero weight planes for them -- 48, 96, 262 here;
    # fewer than 32 channels, or an input ReLU, go through im2col rows; 1x1: the map's rows are the
    # GEMM's rows, K padding = the zero channels)
    for C, Co, H, W, ks, relu_in in ((38, 16, 28, 42, 3, True), (64, 96, 44, 76, 3, True), (64, 96, 44, 76, 3, False),
                                     (128, 258, 31, 45, 3, False), (192, 48, 17, 23, 1, False),
                                     (48, 48, 44, 76, 3, False), (96, 32, 31, 45, 3, False), (262, 64, 23, 31, 3, False),
                                     (6, 40, 28, 42, 3, False), (48, 96, 17, 23, 1, False), (102, 7, 17, 23, 1, False)):
        xin = torch.randn(1, C, H, W, generator=g)
        wt = torch.randn(Co, C, ks, ks, generator=g) / (ks * ks * C) ** 0.5
        bs = torch.randn(Co, generator=g)
        m = N.Map(H, W, C, "cuda")
        m.t[:, :C] = xin[0].permute(1, 2, 0).reshape(-1, C).half().cuda()
        res = N.Map(H, W, Co, "cuda")
        res.t.zero_()
        rin = torch.randn(H * W, Co, generator=g).half()
        res.t[:, :Co] = rin.cuda()
        out = net.conv(m, N._conv_lin(wt, bs, "cuda", pad_cin=not relu_in), ks, N.Map(H, W, Co, "cuda"),
                       act=N.ACT_RELU, relu_in=relu_in, residual=res)
        a = F.relu(xin.half().float()) if relu_in else xin.half().float()
        ref = F.relu(F.conv2d(a, wt.half().float(), bs, padding=ks // 2))[0].permute(1, 2, 0) + rin.float().view(H, W, Co)
        err = float((out.t[:, :Co].float().cpu().view(H, W, Co) - ref).abs().max())
        assert err <= 3e-3 * float(ref.abs().max()), (C, Co, ks, relu_in, err)
        # the same launch into a map allocated WITHOUT a fill: the epilogue writes the zero channels
        raw = N.Map(H, W, Co, "cuda", zero=False)
        raw.t.fill_(7.0)
        out2 = net.conv(m, N._conv_lin(wt, bs, "cuda", pad_cin=not relu_in), ks, raw, act=N.ACT_RELU, relu_in=relu_in,
                        residual=res)
        assert torch.equal(out2.t[:, :Co], out.t[:, :Co]) and out2.zpad and not out2.pad_pending
        assert out2.ld == Co or float(out2.t[:, Co:].abs().max()) == 0.0
    # resize modes and pooling
    xin = torch.randn(1, 16, 11, 13, generator=g).half()
    m = N.Map(11, 13, 16, "cuda")
    m.t.copy_(xin[0].permute(1, 2, 0).reshape(-1, 16).cuda())
    cases = [(N.RESIZE_BILINEAR_AC, (22, 26), dict(mode="bilinear", align_corners=True)),
             (N.RESIZE_BILINEAR_AC, (19, 22), dict(mode="bilinear", align_corners=True)),
             (N.RESIZE_BILINEAR, (30, 17), dict(mode="bilinear", align_corners=False))]
    for mode, (Ho, Wo), kw in cases:
        out = net.resize(m, Ho, Wo, mode)
        ref = F.interpolate(xin.float(), size=(Ho, Wo), **kw)[0].permute(1, 2, 0)
        assert torch.allclose(out.t.float().cpu().view(Ho, Wo, 16), ref, rtol=2e-3, atol=2e-3), (mode, Ho, Wo)
    m2 = N.Map(12, 14, 16, "cuda")
    x2 = torch.randn(1, 16, 12, 14, generator=g).half()
    m2.t.copy_(x2[0].permute(1, 2, 0).reshape(-1, 16).cuda())
    out = net.resize(m2, 42, 49, N.RESIZE_NEAREST)
    ref = F.interpolate(x2.float(), scale_factor=3.5, mode="nearest")[0].permute(1, 2, 0)
    assert torch.equal(out.t.float().cpu().view(42, 49, 16), ref)
    out = net._pool2x(m)
    ref = F.avg_pool2d(xin.float(), 3, stride=2, padding=1)[0].permute(1, 2, 0)
    assert torch.allclose(out.t.float().cpu().view(*ref.shape), ref, rtol=2e-3, atol=2e-3)


def _close(got, ref, max_frac=3e-3, mean_frac=3e-3, what=""):
    """|got - ref| <= max_frac * max|ref| everywhere and mean|got - ref| <= mean_frac * mean|ref|.
    Defaults = 3x the errors round 2 recorded for these fixtures (profiles/parity_r02.json: 4e-4 ..
    1e-3 of the maximum, ~5e-4 .. 1e-3 of the mean): fp16 storage of activations and weights against
    the reference's fp32 modules."""
    got, ref = got.float().cpu(), torch.as_tensor(np.asarray(ref, dtype=np.float32))
    err = (got - ref).abs()
    from tests import parity_log
    parity_log.record("depthnet", what=what, max_abs_err=float(err.max()), mean_abs_err=float(err.mean()),
                      ref_max=float(ref.abs().max()), ref_mean=float(ref.abs().mean()))
    assert float(err.max()) <= max_frac * float(ref.abs().max()), (what, float(err.max()), float(ref.abs().max()))
    assert float(err.mean()) <= mean_frac * float(ref.abs().mean()), (what, float(err.mean()), float(ref.abs().mean()))


def _vit_shapes(D, depth, swiglu_hidden=0):
    s = {"cls_token": (1, 1, D), "pos_embed": (1, 1370, D), "register_tokens": (1, 4, D), "mask_token": (1, D),
         "patch_embed.proj.weight": (D, 3, 14, 14), "patch_embed.proj.bias": (D,), "norm.weight": (D,), "norm.bias": (D,)}
    for i in range(depth):
        p = f"blocks.0.{i}."
        s.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,), p + "attn.qkv.weight": (3 * D, D),
                  p + "attn.qkv.bias": (3 * D,), p + "attn.proj.weight": (D, D), p + "attn.proj.bias": (D,),
                  p + "ls1.gamma": (D,), p + "norm2.weight": (D,), p + "norm2.bias": (D,), p + "ls2.gamma": (D,)})
        if swiglu_hidden:
            h = swiglu_hidden
            s.update({p + "mlp.w12.weight": (2 * h, D), p + "mlp.w12.bias": (2 * h,), p + "mlp.w3.weight": (D, h),
                      p + "mlp.w3.bias": (D,)})
        else:
            s.update({p + "mlp.fc1.weight": (4 * D, D), p + "mlp.fc1.bias": (4 * D,),
                      p + "mlp.fc2.weight": (D, 4 * D), p + "mlp.fc2.bias": (D,)})
    return s


def _dec_shapes(D, fc, dc, hid):
    """Parameter names / shapes of RAFTDepthNormalDPT5 (the reference's own state-dict keys)."""
    s = {}
    for i in range(4):
        p = f"token2feature.read_{i}.readoper."
        s.update({p + "project_patch.weight": (D, D), p + "project_patch.bias": (D,), p + "project_learn.weight": (D, 5 * D)})
    s.update({"token2feature.read_1.sample.weight": (D, fc[1], 2, 2), "token2feature.read_1.sample.bias": (fc[1],),
              "token2feature.read_0.sample.0.weight": (fc[0], D, 1, 1), "token2feature.read_0.sample.0.bias": (fc[0],)})
    for name, cin, cout, br in (("upconv_3", dc[4], dc[3], False), ("upconv_2", dc[3], dc[2], True),
                                ("upconv_1", dc[2], dc[1] + 2, True)):
        p = f"decoder_mono.{name}."
        for w in ("way_trunk",) + (("way_branch",) if br else ()):
            for c in ("conv1", "conv2"):
                s.update({p + f"{w}.{c}.weight": (cin, cin, 3, 3), p + f"{w}.{c}.bias": (cin,)})
        s.update({p + "out_conv.weight": (cout, cin, 1, 1), p + "out_conv.bias": (cout,)})
    s.update({"depth_regressor.0.weight": (256, dc[1], 3, 3), "depth_regressor.0.bias": (256,),
              "depth_regressor.2.weight": (256, 256, 1, 1), "depth_regressor.2.bias": (256,),
              "normal_predictor.0.weight": (128, dc[1], 3, 3), "normal_predictor.0.bias": (128,),
              "normal_predictor.2.weight": (128, 128, 1, 1), "normal_predictor.2.bias": (128,),
              "normal_predictor.4.weight": (128, 128, 1, 1), "normal_predictor.4.bias": (128,),
              "normal_predictor.6.weight": (3, 128, 1, 1), "normal_predictor.6.bias": (3,)})
    for lvl, cin, h in (("04", fc[0], hid[0]), ("08", fc[1], hid[1]), ("16", fc[2], hid[2])):
        for k in (0, 1):
            p = f"context_feature_encoder.outputs{lvl}.{k}."
            s.update({p + "0.conv1.weight": (h, cin, 3, 3), p + "0.conv1.bias": (h,), p + "0.conv2.weight": (h, h, 3, 3),
                      p + "0.conv2.bias": (h,), p + "1.weight": (h, h, 3, 3), p + "1.bias": (h,)})
            for n in ("norm1", "norm2", "norm3"):
                s.update({p + f"0.{n}.weight": (h,), p + f"0.{n}.bias": (h,)})
            s.update({p + "0.downsample.0.weight": (h, cin, 1, 1), p + "0.downsample.0.bias": (h,),
                      p + "0.downsample.1.weight": (h,), p + "0.downsample.1.bias": (h,)})
    for i in range(3):
        s.update({f"context_zqr_convs.{i}.weight": (3 * hid[i], hid[i], 3, 3), f"context_zqr_convs.{i}.bias": (3 * hid[i],)})
    for g, h, cin in (("gru08", hid[2], 6 + hid[1]), ("gru16", hid[1], hid[0] + hid[2]), ("gru32", hid[0], hid[1])):
        for c in ("convz", "convr", "convq"):
            s.update({f"update_block.{g}.{c}.weight": (h, h + cin, 3, 3), f"update_block.{g}.{c}.bias": (h,)})
    h = hid[2]
    s.update({"update_block.flow_head.conv1d.weight": (h, h, 3, 3), "update_block.flow_head.conv1d.bias": (h,),
              "update_block.flow_head.conv2d.weight": (2, h, 3, 3), "update_block.flow_head.conv2d.bias": (2,),
              "update_block.flow_head.conv1n.weight": (h, h, 3, 3), "update_block.flow_head.conv1n.bias": (h,),
              "update_block.flow_head.conv2n.weight": (4, h, 3, 3), "update_block.flow_head.conv2n.bias": (4,),
              "update_block.mask.0.weight": (h, h, 3, 3), "update_block.mask.0.bias": (h,),
              "update_block.mask.2.weight": (144, h, 1, 1), "update_block.mask.2.bias": (144,)})
    return s


SMALL_CFG = dict(embed_dim=128, depth=2, heads=2, feature_channels=[32, 64, 128, 256],
                 decoder_channels=[16, 32, 64, 128, 128], hidden=[16, 16, 16, 16], iters=3)


def _state(cfg, depth_gain=None):
    sd = {"encoder." + k: v for k, v in DW.fill(_vit_shapes(cfg["embed_dim"], cfg["depth"])).items()}
    sd.update({"decoder." + k: v for k, v in DW.fill(_dec_shapes(cfg["embed_dim"], cfg["feature_channels"],
                                                                 cfg["decoder_channels"], cfg["hidden"]),
                                                     depth_gain=depth_gain).items()})
    return sd


@pytest.mark.parametrize("D,heads,key", [(128, 2, "vit128_tokens"), (1024, 16, "vit1024_tokens")])
def test_vit_encoder_vs_reference_golden(N, D, heads, key):
    """DinoVisionTransformer.forward_features (ViT_DINO_reg.py:962-1004) on the deterministic
    weights: (a) a reduced 128-d model, (b) vit_large_reg's block shapes (1024-d, 16 heads)."""
    cfg = dict(SMALL_CFG, embed_dim=D, heads=heads)
    sd = {"encoder." + k: v for k, v in DW.fill(_vit_shapes(D, 2)).items()}
    net = N.Metric3DNet.__new__(N.Metric3DNet)
    net.cfg, net.dev, net.H, net.W, net._scratch = cfg, torch.device("cuda"), 112, 168, {}
    net.gh, net.gw, net.D, net.heads, net.depth = 8, 12, D, heads, 2
    net.n_tok = 1 + 4 + 96
    net._prep_encoder({k[len("encoder."):]: v for k, v in sd.items()})
    tokens = net.encode(DW.image(112, 168))
    assert tokens.shape == (101, D)
    _close(tokens, G[key], what=key)


def test_decoder_vs_reference_golden(N):
    """RAFTDepthNormalDPT5.forward (decoder :890-1004) with a reduced configuration: the token
    read-out, the DPT fusion and the final depth / confidence / normal maps."""
    net = N.Metric3DNet(_state(SMALL_CFG), device="cuda", input_size=(112, 168), config=SMALL_CFG)
    tokens = torch.from_numpy(G["vit128_tokens"]).half().cuda()          # the reference encoder's output
    depth, conf, normal, inter = net.decode(tokens, return_intermediates=True)
    for i, m in enumerate(inter["encfeat"]):
        ref = G[f"dec_encfeat{i}"].astype(np.float32)
        _close(m.t[:, :m.C].reshape(m.H, m.W, m.C), ref, what=f"encfeat{i}")
    rf = inter["ref_feat"]
    _close(rf.t[:, :rf.C].reshape(rf.H, rf.W, rf.C), G["dec_ref_feat"].astype(np.float32), what="ref_feat")
    _close(inter["depth_init"].view(rf.H, rf.W, 6), G["dec_depth_init"], what="depth_init")
    for i in range(3):
        m, c = inter["nets"][i], inter["ctxs"][i]
        # (tanh outputs in [-1, 1]: the largest error, 4.2e-3, is an fp16 ulp pair near saturation)
        _close(m.t[:, :m.C].reshape(m.H, m.W, m.C), G[f"dec_net{i}"].astype(np.float32), max_frac=1.3e-2, what=f"net{i}")
        _close(c.t[:, :c.C].reshape(c.H, c.W, c.C), G[f"dec_ctx{i}"].astype(np.float32), what=f"ctx{i}")
    for i, dl in enumerate(inter["deltas"]):
        _close(dl.view(rf.H, rf.W, 6), G[f"dec_delta{i}"], max_frac=5e-3, mean_frac=5e-3, what=f"delta_flow{i}")
    assert depth.shape == (1, 1, 112, 168) and normal.shape == (1, 4, 112, 168)
    _close(depth[0, 0], G["dec_depth"], max_frac=5e-3, what="depth")
    _close(conf[0, 0], G["dec_conf"], max_frac=5e-3, mean_frac=5e-3, what="confidence")
    _close(normal[0], G["dec_normal"], max_frac=6e-3, mean_frac=5e-3, what="normal")


def test_end_to_end_small_and_predictor(N):
    """Encoder + decoder chained, through Metric3d.predict_depth (metric3d.py:38-139)."""
    M = mod("depth_prediction.predictors.metric3d")
    ifc = mod("depth_prediction.predictors.depth_predictor_interface")
    net = N.Metric3DNet(_state(SMALL_CFG), device="cuda", input_size=(112, 168), config=SMALL_CFG)
    d, c, o = net.inference({"input": DW.image(112, 168)})       # eager pass + graph capture + replay
    _close(d[0, 0], G["dec_depth"], max_frac=5e-3, mean_frac=4e-3, what="e2e depth")
    assert torch.isfinite(o["prediction_normal"]).all()
    # graph replay on a second image == eager on that image
    img2 = DW.image(112, 168).flip(-1).contiguous()
    d2, _, o2 = net.inference({"input": img2})
    net.use_graph = False
    d3, _, o3 = net.inference({"input": img2})
    assert torch.equal(d2, d3) and torch.equal(o2["prediction_normal"], o3["prediction_normal"])
    assert not torch.equal(d2, d)
    # and through the predictor of the reference's interface (metric3d.py:38-139)
    pred = M.Metric3d(None, "cuda", model=net, backbone="vits")
    assert pred.name == "Metric3d_vits"


def test_full_size_vits_runs(N):
    """c3's network at its real size: ViT-S/14-reg + RAFT-DPT at 616x1064 (3349 tokens)."""
    cfg = N.CONFIGS["vits"]
    net = N.Metric3DNet(_state(cfg), backbone="vits", device="cuda")
    d, c, o = net.inference({"input": DW.image(616, 1064)})
    torch.cuda.synchronize()
    assert d.shape == (1, 1, 616, 1064) and o["prediction_normal"].shape == (1, 4, 616, 1064)
    assert torch.isfinite(d).all() and torch.isfinite(c).all() and torch.isfinite(o["prediction_normal"]).all()
    assert float(d.min()) >= 0.1 and float(d.max()) <= 200.0


FG = np.load(Path(__file__).resolve().parent / "golden" / "depthnet_full_golden.npz")


@pytest.mark.parametrize("name", ["vits", "vitl"])
def test_full_depth_networks_vs_reference_golden(N, name):
    """The two networks the reference loads (metric3d.py:20-25) at their FULL configuration --
    vit_small_reg: 384-d, 6 heads, 12 blocks, 4 ConvGRU iterations; vit_large_reg: 1024-d, 16 heads,
    24 blocks, 8 iterations; decoder channel plans of dino_vit_*_reg.dpt_raft.py -- against
    activations recorded from the reference's vendored modules (tests/golden/make_depthnet_golden.py),
    on the 112x168 input the CPU generator affords. 24 residual blocks and 8 recurrent refinements are
    where fp16 drift would accumulate; the 2-block fixtures cannot show it."""
    cfg = N.CONFIGS[name]
    net = N.Metric3DNet(_state(cfg, depth_gain=DW.FULL_DEPTH_GAIN), backbone=name, device="cuda", input_size=(112, 168))
    tokens = net.encode(DW.image(112, 168))
    assert tokens.shape == (101, cfg["embed_dim"])
    _close(tokens, FG[f"{name}_tokens"].astype(np.float32), max_frac=6e-3, mean_frac=4e-3, what=f"{name} tokens")
    # the decoder on the REFERENCE encoder's tokens (so that the two halves are judged separately)
    ref_tokens = torch.from_numpy(FG[f"{name}_tokens"]).half().cuda()
    depth, conf, normal, inter = net.decode(ref_tokens, return_intermediates=True)
    assert len(inter["deltas"]) == cfg["iters"]
    _close(inter["deltas"][-1].view(28, 42, 6), FG[f"{name}_delta_last"], max_frac=1e-2, mean_frac=6e-3,
           what=f"{name} delta_flow[{cfg['iters'] - 1}]")
    _close(conf[0, 0], FG[f"{name}_conf"], max_frac=1e-2, mean_frac=6e-3, what=f"{name} confidence")
    _close(normal[0], FG[f"{name}_normal"], max_frac=1e-2, mean_frac=6e-3, what=f"{name} normal")
    # depth (round 4): the fixture's update head keeps BOTH networks' depth inside the (0.1, 200) clamp on every
    # pixel (dn_weights.FULL_DEPTH_GAIN; round 3's ViT-L map was 99.5 % saturated and never compared), so the
    # whole map -- accumulated updates, convex 4x upsampling, 100 * flow + 200, clamp -- is compared with a
    # MAXIMUM bound. depth = 100 * flow + 200: an absolute flow error of 1e-2 of max|delta| (the bound on
    # delta_flow above) would be 1 depth unit; recorded (profiles/parity_r04.json): 0.03-0.04 units maximum =
    # 2.6e-4 of the map's maximum (5e-4 relative per pixel), mean 1e-4; the bounds are ~3x that.
    ref_d = torch.from_numpy(FG[f"{name}_depth"])
    got_d = depth[0, 0].float().cpu()
    assert float(ref_d.min()) > 0.2 and float(ref_d.max()) < 199.0          # nothing of the reference in the clamp
    assert float(got_d.min()) > 0.2 and float(got_d.max()) < 199.0
    rel = (got_d - ref_d).abs() / ref_d
    from tests import parity_log
    parity_log.record("depthnet", what=f"{name} depth (whole map, unsaturated)", pixels=rel.numel(),
                      max_rel_err=float(rel.max()), p999_rel_err=float(torch.quantile(rel, 0.999)),
                      mean_rel_err=float(rel.mean()), ref_min=float(ref_d.min()), ref_max=float(ref_d.max()))
    _close(got_d, ref_d, max_frac=1e-3, mean_frac=4e-4, what=f"{name} depth")
    # and the whole chain, encoder into decoder
    d2, c2, o2 = net.inference({"input": DW.image(112, 168)})
    _close(c2[0, 0], FG[f"{name}_conf"], max_frac=2e-2, mean_frac=1e-2, what=f"{name} e2e confidence")
    _close(d2[0, 0], ref_d, max_frac=1.5e-3, mean_frac=8e-4, what=f"{name} e2e depth")


def test_vit_giant_swiglu_blocks_vs_reference_golden(N):
    """vit_giant2_reg's block -- 1536-d, 24 heads, SwiGLU FFN with hidden 4096 (ViT_DINO_reg.py:1246-1260,
    300-345) -- four blocks of it, against tokens recorded from the reference's module."""
    D, heads, depth, hidden = 1536, 24, 4, 4096
    cfg = dict(N.CONFIGS["vitg"], depth=depth)
    sd = {"encoder." + k: v for k, v in DW.fill(_vit_shapes(D, depth, swiglu_hidden=hidden)).items()}
    net = N.Metric3DNet.__new__(N.Metric3DNet)
    net.cfg, net.dev, net.H, net.W, net._scratch = cfg, torch.device("cuda"), 112, 168, {}
    net.gh, net.gw, net.D, net.heads, net.depth = 8, 12, D, heads, depth
    net.n_tok = 1 + 4 + 96
    net._prep_encoder({k[len("encoder."):]: v for k, v in sd.items()})
    assert "w12" in net.blocks[0]
    tokens = net.encode(DW.image(112, 168))
    assert tokens.shape == (101, D)
    _close(tokens, FG["vitg4_tokens"].astype(np.float32), max_frac=6e-3, mean_frac=4e-3, what="vitg4 tokens")


def test_full_size_vitl_runs(N):
    """c5's network at its real size: ViT-L/14-reg (24 blocks, 1024-d) + RAFT-DPT (8 iterations) at
    616x1064 (3349 tokens)."""
    cfg = N.CONFIGS["vitl"]
    net = N.Metric3DNet(_state(cfg), backbone="vitl", device="cuda")
    d, c, o = net.inference({"input": DW.image(616, 1064)})
    torch.cuda.synchronize()
    assert d.shape == (1, 1, 616, 1064) and o["prediction_normal"].shape == (1, 4, 616, 1064)
    assert torch.isfinite(d).all() and torch.isfinite(c).all() and torch.isfinite(o["prediction_normal"]).all()
    assert float(d.min()) >= 0.1 and float(d.max()) <= 200.0


def test_gru_virtual_concatenation_equals_copied_input(N):
    """`gsr_dn_conv_gemm2` (the ConvGRU's [h | x] and [r*h | x] read as virtual concatenations, persistent
    buffers) against the copy of h into the concatenated input map: the same values enter the same kernels,
    so the decoder outputs are IDENTICAL."""
    cfg = N.CONFIGS["vitl"]
    net = N.Metric3DNet(_state(cfg), backbone="vitl", device="cuda", input_size=(112, 168), use_graph=False)
    tok = torch.from_numpy(FG["vitl_tokens"]).half().cuda()
    assert net.concat_free_gru
    d1, c1, n1, i1 = net.decode(tok, return_intermediates=True)
    assert len(net._gru_bufs) == 3                       # the path ran at all three levels
    d1b, c1b, n1b = net.decode(tok)                      # and again on the now persistent buffers
    net.concat_free_gru = False
    d0, c0, n0, i0 = net.decode(tok, return_intermediates=True)
    for a, b in zip(i1["deltas"], i0["deltas"]):
        assert torch.equal(a, b)
    assert torch.equal(d1, d0) and torch.equal(c1, c0) and torch.equal(n1, n0)
    assert torch.equal(d1b, d0) and torch.equal(c1b, c0)


@pytest.mark.parametrize("name", ["vits", "vitl"])
def test_batched_inference_equals_single_images(N, name):
    """inference({"input": [B,3,H,W]}): the encoder over all B images at once (GEMMs over B * n_tok rows,
    attention per image), the B decoders as parallel branches of one HIP graph on their own streams and scratch
    maps. Every image of the batch must come out as it does alone (same kernels; only a GEMM's tile shape may
    change with the row count, i.e. fp16 rounding), graph replay == eager, and branches must not share state."""
    cfg = N.CONFIGS[name]
    sd = _state(cfg, depth_gain=DW.FULL_DEPTH_GAIN)
    net = N.Metric3DNet(sd, backbone=name, device="cuda", input_size=(112, 168))
    base = DW.image(112, 168)
    imgs = torch.cat([base, base.flip(-1), base.flip(-2) * 0.5], 0).contiguous()          # three different images
    singles = [tuple(t.clone() for t in net.inference({"input": imgs[i:i + 1]})[:2]) + (net.inference({"input": imgs[i:i + 1]})[2]["prediction_normal"].clone(),)
               for i in range(3)]
    d, c, o = net.inference({"input": imgs})                      # eager pass + capture + replay
    d2, c2, o2 = net.inference({"input": imgs})                   # replay again: nothing left behind by the first
    assert d.shape == (3, 1, 112, 168) and o["prediction_normal"].shape == (3, 4, 112, 168)
    assert torch.equal(d, d2) and torch.equal(c, c2) and torch.equal(o["prediction_normal"], o2["prediction_normal"])
    net_e = N.Metric3DNet(sd, backbone=name, device="cuda", input_size=(112, 168), use_graph=False)
    de, ce, oe = net_e.inference({"input": imgs})
    assert torch.equal(d, de) and torch.equal(c, ce)
    for i, (ds, cs, ns) in enumerate(singles):
        _close(d[i, 0], ds[0, 0].cpu().numpy(), max_frac=2e-3, mean_frac=5e-4, what=f"{name} batch depth {i}")
        _close(c[i, 0], cs[0, 0].cpu().numpy(), max_frac=5e-3, mean_frac=2e-3, what=f"{name} batch confidence {i}")
        _close(o["prediction_normal"][i], ns[0].cpu().numpy(), max_frac=5e-3, mean_frac=2e-3, what=f"{name} batch normal {i}")
    assert float((d[0] - d[1]).abs().max()) > 0.1                 # (the images do differ)
    # through the predictor: predict_depths == [predict_depth]
    M = mod("depth_prediction.predictors.metric3d")
    ifc = mod("depth_prediction.predictors.depth_predictor_interface")
    net2 = N.Metric3DNet(_state(N.CONFIGS["vits"]), backbone="vits", device="cuda")
    pred = M.Metric3d(None, "cuda", model=net2, backbone="vits")
    g = torch.Generator().manual_seed(1)
    photos = [torch.rand(240, 320, 3, generator=g).cuda(), torch.rand(200, 360, 3, generator=g).cuda(),
              torch.rand(240, 320, 3, generator=g).cuda()]
    K = torch.tensor([[300.0, 0, 160], [0, 300.0, 120], [0, 0, 1]])
    intr = [ifc.CameraIntrinsics(K)] * 3
    many = pred.predict_depths(photos, intr)
    for ph, pm in zip(photos, many):
        one = pred.predict_depth(ph, intr[0])
        assert pm.depth.shape == ph.shape[:2]
        assert float((pm.depth - one.depth).abs().max()) <= 2e-3 * float(one.depth.abs().max())
