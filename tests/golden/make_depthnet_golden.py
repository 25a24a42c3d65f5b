"""Generate tests/golden/depthnet_golden.npz by running the REFERENCE's own vendored Metric3D
modules (architecture spec of SURVEY.md row B10):

  gs_init_compare/third_party/metric3d/mono/model/backbones/ViT_DINO_reg.py
      DinoVisionTransformer (:755), vit_large_reg block shapes (:1227)
  gs_init_compare/third_party/metric3d/mono/model/decode_heads/RAFTDepthNormalDPTDecoder5.py
      RAFTDepthNormalDPT5 (:736-1035)

Run only in the build container: python tests/golden/make_depthnet_golden.py

Both files import with plain torch (xformers is optional there and absent -> their own
fallback attention). They are loaded by file path; nothing is stubbed. Weights and inputs are
the deterministic tensors of tests/golden/dn_weights.py (the real weights are a torch.hub
download, unavailable offline), so the fixture holds OUTPUT activations only:
"structurally pinned, random weights". One buffer the decoder would create with
device="cuda" (get_bins, :797-802) is registered from here with the same formula on the CPU.
"""
import importlib.util
import math
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import dn_weights as DW  # noqa: E402

BASE = "/root/reference/gs_init_compare/third_party/metric3d/mono/model"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


V = _load("ref_vit", BASE + "/backbones/ViT_DINO_reg.py")
D = _load("ref_dec", BASE + "/decode_heads/RAFTDepthNormalDPTDecoder5.py")


class NS(dict):
    __getattr__ = dict.__getitem__


def vit(embed_dim, heads, depth, ffn_layer="mlp"):
    m = V.DinoVisionTransformer(img_size=518, patch_size=14, embed_dim=embed_dim, depth=depth, num_heads=heads,
                                mlp_ratio=4, block_fn=V.partial(V.Block, attn_class=V.MemEffAttention),
                                num_register_tokens=4, ffn_layer=ffn_layer).eval()
    m.load_state_dict(DW.fill(m))
    return m


out = {}
H, W = 112, 168
img = DW.image(H, W)
torch.set_num_threads(8)
FULL_ONLY = "--full-only" in sys.argv     # regenerate depthnet_full_golden.npz alone
bins = torch.exp(torch.linspace(math.log(0.1), math.log(200), 256)).unsqueeze(0)
with torch.no_grad():
  if not FULL_ONLY:
      # (a) reduced ViT: 128-d, 2 heads, 2 blocks
      enc = vit(128, 2, 2)
      feats, meta = enc(img)
      out["vit128_tokens"] = feats[0][0].numpy()                               # [101,128] final-norm tokens
      out["vit128_meta"] = np.array(meta, np.int64)
      # (b) vit_large_reg block shapes (1024-d, 16 heads), 2 blocks
      encL = vit(1024, 16, 2)
      featsL, _ = encL(img)
      out["vit1024_tokens"] = featsL[0][0].numpy().astype(np.float16)
      # (c) reduced decoder on the tokens of (a)
      cfg = NS(model=NS(decode_head=NS(in_channels=[128] * 4, use_cls_token=True, feature_channels=[32, 64, 128, 256],
                                       decoder_channels=[16, 32, 64, 128, 128], up_scale=7,
                                       hidden_channels=[16, 16, 16, 16], n_gru_layers=3, n_downsample=2, iters=3,
                                       slow_fast_gru=True, num_register_tokens=4)),
               data_basic=NS(depth_normalize=(0.1, 200)))
      dec = D.RAFTDepthNormalDPT5(cfg).eval()
      dec.load_state_dict(DW.fill(dec))
      bins = torch.exp(torch.linspace(math.log(0.1), math.log(200), 256)).unsqueeze(0)
      dec.register_buffer("depth_expectation_anchor", bins, persistent=False)
      # intermediates through the module's own sub-blocks (for localising a mismatch)
      B, gh, gw, _, _, nreg = meta
      vf = [[ft[:, 1 + nreg:, :].view(B, gh, gw, 128), ft[:, 0:1 + nreg, :].view(B, 1, 1, 128 * (1 + nreg))] for ft in feats]
      ef = dec.token2feature(vf)
      for i, t in enumerate(ef):
          out[f"dec_encfeat{i}"] = t[0].permute(1, 2, 0).numpy().astype(np.float16)      # NHWC
      ref_feat = dec.decoder_mono(ef)
      out["dec_ref_feat"] = ref_feat[0].permute(1, 2, 0).numpy().astype(np.float16)
      # heads and context encoder on a fresh copy of the features (decoder_mono rectified `ef`
      # in place, exactly what the full forward below hands to the context encoder)
      fmap = ref_feat[:, :-2]
      dpred, _ = dec.regress_depth(fmap)
      npred = dec.pred_normal(fmap, ref_feat[:, -1:])
      depth_init = torch.cat((dpred, ref_feat[:, -2:-1], npred), dim=1)
      out["dec_depth_init"] = depth_init[0].permute(1, 2, 0).numpy()                       # [H4, W4, 6]
      cnet = dec.context_feature_encoder(ef[::-1])
      for i, pair in enumerate(cnet):
          out[f"dec_net{i}"] = torch.tanh(pair[0])[0].permute(1, 2, 0).numpy().astype(np.float16)
          out[f"dec_ctx{i}"] = dec.context_zqr_convs[i](torch.relu(pair[1]))[0].permute(1, 2, 0).numpy().astype(np.float16)
      deltas = []
      hook = dec.update_block.register_forward_hook(
          lambda m, i, o: deltas.append(o[2][0].permute(1, 2, 0).numpy()) if isinstance(o, tuple) and len(o) == 3 else None)
      o = dec([feats, meta])
      hook.remove()
      for i, dl in enumerate(deltas):
          out[f"dec_delta{i}"] = dl
      out["dec_depth"] = o["prediction"][0, 0].numpy()
      out["dec_conf"] = o["confidence"][0, 0].numpy()
      out["dec_normal"] = o["prediction_normal"][0].numpy()
      for k in ("dec_depth", "dec_conf", "dec_normal", "dec_ref_feat", "vit128_tokens", "vit1024_tokens"):
          print(k, out[k].shape, float(np.abs(out[k].astype(np.float32)).mean()), float(np.abs(out[k].astype(np.float32)).max()))
if not FULL_ONLY:
    np.savez_compressed(HERE / "depthnet_golden.npz", **out)
    print("wrote depthnet_golden.npz")


# ---------------------------------------------------------------------------------------------
# Round 3: the two networks the reference actually loads (predictors/metric3d.py:20-25), at their
# FULL depth, width and iteration count -- vit_small_reg (:1211) / vit_large_reg (:1227) with the
# decoder configurations of dino_vit_{small,large}_reg.dpt_raft.py -- on the small 112x168 input
# (101 tokens), which the CPU affords: 12 / 24 residual blocks and 4 / 8 ConvGRU iterations are
# where fp16 drift would accumulate. Written to a second file (depthnet_full_golden.npz).
# ---------------------------------------------------------------------------------------------
FULL = {
    "vits": dict(embed_dim=384, depth=12, heads=6, feature_channels=[96, 192, 384, 768],
                 decoder_channels=[48, 96, 192, 384, 384], hidden=[48, 48, 48, 48], iters=4),
    "vitl": dict(embed_dim=1024, depth=24, heads=16, feature_channels=[256, 512, 1024, 1024],
                 decoder_channels=[128, 256, 512, 1024, 1024], hidden=[128, 128, 128, 128], iters=8),
}
full = {}
with torch.no_grad():
    for name, c in FULL.items():
        enc = vit(c["embed_dim"], c["heads"], c["depth"])
        feats, meta = enc(img)
        full[f"{name}_tokens"] = feats[0][0].numpy().astype(np.float16)
        cfg = NS(model=NS(decode_head=NS(in_channels=[c["embed_dim"]] * 4, use_cls_token=True,
                                         feature_channels=c["feature_channels"], decoder_channels=c["decoder_channels"],
                                         up_scale=7, hidden_channels=c["hidden"], n_gru_layers=3, n_downsample=2,
                                         iters=c["iters"], slow_fast_gru=True, num_register_tokens=4)),
                 data_basic=NS(depth_normalize=(0.1, 200)))
        dec = D.RAFTDepthNormalDPT5(cfg).eval()
        # (round 4: depth channel of the update head rescaled so that the depth map is not clamp-saturated,
        # dn_weights.FULL_DEPTH_GAIN)
        dec.load_state_dict(DW.fill(dec, depth_gain=DW.FULL_DEPTH_GAIN))
        dec.register_buffer("depth_expectation_anchor", bins, persistent=False)
        deltas = []
        hook = dec.update_block.register_forward_hook(
            lambda m, i, o: deltas.append(o[2][0].permute(1, 2, 0).numpy()) if isinstance(o, tuple) and len(o) == 3 else None)
        o = dec([feats, meta])
        hook.remove()
        assert len(deltas) == c["iters"]
        dd = o["prediction"][0, 0]
        assert float(dd.min()) > 0.2 and float(dd.max()) < 199.0, "depth map touches the clamp"
        full[f"{name}_delta_last"] = deltas[-1]
        full[f"{name}_depth"] = o["prediction"][0, 0].numpy()
        full[f"{name}_conf"] = o["confidence"][0, 0].numpy()
        full[f"{name}_normal"] = o["prediction_normal"][0].numpy()
        for k in ("tokens", "depth", "conf", "normal", "delta_last"):
            a = full[f"{name}_{k}"].astype(np.float32)
            print(name, k, a.shape, float(np.abs(a).mean()), float(np.abs(a).max()))
        del enc, dec
    # vit_giant2_reg's block (ViT_DINO_reg.py:1246-1260: 1536-d, 24 heads, ffn_layer='swiglu' -> SwiGLUFFN
    # :300-345, hidden 4096), 4 of its 40 blocks (the full stack is 1.1 G parameters)
    encG = vit(1536, 24, 4, ffn_layer="swiglu")
    assert type(encG.blocks[0][0].mlp).__name__.startswith("SwiGLU"), type(encG.blocks[0][0].mlp)
    featsG, _ = encG(img)
    full["vitg4_tokens"] = featsG[0][0].numpy().astype(np.float16)
    print("vitg4 tokens", full["vitg4_tokens"].shape, float(np.abs(full["vitg4_tokens"].astype(np.float32)).mean()))
np.savez_compressed(HERE / "depthnet_full_golden.npz", **full)
print("wrote depthnet_full_golden.npz")
