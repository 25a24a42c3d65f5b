import sys, importlib, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0,'.')
import tests.test_gpu_depthnet as T
N=importlib.import_module('3dgs_monocular_depth_init_amd.depth_prediction.predictors.metric3d_net')
G=T.G
sd=T._state(T.SMALL_CFG)
net=N.Metric3DNet(sd, device="cuda", input_size=(112,168), config=T.SMALL_CFG)
def nchw(m): return m.t[:, :m.C].float().reshape(m.H,m.W,m.C).permute(2,0,1)[None]
def cmp(name, m, ref):
    got=nchw(m); print(name, "err", float((got-ref).abs().max()), "refmax", float(ref.abs().max()), "nan", int(torch.isnan(got).sum()))
tokens=torch.from_numpy(G["vit128_tokens"]).half().cuda()
d,c,n,inter=net.decode(tokens, return_intermediates=True)
x0=inter["encfeat"][3]
x=nchw(x0)
p="decoder.context_feature_encoder.outputs04.0."
w=lambda k: sd[p+k].cuda().float()
rb=net.ctx["04"][0]
y=net.conv(x0, rb["c1"], 3, N.Map(x0.H,x0.W,16,"cuda"))
ry=F.conv2d(x, w("0.conv1.weight").half().float(), w("0.conv1.bias"), padding=1)
cmp("conv1", y, ry)
net.layernorm2d(y, rb["norm1"], relu=True)
ry=F.relu(F.layer_norm(ry.permute(0,2,3,1),(16,),w("0.norm1.weight"),w("0.norm1.bias"),1e-5).permute(0,3,1,2))
cmp("ln1", y, ry)
y2=net.conv(y, rb["c2"], 3, N.Map(x0.H,x0.W,16,"cuda"))
ry2=F.conv2d(ry, w("0.conv2.weight").half().float(), w("0.conv2.bias"), padding=1)
cmp("conv2", y2, ry2)
xs=net.conv(x0, rb["ds"], 1, N.Map(x0.H,x0.W,16,"cuda"))
rxs=F.conv2d(x, w("0.downsample.0.weight").half().float(), w("0.downsample.0.bias"))
cmp("ds", xs, rxs)
# normal predictor
rf=inter["ref_feat"]; feat=rf.chan(0,32); xf=nchw(feat)
q="decoder.normal_predictor."
a=net.conv(feat, net.npred[0], 3, N.Map(rf.H,rf.W,128,"cuda"), act=N.ACT_RELU)
ra=F.relu(F.conv2d(xf, sd[q+"0.weight"].cuda().half().float(), sd[q+"0.bias"].cuda(), padding=1)); cmp("np0", a, ra)
b=net.conv(a, net.npred[1], 1, N.Map(rf.H,rf.W,128,"cuda"), act=N.ACT_RELU)
rb_=F.relu(F.conv2d(ra, sd[q+"2.weight"].cuda().half().float(), sd[q+"2.bias"].cuda())); cmp("np1", b, rb_)
c2=net.conv(b, net.npred[2], 1, N.Map(rf.H,rf.W,128,"cuda"), act=N.ACT_RELU)
rc=F.relu(F.conv2d(rb_, sd[q+"4.weight"].cuda().half().float(), sd[q+"4.bias"].cuda())); cmp("np2", c2, rc)
e=net.conv(c2, net.npred[3], 1, N.Map(rf.H,rf.W,3,"cuda"))
re=F.conv2d(rc, sd[q+"6.weight"].cuda().half().float(), sd[q+"6.bias"].cuda()); cmp("np3", e, re)
print(e.t[:3])
