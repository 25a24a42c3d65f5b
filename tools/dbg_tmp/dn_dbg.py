import sys, importlib, numpy as np, torch
sys.path.insert(0,'.')
import tests.test_gpu_depthnet as T
N=importlib.import_module('3dgs_monocular_depth_init_amd.depth_prediction.predictors.metric3d_net')
G=T.G
net=N.Metric3DNet(T._state(T.SMALL_CFG), device="cuda", input_size=(112,168), config=T.SMALL_CFG)
tokens=torch.from_numpy(G["vit128_tokens"]).half().cuda()
d,c,n,inter=net.decode(tokens, return_intermediates=True)
di=inter["depth_init"].cpu(); ref=torch.from_numpy(G["dec_depth_init"]).reshape(-1,6)
for ch in range(6):
    print("ch",ch,"nan",int(torch.isnan(di[:,ch]).sum()),"err",float((di[:,ch]-ref[:,ch]).abs().nan_to_num(0).max()),"refmax",float(ref[:,ch].abs().max()))
for i in range(3):
    m=inter["nets"][i]; r=torch.from_numpy(G[f"dec_net{i}"].astype(np.float32)).reshape(-1,m.C)
    print("net",i,float((m.t[:,:m.C].float().cpu()-r).abs().max()), float(r.abs().max()))
    m=inter["ctxs"][i]; r=torch.from_numpy(G[f"dec_ctx{i}"].astype(np.float32)).reshape(-1,m.C)
    print("ctx",i,float((m.t[:,:m.C].float().cpu()-r).abs().max()), float(r.abs().max()))
for i,dl in enumerate(inter["deltas"]):
    r=torch.from_numpy(G[f"dec_delta{i}"]).reshape(-1,6)
    print("delta",i,[float((dl.cpu()[:,ch]-r[:,ch]).abs().nan_to_num(0).max()) for ch in range(6)], float(r.abs().max()))
