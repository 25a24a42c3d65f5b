import sys, importlib, torch
sys.path.insert(0, '.')
from tests import scenes
R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
W,H=1920,1080
for N in (1_000_000, 2_000_000):
    sc = scenes.make_scene(N, 0); p={k:v.cuda() for k,v in sc.items()}
    vm,K = scenes.cameras([31]); vm,K=vm.cuda(),K.cuda()
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    w = torch.stack([torch.sin(xx / 97.0 + yy / 131.0 + c) for c in range(3)], -1)[None].cuda() / (H * W)
    def render(pp):
        return R.rasterization(pp["means"],pp["quats"],pp["scales"],pp["opacities"],(pp["sh0"],pp["shN"]),vm,K,W,H,sh_degree=3,packed=False)
    leaves={k:v.clone().requires_grad_(True) for k,v in p.items()}
    rc,_,_=render(leaves); (rc*w).sum().backward()
    g=torch.Generator().manual_seed(4)
    d=torch.randn(p["means"].shape,generator=g).cuda(); d[:,2]=0; d=d@vm[0,:3,:3]
    an=float((leaves["means"].grad.double()*d.double()).sum())
    def loss(pp):
        with torch.no_grad(): return float((render(pp)[0].double()*w.double()).sum())
    for eps in (4e-4,2e-4,1e-4,5e-5,2.5e-5,1e-5):
        plus=dict(p); plus["means"]=p["means"]+eps*d; minus=dict(p); minus["means"]=p["means"]-eps*d
        fd=(loss(plus)-loss(minus))/(2*eps)
        print(N, "eps",eps,"fd",fd,"analytic",an, flush=True)
