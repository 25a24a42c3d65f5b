import importlib, sys, torch
sys.path.insert(0, '/root/repo')
lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
x = torch.zeros(8, 64)
for k in range(8):
    x[k] = (k + 1)
x[0] += torch.arange(64) * 0.001
xin = x.cuda().contiguous()
out = torch.empty(128, device="cuda"); idx = torch.empty(64, dtype=torch.int32, device="cuda")
lib.call("gsr_debug_tree_reduce8", xin.data_ptr(), out.data_ptr(), idx.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("tree", out[:64:4].cpu().tolist())
print("wsum", out[64::8].cpu().tolist())
print("idx ", idx[::4].cpu().tolist())
print("tot ", x.sum(1).tolist())
