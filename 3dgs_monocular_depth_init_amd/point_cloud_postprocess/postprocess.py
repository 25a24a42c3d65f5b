"""Point-cloud post-processing: the reference's native subsampler on the GPU.

Mirror of /root/reference/gs_init_compare/point_cloud_postprocess/postprocess.py:25-77
(`postprocess_point_cloud`) and of the pybind entry it calls,
`pointcloud_subsampling.subsample_pointcloud(points, rgbs, intrinsic_matrices,
camera_2_world_matrices, image_sizes, params)`
(native_modules/subsampling/src/pointcloud_subsampling.cpp:22-67, C++/Eigen there;
csrc/pointcloud.hip here). Both switches are off by default in the reference
(point_cloud_postprocess/config.py:15-17). LOF outlier removal (postprocess.py:16-22: scikit-learn's
LocalOutlierFactor on the CPU, all cores) is `knn.local_outlier_factor` on the GPU: exact 40 nearest
neighbours by a cell-grid ring search, then the two LOF passes; pinned by the output of
scikit-learn itself (tests/golden/make_lof_golden.py).
"""
from __future__ import annotations

import numpy as np
import torch

from .._lib import call, load, ptr
from .config import OutlierRemovalMethod, PointCloudPostprocessConfig


def _st():
    return torch.cuda.current_stream().cuda_stream


@torch.no_grad()
def subsample_pointcloud_device(points: torch.Tensor, rgbs: torch.Tensor, Ks: torch.Tensor, Ps: torch.Tensor,
                                image_sizes: torch.Tensor, params):
    """Device tensors in, device tensors out: (points' [n,3], rgbs' [n,3], extents [N]).
    Ks [C,3,3], Ps [C,3,4] (the projection matrices K R [I|-C] the reference passes under the
    name camera_2_world_matrices), image_sizes [C,2] int32 (width, height)."""
    lib = load()
    dev = points.device
    points = points.float().contiguous()
    rgbs = rgbs.float().contiguous()
    N, C = points.shape[0], Ks.shape[0]
    if rgbs.shape != (N, 3) or points.shape != (N, 3):
        raise ValueError("Input points / rgbs arrays must have shape (N, 3) with equal N")
    if Ps.shape != (C, 3, 4) or image_sizes.shape != (C, 2):
        raise ValueError("Number of intrinsic_matrices must match number of camera_2_world_matrices and image_sizes.")
    Ks = Ks.to(dev).float().contiguous()
    Ps = Ps.to(dev).float().contiguous()
    sizes = image_sizes.to(dev).to(torch.int32).contiguous()
    extents = torch.empty(N, dtype=torch.float32, device=dev)
    call("gsr_pc_min_extents", N, C, ptr(points), ptr(Ks), ptr(Ps), ptr(sizes), ptr(extents), _st())
    nbytes = int(lib.gsr_pc_subsample_workspace_bytes(N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out_p = torch.empty(max(N, 1), 3, dtype=torch.float32, device=dev)
    out_c = torch.empty(max(N, 1), 3, dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    call("gsr_pc_subsample", N, ptr(points), ptr(rgbs), ptr(extents), float(params.max_bbox_aspect_ratio),
         float(params.min_extent_multiplier), ptr(ws), nbytes, ptr(out_p), ptr(out_c), ptr(count), _st())
    n = int(count.item())
    return out_p[:n], out_c[:n], extents


def subsample_pointcloud(points, rgbs, intrinsic_matrices, camera_2_world_matrices, image_sizes, params,
                         device="cuda"):
    """The pybind signature (numpy arrays / lists of matrices in, numpy out). Returns the
    reference's 5-tuple (points, rgbs, min_gaussian_extents, debug_points, debug_rgbs); the two
    debug arrays (a colour-coded copy of the merged groups, used only for a PLY export) are empty."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
    p, c, e = subsample_pointcloud_device(
        t(points), t(rgbs), t(np.stack([np.asarray(k) for k in intrinsic_matrices])),
        t(np.stack([np.asarray(m) for m in camera_2_world_matrices])),
        torch.as_tensor(np.asarray(image_sizes), dtype=torch.int32, device=device), params)
    empty = np.zeros((0, 3), np.float32)
    return p.cpu().numpy(), c.cpu().numpy(), e.cpu().numpy(), empty, empty.copy()


def lof_outlier_removal(pts: torch.Tensor, config: PointCloudPostprocessConfig) -> torch.Tensor:
    """postprocess.py:16-22: True where LocalOutlierFactor(n_neighbors=config.lof_num_neighbors) predicts -1."""
    from ..knn import local_outlier_factor
    return local_outlier_factor(pts, config.lof_num_neighbors)[0]


def get_outlier_removal_func(method: OutlierRemovalMethod):          # postprocess.py:25-29
    if method == OutlierRemovalMethod.lof:
        return lof_outlier_removal
    raise ValueError(f"Unknown outlier removal method: {method}")


def postprocess_point_cloud(pts: torch.Tensor, rgbs: torch.Tensor, intrinsic_matrices, proj_matrices, image_sizes,
                            config: PointCloudPostprocessConfig, device):
    """postprocess.py:25-77 (PLY debug exports not produced)."""
    if config.outlier_removal != OutlierRemovalMethod.off:
        outliers = get_outlier_removal_func(config.outlier_removal)(pts.to(device), config)
        pts, rgbs = pts.to(device)[~outliers], rgbs.to(device)[~outliers]     # (postprocess.py:41-58, no PLY exports)
    if config.subsample:
        Ks = torch.as_tensor(np.stack([np.asarray(k) for k in intrinsic_matrices]), dtype=torch.float32)
        Ps = torch.as_tensor(np.stack([np.asarray(m) for m in proj_matrices]), dtype=torch.float32)
        sizes = torch.as_tensor(np.asarray(image_sizes), dtype=torch.int32)
        pts, rgbs, _ = subsample_pointcloud_device(pts.to(device), rgbs.to(device), Ks, Ps, sizes,
                                                   config.subsample_params)
    return pts, rgbs
