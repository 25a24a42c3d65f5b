"""Merging of segmentation regions that hold too few SfM points or whose border carries no depth edge.

Restatement of /root/reference/gs_init_compare/depth_alignment/segmentation/region_merging.py:29-152
(`merge_segmentation_regions`). **Parity unpinned**: the reference builds its region adjacency
graph and its border masks with scikit-image (`ski.graph.rag_boundary`, `ski.morphology.binary_*`,
`relabel_sequential`), which is absent from this image, so the reference function cannot be run
to record fixtures; the scikit-image pieces are restated from their documented behaviour
(8-connected region adjacency; 3x3 cross footprint, erosion with the border counted as set,
dilation with the border counted as clear; sequential relabelling in ascending label order). Like
the reference this is host code: a greedy loop over at most a few dozen regions, once per image."""
from typing import Dict

import numpy as np
import scipy.ndimage as ndi
import torch

from .region_margin import get_actual_margin_size

_CROSS = ndi.generate_binary_structure(2, 1)


def _adjacency(seg: np.ndarray) -> Dict[int, set]:
    """8-connected neighbours of every label."""
    adj: Dict[int, set] = {int(v): set() for v in np.unique(seg)}
    for a, b in ((seg[:, :-1], seg[:, 1:]), (seg[:-1, :], seg[1:, :]), (seg[:-1, :-1], seg[1:, 1:]),
                 (seg[:-1, 1:], seg[1:, :-1])):
        diff = a != b
        for u, v in np.unique(np.stack([a[diff], b[diff]], 1), axis=0):
            adj[int(u)].add(int(v))
            adj[int(v)].add(int(u))
    return adj


def merge_segmentation_regions(pred_depth, sfm_points_camera_coords: torch.Tensor, segmentation, config):
    """segmentation: integer label array [H, W] (numpy or tensor). Returns an int64 tensor on the
    depth map's device with labels 0..K-1."""
    depth = pred_depth.depth
    seg = segmentation.detach().cpu().numpy() if isinstance(segmentation, torch.Tensor) else np.asarray(segmentation)
    seg = seg.astype(np.int64)
    if np.unique(seg).size == 1:                                            # :35-38
        return torch.zeros(seg.shape, dtype=torch.int64, device=depth.device)
    seg = seg + 1                                                           # :43-44
    adj = _adjacency(seg)
    depth_norm = depth / (depth.max() - depth.min() + 1e-8)                 # :46-50
    gy, gx = torch.gradient(depth_norm)
    grad2 = (gy ** 2 + gx ** 2).cpu().numpy()
    pts = sfm_points_camera_coords.cpu().numpy()
    margin = get_actual_margin_size(depth.shape, config.region_margin)
    valid = pred_depth.mask.cpu().numpy()

    def num_pts(rid):                                                       # :57-66
        inner = ndi.binary_erosion(seg == rid, iterations=margin)
        return int((inner & valid)[pts[1], pts[0]].sum())

    def border(rid):                                                        # :68-72
        m = seg == rid
        return ndi.binary_dilation(m, structure=_CROSS) != ndi.binary_erosion(m, structure=_CROSS, border_value=1)

    def border_grad(rid):                                                   # :74-75
        return float(grad2[border(rid)].mean())

    data = {int(i): [num_pts(i), border_grad(i)] for i in np.unique(seg)}   # [num_sfm_pts, mean_border_grad]
    rename: Dict[int, int] = {}
    while len(data) > 1:                                                    # :90-145
        min_grad = min(data, key=lambda i: data[i][1])
        min_pts = min(data, key=lambda i: data[i][0])
        grad_pass = data[min_grad][1] >= config.min_border_grad_threshold
        pts_pass = data[min_pts][0] >= config.min_sfm_pts_in_region
        if grad_pass and pts_pass:
            break
        target = min_grad if not grad_pass else min_pts
        neigh = []
        for n in adj[target]:
            og = n
            while n in rename:
                n = rename[n]
            if og in rename:
                rename[og] = n                                              # path shortcut (:112-114)
            neigh.append(n)
        neigh = np.unique(np.array([n for n in neigh if n != target], dtype=np.int64))
        if neigh.size == 0:                                                 # disconnected (:117-124)
            data[target] = [float("inf"), float("inf")]
            continue
        tb = border(target)
        factors = np.array([float(grad2[tb & border(int(n))].mean()) for n in neigh])
        best = int(neigh[factors.argmin()])
        seg[seg == target] = best
        data[best] = [num_pts(best), border_grad(best)]
        for n in neigh:
            if int(n) != best:
                adj[best].add(int(n))
                adj[int(n)].add(best)
        data.pop(target)
        rename[target] = best
    _, inv = np.unique(seg, return_inverse=True)                            # relabel 0..K-1 (:147-150)
    return torch.from_numpy(inv.reshape(seg.shape).astype(np.int64)).to(depth.device)
