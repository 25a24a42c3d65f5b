"""DepthAlignmentPipeline (/root/reference/gs_init_compare/depth_alignment/pipeline.py:170-293).

Both branches of `align`: one region holding every pixel and SfM point (:248-251), or the regions
of a segmentation (:193-247) -- label map from a segmenter callable, region merging, the margin
mask around region boundaries (`gsr_region_margin_mask`), then one alignment per region on the SfM
points inside it, written through the region's pixels. The reference's segmenters (SLIC, SAM) are
third-party and not built: `segmentation` is any callable with their signature
(interface.py:44-46). Debug exports are not produced."""
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Optional

import torch

from .interface import DepthAlignmentResult, DepthAlignmentStrategy
from .segmentation.region_margin import calculate_region_margin_mask
from .segmentation.region_merging import merge_segmentation_regions

INVALID_DEPTH_VAL = -42.0            # pipeline.py:253


@dataclass
class DepthAlignmentPipeline:
    config: object
    segmentation: Optional[Callable]          # DepthSegmentationFn or None
    alignment: DepthAlignmentStrategy
    merge: Callable = merge_segmentation_regions

    @staticmethod
    def from_config(config):
        segmentation = None
        if config.mdi.alignment.segmenter is not None:
            segmentation = config.mdi.alignment.segmenter.get_implementation()
        return DepthAlignmentPipeline(config, segmentation, config.mdi.alignment.aligner.get_implementation())

    def align(self, image, predicted_depth, sfm_points_camera_coords, sfm_points_depth, config,
              debug_export_dir=None) -> DepthAlignmentResult:
        depth = predicted_depth.depth
        dev = depth.device
        xs, ys = sfm_points_camera_coords[0], sfm_points_camera_coords[1]
        if self.segmentation:
            seg_cfg = config.mdi.alignment.segmentation
            seg = self.segmentation(predicted_depth, Path(config.mdi.cache_dir) / "checkpoints", seg_cfg)
            seg = self.merge(predicted_depth, sfm_points_camera_coords, seg, seg_cfg)
            seg = torch.as_tensor(seg).to(dev)
            deadzone = calculate_region_margin_mask(seg, seg_cfg.region_margin)
            region_ids = torch.unique(seg[predicted_depth.mask]).tolist()                    # :230
            if seg_cfg.propagate_mask:
                predicted_depth.mask = predicted_depth.mask & deadzone                       # :233-234
            # SfM points per region, inside the margin mask (:236-245): ONE stable sort by region id
            # instead of one boolean pass and one host round trip per region -- within a region
            # the points keep their order, as torch.where gives them
            pts_region = seg[ys, xs]
            pts_ok = deadzone[ys, xs]
            big = int(seg.max().item()) + 1 if seg.numel() else 1
            key = torch.where(pts_ok, pts_region, torch.full_like(pts_region, big))
            order = torch.argsort(key, stable=True)
            ids_t = torch.tensor(region_ids, dtype=key.dtype, device=dev)
            lo = torch.searchsorted(key[order], ids_t, right=False).tolist()
            hi = torch.searchsorted(key[order], ids_t, right=True).tolist()
            region_points = [order[a:b] for a, b in zip(lo, hi)]
        else:
            seg = None
            region_ids = [0]
            region_points = [torch.arange(sfm_points_depth.shape[0], device=dev)]           # :248-251

        out_depth = torch.full_like(depth, INVALID_DEPTH_VAL)
        for region in region_ids:
            # (the reference indexes its per-region list with the region ID, :258-261 -- the ids of a
            # merged segmentation are 0..K-1, and a region without a valid pixel shifts the list
            # under the ids after it, up to an IndexError; reproduced as is)
            idx = region_points[int(region)]
            if idx.numel() == 0:
                continue                                  # region dropped: stays invalid (:264-270)
            res = self.alignment.align(predicted_depth, sfm_points_camera_coords[:, idx], sfm_points_depth[idx],
                                       config, debug_export_dir)
            if seg is None:
                out_depth = res.aligned_depth
            else:
                out_depth = torch.where(seg == region, res.aligned_depth, out_depth)        # :272-283
        return DepthAlignmentResult(
            aligned_depth=out_depth,
            mask=(out_depth != INVALID_DEPTH_VAL) & predicted_depth.mask)                    # :285-288
