"""DepthAlignmentPipeline, no-segmentation branch
(/root/reference/gs_init_compare/depth_alignment/pipeline.py:170-293; the
segmentation branch 201-247 is out of scope: SLIC/SAM/region merging)."""
from dataclasses import dataclass

import torch

from .interface import DepthAlignmentResult, DepthAlignmentStrategy

INVALID_DEPTH_VAL = -42.0            # pipeline.py:253


@dataclass
class DepthAlignmentPipeline:
    config: object
    alignment: DepthAlignmentStrategy

    @staticmethod
    def from_config(config):
        if config.mdi.alignment.segmenter is not None:
            raise NotImplementedError("segmentation-based alignment is out of scope of this build")
        return DepthAlignmentPipeline(config, config.mdi.alignment.aligner.get_implementation())

    def align(self, image, predicted_depth, sfm_points_camera_coords, sfm_points_depth, config,
              debug_export_dir=None) -> DepthAlignmentResult:
        # one region (id 0) holding every SfM point and every pixel (pipeline.py:248-251, 257-283)
        if sfm_points_depth.shape[0] == 0:
            out_depth = torch.full_like(predicted_depth.depth, INVALID_DEPTH_VAL)   # region dropped
        else:
            res = self.alignment.align(predicted_depth, sfm_points_camera_coords, sfm_points_depth,
                                       config, debug_export_dir)
            out_depth = res.aligned_depth
        return DepthAlignmentResult(
            aligned_depth=out_depth,
            mask=(out_depth != INVALID_DEPTH_VAL) & predicted_depth.mask)     # pipeline.py:285-288
