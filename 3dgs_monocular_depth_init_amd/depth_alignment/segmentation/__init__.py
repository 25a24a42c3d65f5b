"""Segmentation-based alignment (SURVEY.md row F4 tail): the region margin mask on the GPU and
region merging on the host; the segmenters themselves (SLIC: scikit-image, SAM: a ViT-H
checkpoint) are third-party and not part of this build -- any callable with the reference's
`DepthSegmentationFn` signature (depth_alignment/interface.py:44-46) is accepted instead."""
from .region_margin import calculate_region_margin_mask, get_actual_margin_size  # noqa: F401
from .region_merging import merge_segmentation_regions  # noqa: F401
