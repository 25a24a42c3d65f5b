"""Region margin mask: the pixels farther than the margin from every region boundary.

Mirror of /root/reference/gs_init_compare/depth_alignment/segmentation/region_margin.py:6-35
(`snap_to_int_if_close`, `get_actual_margin_size`, `calculate_region_margin_mask`): the
reference box-blurs the label map in fp32 (kornia-style filter2d, replicate padding) and keeps
the pixels whose blurred label equals their own; `gsr_region_margin_mask` takes the window sums
in exact integer arithmetic (csrc/init_depth.hip). Pinned by tests/golden/segalign_golden.npz."""
import torch

from ..._lib import call, ptr

KERNEL_REFERENCE_IMSIZE = 1297          # region_margin.py:17


def get_actual_margin_size(image_shape, region_margin) -> int:
    return int(region_margin * max(image_shape) / KERNEL_REFERENCE_IMSIZE)


@torch.no_grad()
def calculate_region_margin_mask(region_map: torch.Tensor, region_margin: int) -> torch.Tensor:
    if region_margin == 0:
        return torch.ones_like(region_map, dtype=torch.bool)
    if not region_map.is_cuda:
        raise RuntimeError("calculate_region_margin_mask: the label map must live on the GPU (no CPU fallback)")
    H, W = region_map.shape
    half = get_actual_margin_size(region_map.shape, region_margin)
    labels = region_map.to(torch.int32).contiguous()
    rows = torch.empty(H * W, dtype=torch.int64, device=labels.device)
    mask = torch.empty(H, W, dtype=torch.uint8, device=labels.device)
    call("gsr_region_margin_mask", H, W, half, ptr(labels), ptr(rows), ptr(mask),
         torch.cuda.current_stream().cuda_stream)
    return mask.bool()
