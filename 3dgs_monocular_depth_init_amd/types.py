"""gs_init_compare/types.py:6-10 of the reference."""
from typing import NamedTuple

import torch


class InputImage(NamedTuple):
    data: torch.Tensor        # [H,W,3] float in [0,1]
    name: str
    cam2world: torch.Tensor   # [4,4]
    K: torch.Tensor           # [3,3]
