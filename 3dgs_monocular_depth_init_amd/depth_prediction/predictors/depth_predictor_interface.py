"""Import path of the reference kept; the definitions live in `contracts.py`."""
from ...contracts import CameraIntrinsics, DepthPredictor, PredictedDepth  # noqa: F401
