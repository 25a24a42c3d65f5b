"""Import path of the reference kept; the definitions live in `contracts.py`."""
from ..contracts import AdaptiveSubsamplingConfig, NumSfMPointsMaskConfig  # noqa: F401
