"""Import path of the reference kept; the definition lives in `contracts.py`."""
from .contracts import InputImage  # noqa: F401
