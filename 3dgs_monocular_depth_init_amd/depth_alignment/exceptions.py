class LowDepthAlignmentConfidenceError(Exception):
    """depth_alignment/exceptions.py of the reference."""
