"""`gsplat` name shim: put `<repo>/shims` (and `<repo>`) on PYTHONPATH and the
reference's `from gsplat.rendering import rasterization` (runner.py:19) resolves
to the MI355X build. Only the hot-path operator is provided; the rest of gsplat
(strategy, compression, exporter, distributed.cli) is out of this tier's scope."""
