"""`gsplat` name shim: put `<repo>/shims` (and `<repo>`) on PYTHONPATH and the
reference's `from gsplat.rendering import rasterization` (runner.py:19) resolves
to the MI355X build. Provided: the hot-path operator (`gsplat.rendering`), the two
densification strategies (`gsplat.strategy`) and the process launcher
(`gsplat.distributed.cli`, trainer.py:58); the rest of gsplat (compression, exporter, the
Gaussian-sharded all-to-all helpers) is out of this tier's scope."""
