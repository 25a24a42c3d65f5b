"""MI355X-native hot path of deivse/3dgs_monocular_depth_init.

The directory name starts with a digit (it mirrors the reference's repo name),
so import it with `importlib.import_module("3dgs_monocular_depth_init_amd")`
or through the alias module `gsrast_amd` at the repo root.

Public surface:
  rendering.rasterization(...)      gsplat.rendering.rasterization drop-in
  build.build()                     compile libgsrast.so for gfx950
"""
from . import _lib  # noqa: F401
from .build import build  # noqa: F401

__all__ = ["build", "rasterization"]


def __getattr__(name):
    if name == "rasterization":
        from .rendering import rasterization
        return rasterization
    raise AttributeError(name)
