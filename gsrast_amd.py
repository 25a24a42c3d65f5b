"""Importable alias of the package directory `3dgs_monocular_depth_init_amd/`
(a Python identifier cannot start with a digit): `import gsrast_amd` gives the
same module object as importlib.import_module("3dgs_monocular_depth_init_amd")."""
import importlib
import sys

_pkg = importlib.import_module("3dgs_monocular_depth_init_amd")
sys.modules[__name__] = _pkg
