import importlib

rasterization = importlib.import_module("3dgs_monocular_depth_init_amd.rendering").rasterization
