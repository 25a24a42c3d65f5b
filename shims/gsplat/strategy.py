"""`from gsplat.strategy import DefaultStrategy, MCMCStrategy` (reference: config.py:5,
trainer.py:11, runner.py:21) resolves to this build's strategies."""
import importlib

_s = importlib.import_module("3dgs_monocular_depth_init_amd.strategy")
DefaultStrategy = _s.DefaultStrategy
MCMCStrategy = _s.MCMCStrategy
