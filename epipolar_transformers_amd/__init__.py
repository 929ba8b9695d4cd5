"""MI355X-native Epipolar Transformer hot path.

Importing the package does not load the HIP library; the first operator call
does, and raises if it was not built -- there is no CPU fallback.
"""
from .config import CfgNode, default_cfg, get_cfg, use_cfg  # noqa: F401

__all__ = ["CfgNode", "default_cfg", "get_cfg", "use_cfg", "Epipolar"]


def __getattr__(name):
    if name == "Epipolar":
        from .epipolar import Epipolar

        return Epipolar
    raise AttributeError(name)
