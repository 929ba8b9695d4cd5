"""Read-only import harness for the upstream reference tree (TEST INFRASTRUCTURE ONLY).

This file lets the *real* reference implementation (`/root/reference`, pure
Python/PyTorch) be imported and executed on CPU inside the build container so
that (a) the restated oracle in `oracle/` can be pinned against it and (b) the
golden fixtures under `tests/golden/` can be generated from it
(`tests/golden/make_golden.py`).

Nothing in the product package, `bench.py` timed region or the `-m gpu` tests
may import this module: `/root/reference` does not exist on the GPU box.

What is shimmed:
  * `yacs.config.CfgNode`  -- absent from the image; a minimal attribute-dict
    stand-in with `merge_from_file` / `merge_from_list` / `freeze` / `clone`.
  * empty stub modules for `cv2`, `torchvision`, `IPython`, `tensorboardX`
    which the reference imports at module scope but never uses on the path.
  * `PIL.PILLOW_VERSION` alias (reference data/transforms/image.py:6).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("EPIPOLAR_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modeling", "layers"))


def _cfgnode_class():
    # the yacs stand-in lives in the product package (it is also the product's
    # own config class); test infrastructure may import the product, never the
    # other way round
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from epipolar_transformers_amd.config import CfgNode

    return CfgNode


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


_INSTALLED = False


def install():
    """Install shims and put the reference tree on sys.path (idempotent)."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    if "yacs" not in sys.modules:
        try:
            import yacs.config  # noqa: F401
        except ImportError:
            yacs = _stub("yacs")
            yacs.config = _stub("yacs.config", CfgNode=_cfgnode_class())
    for name, attrs in [
        ("cv2", dict(IMREAD_COLOR=1, IMREAD_IGNORE_ORIENTATION=128, INTER_LINEAR=1)),
        ("torchvision", {}),
        ("torchvision.transforms", {}),
        ("torchvision.transforms.functional", {}),
        ("torchvision.datasets", {}),
        ("torchvision.datasets.folder", dict(default_loader=lambda path: None)),
        ("IPython", dict(embed=lambda *a, **k: None)),
        ("tensorboardX", dict(SummaryWriter=object)),
    ]:
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                _stub(name, **attrs)
    try:
        import PIL

        if not hasattr(PIL, "PILLOW_VERSION"):
            PIL.PILLOW_VERSION = PIL.__version__
    except ImportError:
        pass
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _INSTALLED = True


def load_cfg(yaml_relpath=None, overrides=()):
    """Return the reference's global `cfg` singleton, reset to defaults and
    merged with `yaml_relpath` (relative to the reference root) + overrides."""
    install()
    import importlib

    import core.config as rc  # reference core/config.py

    global _DEFAULTS
    try:
        _DEFAULTS
    except NameError:
        _DEFAULTS = rc._C.clone()
    cfg = rc._C
    cfg.defrost()
    fresh = _DEFAULTS.clone()
    for k in list(cfg.keys()):
        dict.__delitem__(cfg, k)
    for k, v in fresh.items():
        dict.__setitem__(cfg, k, v)
    if yaml_relpath:
        cfg.merge_from_file(os.path.join(REFERENCE_ROOT, yaml_relpath))
    if overrides:
        cfg.merge_from_list(list(overrides))
    return cfg


def reference_epipolar(yaml_relpath="configs/epipolar/keypoint_h36m_zresidual_fixed.yaml",
                       overrides=(), debug=False):
    """Build the reference `Epipolar` module (modeling/layers/epipolar.py:11)."""
    cfg = load_cfg(yaml_relpath, overrides)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from modeling.layers.epipolar import Epipolar  # reference

        mod = Epipolar(debug=debug)
    return mod, cfg
