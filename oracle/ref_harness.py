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

import ast
import copy
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("EPIPOLAR_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modeling", "layers"))


class _CfgNode(dict):
    """Minimal yacs.config.CfgNode stand-in (attribute access + merge)."""

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        if init:
            for k, v in init.items():
                self[k] = _CfgNode(v) if isinstance(v, dict) and not isinstance(v, _CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError("cfg is frozen: cannot set %s" % name)
        self[name] = value

    # -- yacs surface used by the reference --------------------------------
    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, _CfgNode):
                v._set_frozen(flag)

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = _CfgNode()
        for k, v in self.items():
            dict.__setitem__(new, k, copy.deepcopy(v, memo))
        return new

    @staticmethod
    def _coerce(value, like=None):
        if isinstance(value, str):
            try:
                value = ast.literal_eval(value)
            except (ValueError, SyntaxError):
                pass
        if isinstance(like, tuple) and isinstance(value, list):
            value = tuple(value)
        if isinstance(like, float) and isinstance(value, int) and not isinstance(value, bool):
            value = float(value)
        return value

    def _merge_dict(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError("non-existent config key: %s%s" % (path, k))
            if isinstance(self[k], _CfgNode):
                if not isinstance(v, dict):
                    raise TypeError("expected mapping for %s%s" % (path, k))
                self[k]._merge_dict(v, path + k + ".")
            else:
                dict.__setitem__(self, k, self._coerce(v, self[k]))

    def merge_from_file(self, filename):
        import yaml

        with open(filename, "r") as fh:
            loaded = yaml.safe_load(fh) or {}
        self._merge_dict(loaded)

    def merge_from_other_cfg(self, other):
        self._merge_dict(other)

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "override list must be KEY VALUE pairs"
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("non-existent config key: %s" % key)
            dict.__setitem__(node, parts[-1], self._coerce(value, node[parts[-1]]))


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
            yacs.config = _stub("yacs.config", CfgNode=_CfgNode)
    for name, attrs in [
        ("cv2", dict(IMREAD_COLOR=1, IMREAD_IGNORE_ORIENTATION=128, INTER_LINEAR=1)),
        ("torchvision", {}),
        ("torchvision.transforms", {}),
        ("torchvision.transforms.functional", {}),
        ("torchvision.datasets", {}),
        ("torchvision.datasets.folder", {}),
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
