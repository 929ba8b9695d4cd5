"""Config surface of the hot path.

The reference drives everything from one global yacs `cfg`
(core/config.py:5-292) that `Epipolar` reads at construction and at call time.
yacs is not installed offline, so `CfgNode` below is a small stand-in with the
same surface the reference uses (attribute access, merge_from_file /
merge_from_list, freeze/defrost, clone).  `default_cfg()` carries the reference's
keys *for this path only*, verbatim, plus one new sub-node `EPIPOLAR_AMD` for
kernel-only knobs so every reference YAML parses unchanged (unknown keys from
other subsystems are kept, not rejected).

`get_cfg()` returns the active global config: this package's own, or -- after
`use_cfg(reference_cfg)` -- the reference's `core.cfg` singleton, which is how
the layer drops into the reference's `main.py --cfg` flow.
"""
from __future__ import annotations

import ast
import copy


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        if init:
            for k, v in init.items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError("cfg is frozen: cannot set %s" % name)
        self[name] = value

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = type(self)()
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

    def _merge_dict(self, other, path="", strict=True):
        for k, v in other.items():
            if k not in self:
                if strict:
                    raise KeyError("non-existent config key: %s%s" % (path, k))
                dict.__setitem__(self, k, type(self)(v) if isinstance(v, dict) else self._coerce(v))
                continue
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise TypeError("expected a mapping for %s%s" % (path, k))
                self[k]._merge_dict(v, path + k + ".", strict)
            else:
                dict.__setitem__(self, k, self._coerce(v, self[k]))

    def merge_from_file(self, filename, strict=None):
        import yaml

        with open(filename, "r") as fh:
            loaded = yaml.safe_load(fh) or {}
        self._merge_dict(loaded, strict=self._strict() if strict is None else strict)

    def merge_from_other_cfg(self, other):
        self._merge_dict(other, strict=self._strict())

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError("override list must be KEY VALUE pairs")
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for part in parts[:-1]:
                node = node[part]
            if parts[-1] not in node:
                if self._strict():
                    raise KeyError("non-existent config key: %s" % key)
                dict.__setitem__(node, parts[-1], self._coerce(value))
            else:
                dict.__setitem__(node, parts[-1], self._coerce(value, node[parts[-1]]))

    def _strict(self):
        return True


class LenientCfgNode(CfgNode):
    """Accepts keys it does not know (other subsystems of the reference)."""

    def _strict(self):
        return False


def default_cfg() -> CfgNode:
    """Defaults copied key-for-key from the reference (core/config.py:16-25,
    48-55,69-118,128-140,277-279) for the keys this path reads."""
    C = LenientCfgNode()
    C.BACKBONE = LenientCfgNode()
    C.BACKBONE.ENABLED = False
    C.BACKBONE.BODY = "R-50"
    C.BACKBONE.PRETRAINED = True
    C.BACKBONE.PRETRAINED_WEIGHTS = ""
    C.BACKBONE.DOWNSAMPLE = 4
    C.BACKBONE.BN_MOMENTUM = 0.1
    C.BACKBONE.SYNC_BN = False
    C.KEYPOINT = LenientCfgNode()
    C.KEYPOINT.ENABLED = False
    C.KEYPOINT.SIGMA = 25.0
    C.KEYPOINT.NUM_PTS = 21
    C.KEYPOINT.HEATMAP_SIZE = (224, 224)
    C.KEYPOINT.NFEATS = 256
    C.EPIPOLAR = LenientCfgNode()
    C.EPIPOLAR.VIS = False
    C.EPIPOLAR.TOPK = 1
    C.EPIPOLAR.TOPK_RANGE = (1, 2)
    C.EPIPOLAR.ATTENTION = "max"
    C.EPIPOLAR.SIMILARITY = "dot"
    C.EPIPOLAR.SAMPLESIZE = 64
    C.EPIPOLAR.SOFTMAX_ENABLED = True
    C.EPIPOLAR.SOFTMAXSCALE = 1 / C.EPIPOLAR.SAMPLESIZE ** 0.5   # frozen at 1/8 (config.py:86)
    C.EPIPOLAR.SOFTMAXBETA = True
    C.EPIPOLAR.MERGE = "early"
    C.EPIPOLAR.OTHER_ONLY = False
    C.EPIPOLAR.OTHER_GRAD = ("other1", "other2")
    C.EPIPOLAR.SHARE_WEIGHTS = False
    C.EPIPOLAR.PARAMETERIZED = ()
    C.EPIPOLAR.ZRESIDUAL = False
    C.EPIPOLAR.MULTITEST = False
    C.EPIPOLAR.WARPEDHEATMAP = False
    C.EPIPOLAR.PRIOR = False
    C.EPIPOLAR.PRIORMUL = False
    C.EPIPOLAR.REPROJECT_LOSS_WEIGHT = 0.0
    C.EPIPOLAR.SIM_LOSS_WEIGHT = 0.0
    C.EPIPOLAR.PRETRAINED = True
    C.EPIPOLAR.FIND_CORR = "feature"
    C.EPIPOLAR.BOTTLENECK = 1
    C.EPIPOLAR.POOLING = False
    C.EPIPOLAR.USE_CORRECT_NORMALIZE = False
    C.DATASETS = LenientCfgNode()
    C.DATASETS.IMAGE_SIZE = (256, 256)
    C.DATASETS.IMAGE_RESIZE = 1.0
    C.DATASETS.PREDICT_RESIZE = 1.0
    C.DATASETS.CAMERAS = ()
    C.VIS = LenientCfgNode()
    C.VIS.EPIPOLAR_LINE = False
    C.DEVICE = "cuda"
    # ---- new, kernel-only knobs (not in the reference) -------------------------
    C.EPIPOLAR_AMD = LenientCfgNode()
    C.EPIPOLAR_AMD.ALIGN_CORNERS = False      # F.grid_sample semantics to reproduce
    C.EPIPOLAR_AMD.VARIANT = 0                # EtLayerDesc.variant bits
    C.EPIPOLAR_AMD.FUSED_EPILOGUE = True      # eval: fold BN and fuse the residual adds in one kernel
    C.EPIPOLAR_AMD.SHARD_P2P = False          # view-sharded partition: source maps by one all-to-all (each block to the rank that
                                              # samples it) instead of the all-gather BASELINE.json's north star names (G x the bytes)
    C.EPIPOLAR_AMD.TRUNK_DTYPE = "fp32"       # arithmetic of the stock trunk (convolutions of `PoseResNet.trunk`): fp32 (the reference's)
                                              # | bf16 | fp16 = torch.autocast around it; the epipolar layer stays fp32 either way
    return C


_ACTIVE = default_cfg()


def get_cfg():
    return _ACTIVE


def use_cfg(cfg):
    """Make `cfg` (e.g. the reference's `core.cfg`) the config the layer reads."""
    global _ACTIVE
    _ACTIVE = cfg
    return cfg


def amd_knob(cfg, name, default):
    node = cfg.get("EPIPOLAR_AMD") if hasattr(cfg, "get") else None
    if node is None or name not in node:
        return default
    return node[name]
