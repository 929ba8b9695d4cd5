"""The stacked-hourglass callers of the epipolar layer: registry names `HG*` / `epipolarHG*` /
`simplemultiviewHG*`, one call of the operator per hourglass stack (two with MERGE both).

Reference surface (modeling/backbones/ProHG.py):
  * factories `hourglass` / `hourglass1` / `hourglass11`                        :326-395  (3 stacks x depth 3, 1 x 3, 1 x 1;
    KEYPOINT.NFEATS features, pre-activation bottleneck modules, no sigmoid, downsample 4)
  * `HourGlassNet.forward(inputs, other_inputs=[other_features, other_KRT, other_heatmaps, KRT, camera, other_camera,
     other_img])` -> `(features, heatmaps, batch_locs, batch_scos, corr_pos, depths, sample_locs, warpedheatmap)`   :193-316
    with `other_features` a LIST (one map per fusion point) and `features` / `heatmaps` lists (one per stack)
  * `getOtherFeat`                                                               :202-237  (FIND_CORR feature | rgb, OTHER_ONLY)
  * attribute `epipolar_sampler`; parameter names of every sub-module as there, so a reference checkpoint loads
    (`conv.{0,1,3,4,6,7}`, `ress.{0,2,3}`, `features.<i>.<j>`, `tmpOuts.<i>`, `trsfeas.<i>`, `trstmps.<i>`, and inside a module
    `conv_A|B|C.{0,2}`, `branch.{0,2}`, inside an hourglass `res.<n>`, `down.<1+n>`, `mid...`, `up.<n>`).

The convolutions are stock PyTorch-ROCm (MIOpen); the fusion is the HIP operator (`epipolar.Epipolar`: with the default 256
features the MFMA tile kernels, eval mode through the one-kernel layer).  Not built: the `meta*` variants (modeling/layers/meta.py is
outside the path) and EPIPOLAR.WARPEDHEATMAP (a visualisation aid, :301-304) -- both raise.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F

from .backbones import BACKBONES, find_peaks
from .config import get_cfg
from .epipolar import Epipolar


def _bn_relu_conv(cin, cout, k):
    """batch norm -> ReLU -> convolution (keys 0 and 2 of the Sequential, as in the reference's modules)"""
    return nn.Sequential(nn.BatchNorm2d(cin), nn.ReLU(inplace=True), nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=True))


class PreActBottleneck(nn.Module):
    """ProHG.py:18-51 `Residual`: 1x1 -> 3x3 -> 1x1 at half width, pre-activation, projected skip when the widths differ."""

    def __init__(self, cin, cout):
        super().__init__()
        mid = cout // 2
        self.conv_A = _bn_relu_conv(cin, mid, 1)
        self.conv_B = _bn_relu_conv(mid, mid, 3)
        self.conv_C = _bn_relu_conv(mid, cout, 1)
        if cin != cout:
            self.branch = _bn_relu_conv(cin, cout, 1)

    def forward(self, x):
        skip = self.branch(x) if hasattr(self, "branch") else x
        return self.conv_C(self.conv_B(self.conv_A(x))) + skip


def _modules(n, feats):
    return [PreActBottleneck(feats, feats) for _ in range(n)]


class HourglassStage(nn.Module):
    """ProHG.py:93-118: skip branch + (pool, modules, recursive middle, modules, bilinear up-sampling to the skip's size)."""

    def __init__(self, depth, n_modules, feats):
        super().__init__()
        self.res = nn.Sequential(*_modules(n_modules, feats))
        self.down = nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=2), *_modules(n_modules, feats))
        self.mid = HourglassStage(depth - 1, n_modules, feats) if depth > 1 else nn.Sequential(*_modules(n_modules, feats))
        self.up = nn.Sequential(*_modules(n_modules, feats))

    def forward(self, x):
        skip = self.res(x)
        y = self.up(self.mid(self.down(skip)))
        return skip + F.interpolate(y, [skip.size(2), skip.size(3)], mode="bilinear", align_corners=True)


class HourGlassPoseNet(nn.Module):
    """ProHG.py:120-316."""

    def __init__(self, cfg, stacks, depth, n_modules=1):
        super().__init__()
        self.cfg = cfg
        body = cfg.BACKBONE.BODY
        if "meta" in body:
            raise NotImplementedError("%s: the Meta fusion layer (modeling/layers/meta.py) is outside this package's path" % body)
        feats, joints = cfg.KEYPOINT.NFEATS, cfg.KEYPOINT.NUM_PTS
        self.nStack, self.sigma, self.downsample = stacks, cfg.KEYPOINT.SIGMA, 4
        stem = []
        for cin, cout, stride in ((3, 32, 2), (32, 32, 1), (32, 64, 1)):
            stem += [nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=True), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
        self.conv = nn.Sequential(*stem)
        self.ress = nn.Sequential(PreActBottleneck(64, 128), nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
                                  PreActBottleneck(128, 128), PreActBottleneck(128, feats))
        self.features = nn.ModuleList(
            nn.Sequential(HourglassStage(depth, n_modules, feats), *_modules(n_modules, feats),
                          nn.Conv2d(feats, feats, kernel_size=1, bias=True), nn.BatchNorm2d(feats), nn.ReLU(inplace=True))
            for _ in range(stacks))
        self.tmpOuts = nn.ModuleList(nn.Conv2d(feats, joints, kernel_size=1, bias=True) for _ in range(stacks))
        self.trsfeas = nn.ModuleList(nn.Conv2d(feats, feats, kernel_size=1, bias=True) for _ in range(stacks - 1))
        self.trstmps = nn.ModuleList(nn.Conv2d(joints, feats, kernel_size=1, bias=True) for _ in range(stacks - 1))
        self.sigmoid = None                                                  # (every factory of the reference: "sigmoid": 0)
        if "epipolarHG" in body:
            self.epipolar_sampler = Epipolar(cfg=cfg)
        self.avgpool = nn.AvgPool2d(kernel_size=4)

    # ------------------------------------------------------------------------------------------------ fusion
    def _fuse(self, i, feat, inputs, other_inputs):
        """getOtherFeat (ProHG.py:202-237) at fusion point i: (fused map, corr_pos, depth, sample_locs)."""
        other_features, other_KRT, _, KRT, camera, other_camera, other_img = other_inputs
        if other_features is None:
            return feat, None, None, None
        e = self.cfg.EPIPOLAR
        body = self.cfg.BACKBONE.BODY
        corr_pos = depth = sample_locs = None
        if "simplemultiviewHG" in body:
            ret = other_features[i]
        elif "epipolarHG" in body:
            if e.FIND_CORR == "rgb":                                          # :218-228: 4 x 4 average-pooled images as the
                assert not e.PRIOR                                            # correspondence maps
                ref1, ref2 = self.avgpool(inputs).detach(), self.avgpool(other_img).detach()
            else:
                ref1, ref2 = feat, other_features[i]
                if not e.OTHER_ONLY and not self.cfg.VIS.EPIPOLAR_LINE:
                    # the layer + `ret + feat` as the fused call the ResNet caller takes (eval mode, 256 features: one data kernel)
                    return self.epipolar_sampler.forward_fused(feat, other_features[i], KRT, other_KRT, camera=camera,
                                                               other_camera=other_camera)
            ret, corr_pos, depth, sample_locs = self.epipolar_sampler(feat, other_features[i], KRT, other_KRT, camera=camera,
                                                                      other_camera=other_camera, ref1=ref1, ref2=ref2)
        else:
            raise NotImplementedError("%s with other_features: no fusion is defined for a plain hourglass (ProHG.py:202-237)" % body)
        if e.OTHER_ONLY:
            return ret, corr_pos, depth, sample_locs
        return ret + feat, corr_pos, depth, sample_locs

    def forward(self, inputs, other_inputs=(None, None, None, None, None, None, None)):
        if inputs.dim() != 4:
            raise ValueError("This model accepts 4 dimension input tensor: %s" % (tuple(inputs.shape),))
        other_features = other_inputs[0]
        merge = self.cfg.EPIPOLAR.MERGE
        finetune = bool(self.cfg.SOLVER.FINETUNE) if "SOLVER" in self.cfg and "FINETUNE" in self.cfg.SOLVER else False
        if self.cfg.EPIPOLAR.WARPEDHEATMAP and other_inputs[2] is not None:
            raise NotImplementedError("EPIPOLAR.WARPEDHEATMAP (ProHG.py:301-304) is a visualisation aid and not built")
        x = self.ress(self.conv(inputs))
        features, heatmaps, corrs, depths = [], [], [], []
        sample_locs = None
        point = 0                                                             # index of the next fusion point (other_features[point])
        for i in range(self.nStack):
            if merge in ("early", "both"):
                fused, corr_pos, depth, sample_locs = self._fuse(point, x, inputs, other_inputs)
                point += 1
                features.append(x)
                corrs.append(corr_pos)
                depths.append(depth)
                feature = self.features[i](fused.detach() if finetune else fused)
                if merge == "both":
                    feature, corr_pos, depth, sample_locs = self._fuse(point, feature, inputs, other_inputs)
                    point += 1
                    features.append(feature)
                    corrs.append(corr_pos)
                    depths.append(depth)
            elif merge == "late":
                feature = self.features[i](x)
                if finetune:
                    feature = feature.detach()
                feature, corr_pos, depth, sample_locs = self._fuse(point, feature, inputs, other_inputs)
                point += 1
                features.append(feature)
                corrs.append(corr_pos)
                depths.append(depth)
            elif merge == "none":
                features.append(x)
                # (ProHG.py:293-294 leaves `feature` unset here and fails at :298 on the first stack; the single-view net the
                #  registry name HG stands for is MERGE late without other_features)
                feature = self.features[i](x)
            else:
                raise NotImplementedError("EPIPOLAR.MERGE %r" % (merge,))
            heat = self.tmpOuts[i](feature)
            heatmaps.append(heat)
            if i < self.nStack - 1:
                x = x + self.trsfeas[i](feature) + self.trstmps[i](heat)
        locs, scos = find_peaks(heatmaps[-1], self.sigma, self.downsample)
        if other_features is None:
            corr_pos, depth = None, None
        else:
            corr_pos, depth = corrs[-1], depths[-1]
        return features, heatmaps, locs, scos, corr_pos, depth, sample_locs, None


def _factory(stacks, depth):
    def build(cfg=None, **kwargs):
        cfg = cfg if cfg is not None else get_cfg()
        if cfg.BACKBONE.PRETRAINED:
            raise NotImplementedError("the reference has no pretrained hourglass either (ProHG.py:348-349)")
        return HourGlassPoseNet(cfg, stacks, depth)

    return build


for _suffix, (_stacks, _depth) in (("", (3, 3)), ("1", (1, 3)), ("11", (1, 1))):          # ProHG.py:319-395
    for _prefix in ("HG", "simplemultiviewHG", "epipolarHG", "metaHG", "metaepipolarHG"):   # (meta*: the constructor raises)
        BACKBONES.register(_prefix + _suffix, _factory(_stacks, _depth))
