"""Drop-in boundary above the operator: the pose backbone that owns and calls
the epipolar layer, and the backbone registry it is looked up in.

Reference surface:
  * `registry.BACKBONES[cfg.BACKBONE.BODY](cfg)`         modeling/registry.py:5, utils/registry.py:6-39
  * names `poseR-{18,34,50,101,152}`, `epipolarposeR-*`   modeling/backbones/resnet.py:495-504
  * `forward(x, other_inputs=[other_features, other_KRT, other_heatmaps, KRT, camera,
     other_camera, other_img])` -> 8-tuple                modeling/backbones/resnet.py:364-437
  * attribute `epipolar_sampler` (`epipolar_sampler1` for MERGE both) so checkpoint keys match.

The convolutional trunk is the standard ResNet + three stride-2 deconvolutions +
1x1 head; it stays stock PyTorch-ROCm (MIOpen / hipBLASLt run it on the MFMA
units) and is kept in channels_last memory so the deconv head hands the fused
kernel the NHWC layout it wants without a transpose.  Parameter names follow the
reference/torchvision convention so `load_state_dict` of a released checkpoint
works unchanged.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F

from .config import amd_knob, get_cfg
from .epipolar import Epipolar


class Registry(dict):
    """name -> factory(cfg) mapping with a decorator (utils/registry.py:6-39)."""

    def register(self, name, factory=None):
        def _put(fn):
            if name in self:
                raise KeyError("backbone %r registered twice" % name)
            self[name] = fn
            return fn

        return _put(factory) if factory is not None else _put


BACKBONES = Registry()


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, momentum=0.1):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes, momentum=momentum)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes, momentum=momentum)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + skip)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, momentum=0.1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=momentum)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=momentum)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, momentum=momentum)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + skip)


RESNET_SPEC = {"18": (BasicBlock, (2, 2, 2, 2)), "34": (BasicBlock, (3, 4, 6, 3)), "50": (Bottleneck, (3, 4, 6, 3)),
               "101": (Bottleneck, (3, 4, 23, 3)), "152": (Bottleneck, (3, 8, 36, 3))}


def soft_argmax_peaks(heatmaps: torch.Tensor, radius: float, downsample: int, threshold: float = 1e-6,
                      legacy_floor_division: bool = False):
    """Batched restatement of find_tensor_peak_batch (modeling/backbones/basic_batch.py:17-63), which the
    reference calls once per sample in a Python loop (resnet.py:424-430).
    heatmaps: (N, J, H, W) -> locations (N, J, 2) in image coordinates, scores (N, J).

    `index / W` is true division under the torch in this image (it was integer division before torch 1.5,
    SURVEY.md section 8c); legacy_floor_division=True restores the behaviour the authors trained with."""
    n, j, h, w = heatmaps.shape
    flat = heatmaps.reshape(n * j, h * w)
    score, index = flat.max(1)
    index_w = (index % w).float()
    index_h = torch.div(index, w, rounding_mode="floor").float() if legacy_floor_division else (index / w).float()

    def norm(x, length):
        return -1.0 + 2.0 * x / (length - 1)

    x0, y0 = norm(index_w - radius, w), norm(index_h - radius, h)
    x1, y1 = norm(index_w + radius, w), norm(index_h + radius, h)
    iradius = int(radius + 0.5)
    theta = torch.zeros((n * j, 2, 3), dtype=heatmaps.dtype, device=heatmaps.device)
    theta[:, 0, 0] = (x1 - x0) / 2
    theta[:, 0, 2] = (x1 + x0) / 2
    theta[:, 1, 1] = (y1 - y0) / 2
    theta[:, 1, 2] = (y1 + y0) / 2
    side = iradius * 2 + 1
    grid = F.affine_grid(theta, torch.Size([n * j, 1, side, side]), align_corners=False)
    sub = F.grid_sample(heatmaps.reshape(n * j, 1, h, w), grid, mode="bilinear", padding_mode="zeros",
                        align_corners=False).squeeze(1)
    sub = F.threshold(sub, threshold, 0)
    ramp = torch.arange(-radius, radius + 0.0001, radius * 1.0 / iradius, dtype=heatmaps.dtype, device=heatmaps.device)
    total = sub.reshape(n * j, -1).sum(1) + 2.220446049250313e-16           # np.finfo(float).eps
    x = (sub * ramp.view(1, 1, side)).reshape(n * j, -1).sum(1) / total + index_w
    y = (sub * ramp.view(1, side, 1)).reshape(n * j, -1).sum(1) / total + index_h
    x = x * downsample + downsample / 2.0 - 0.5                             # pix2coord, multiview.py:154-157
    y = y * downsample + downsample / 2.0 - 0.5
    return torch.stack([x, y], 1).view(n, j, 2), score.view(n, j)


def find_peaks(heatmaps: torch.Tensor, radius: float, downsample: float, legacy_floor_division: bool = False):
    """Peak finder of the head: the fused HIP kernel (ops.heatmap_peaks, one launch) for float32 heat maps on the GPU,
    the batched torch restatement (soft_argmax_peaks) otherwise (CPU plumbing runs of the single-view net)."""
    if heatmaps.is_cuda and heatmaps.dtype == torch.float32:
        from . import ops

        return ops.heatmap_peaks(heatmaps, radius, downsample, legacy_floor_division=legacy_floor_division)
    return soft_argmax_peaks(heatmaps, radius, downsample, legacy_floor_division=legacy_floor_division)


class PoseResNet(nn.Module):
    def __init__(self, block, layers, cfg, **kwargs):
        super().__init__()
        self.cfg = cfg
        momentum = None if cfg.BACKBONE.BN_MOMENTUM < 0 else cfg.BACKBONE.BN_MOMENTUM
        self._momentum = momentum
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=momentum)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(block, 64, layers[0])
        self.layer2 = self._stage(block, 128, layers[1], stride=2)
        self.layer3 = self._stage(block, 256, layers[2], stride=2)
        self.layer4 = self._stage(block, 512, layers[3], stride=2)
        head = []
        for planes in (256, 256, 256):                                      # resnet.py:266-268
            head += [nn.ConvTranspose2d(self.inplanes, planes, kernel_size=4, stride=2, padding=1, output_padding=0,
                                        bias=False),
                     nn.BatchNorm2d(planes, momentum=momentum), nn.ReLU(inplace=True)]
            self.inplanes = planes
        self.deconv_layers = nn.Sequential(*head)
        self.final_layer = nn.Conv2d(256, cfg.KEYPOINT.NUM_PTS, kernel_size=1, stride=1, padding=0)
        if "epipolarpose" in cfg.BACKBONE.BODY:                             # resnet.py:299-305
            if cfg.EPIPOLAR.MERGE == "both":
                self.epipolar_sampler1 = Epipolar(cfg=cfg)
            self.epipolar_sampler = Epipolar(cfg=cfg)
        else:
            self.epipolar_sampler = None
            self.epipolar_sampler1 = None

    def _stage(self, block, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion, momentum=self._momentum))
        mods = [block(self.inplanes, planes, stride, down, self._momentum)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes, momentum=self._momentum) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def _fuse(self, feat, sampler, other_features, KRT, other_KRT, camera, other_camera):
        """getOtherFeat (resnet.py:377-388): epipolar layer + `ret + feat`, with the adds fused into the
        epilogue kernel when the layer runs in eval mode."""
        if other_features is None:
            return feat, None, None, None
        if self.cfg.VIS.EPIPOLAR_LINE:
            ret, corr_pos, depth, sample_locs = sampler(feat, other_features, KRT, other_KRT,
                                                        camera=camera, other_camera=other_camera)
            return ret + feat, corr_pos, depth, sample_locs
        return sampler.forward_fused(feat, other_features, KRT, other_KRT, camera=camera, other_camera=other_camera)

    def trunk(self, x):
        """Image -> the pre-fusion feature map of resnet.py:406 (ResNet stages + the three deconvolutions), without
        any fusion: what `forward(x, other_inputs=None)` returns as element 0 (model.py:244 calls it for that).
        (Round 4 tried slicing large eval batches -- 128 images of 384 x 384 are 1.2 GB of activations per layer -- on the
        suspicion that MIOpen leaves its tuned fp32 solvers there: it does not, one pass is 4 % faster, scripts/e2e_shapes.py.)"""
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)             # NHWC all the way to the fused kernel
        dt = str(amd_knob(self.cfg, "TRUNK_DTYPE", "fp32"))
        if dt != "fp32":
            # the stock convolutions in reduced precision (an option: NOT the reference's arithmetic; bench.py reports what it does
            # to the detections); the fused layer takes fp32 maps, channels_last as they come
            if dt not in ("bf16", "fp16"):
                raise ValueError("EPIPOLAR_AMD.TRUNK_DTYPE must be fp32, bf16 or fp16, not %r" % (dt,))
            with torch.autocast(x.device.type, dtype=torch.bfloat16 if dt == "bf16" else torch.float16):
                x = self.layer1(self.maxpool(self.relu(self.bn1(self.conv1(x)))))
                x = self.deconv_layers(self.layer4(self.layer3(self.layer2(x))))
            return x.float()
        x = self.layer1(self.maxpool(self.relu(self.bn1(self.conv1(x)))))
        return self.deconv_layers(self.layer4(self.layer3(self.layer2(x))))

    def forward(self, x, other_inputs=(None, None, None, None, None, None, None)):
        other_features, other_KRT, other_heatmaps, KRT, camera, other_camera, other_img = other_inputs
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)             # NHWC all the way to the fused kernel
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer1(x)
        merge = self.cfg.EPIPOLAR.MERGE
        corr_pos = depth = sample_locs = None
        if merge == "early":
            x, corr_pos, depth, sample_locs = self._fuse(x, self.epipolar_sampler, other_features, KRT, other_KRT,
                                                         camera, other_camera)
        elif merge == "both":
            x, _, _, _ = self._fuse(x, self.epipolar_sampler, other_features, KRT, other_KRT, camera, other_camera)
        x = self.layer4(self.layer3(self.layer2(x)))
        feature = self.deconv_layers(x)
        if merge == "late":
            x, corr_pos, depth, sample_locs = self._fuse(feature, self.epipolar_sampler, other_features, KRT,
                                                         other_KRT, camera, other_camera)
        elif merge == "both":
            x, corr_pos, depth, sample_locs = self._fuse(feature, self.epipolar_sampler1, other_features, KRT,
                                                         other_KRT, camera, other_camera)
        else:
            x = feature
        heatmap = self.final_layer(x)
        locs, scos = find_peaks(heatmap, self.cfg.KEYPOINT.SIGMA, self.cfg.BACKBONE.DOWNSAMPLE)
        if other_features is None:
            corr_pos, depth = None, None
        return feature, [heatmap], locs, scos, corr_pos, depth, sample_locs, None

    def init_weights(self, pretrained=None):
        """Load a trunk checkpoint (state_dict or path) the way the reference does (resnet.py:439-471 +
        utils/model_serialization.py:10-120): the deconvolution head and the final 1x1 convolution are first
        re-initialised (normal(0, 0.001) weights, BN 1/0), `cfg.WEIGHTS_PREFIX` ('module.' of a DataParallel
        checkpoint) is stripped / replaced, every parameter of this model takes the loaded tensor whose key is the
        LONGEST SUFFIX of its own name, and `final_layer.*` is never loaded.  Returns the list of parameters that were
        matched; raises if nothing matched at all (a silently random-initialised trunk is never what was meant)."""
        if pretrained is None:
            return []
        import logging
        import warnings

        log = logging.getLogger(__name__)
        state = torch.load(pretrained, map_location="cpu") if isinstance(pretrained, str) else pretrained
        if isinstance(state, dict) and "model" in state:
            state = state["model"]
        for m in self.deconv_layers.modules():                              # resnet.py:448-459
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.final_layer.weight, std=0.001)                 # resnet.py:461-467
        nn.init.constant_(self.final_layer.bias, 0)
        # strip_prefix_if_present (model_serialization.py:62-80)
        prefix = getattr(self.cfg, "WEIGHTS_PREFIX", "module.")
        replace = getattr(self.cfg, "WEIGHTS_PREFIX_REPLACE", "")
        allow = getattr(self.cfg, "WEIGHTS_ALLOW_DIFF_PREFIX", False)
        keys = sorted(state.keys())
        if all(k.startswith(prefix) for k in keys) or allow:
            if not all(k.startswith(prefix) for k in keys):
                warnings.warn("[Warning] Not all keys contain the prefix " + prefix)
            stripped = {}
            for k, v in state.items():
                if k != "" and not k.startswith(prefix):
                    continue
                stripped[(replace + k) if (prefix == "" and replace != "") else k.replace(prefix, replace)] = v
            state = stripped
        elif prefix:
            warnings.warn("[Warning] Not all keys contain the prefix " + prefix)
        # align_and_update_state_dicts (model_serialization.py:10-60): longest loaded key that is a suffix
        own = self.state_dict()
        ignored = ("final_layer.bias", "final_layer.weight")
        loaded_keys = sorted(state.keys())
        matched, update = [], {}
        for k in sorted(own.keys()):
            if k in ignored:
                continue
            best = max((j for j in loaded_keys if k.endswith(j)), key=len, default=None)
            if best is None:
                continue
            if tuple(state[best].shape) != tuple(own[k].shape):
                raise RuntimeError("checkpoint tensor %s %s does not fit parameter %s %s" %
                                   (best, tuple(state[best].shape), k, tuple(own[k].shape)))
            update[k] = state[best]
            matched.append(k)
        trunk = [k for k in own if not k.startswith(("final_layer.", "epipolar_sampler")) and "num_batches_tracked" not in k]
        if not matched:
            raise RuntimeError("checkpoint has no key matching this model (WEIGHTS_PREFIX=%r): nothing was loaded" % prefix)
        missing = [k for k in trunk if k not in update]
        if missing:
            log.warning("init_weights: %d of %d trunk tensors not found in the checkpoint (first: %s)",
                        len(missing), len(trunk), missing[0])
        self.load_state_dict(update, strict=False)
        return matched


def get_pose_net(cfg=None, **kwargs):
    cfg = cfg if cfg is not None else get_cfg()
    depth = cfg.BACKBONE.BODY.split("-")[-1]
    block, layers = RESNET_SPEC[depth]
    model = PoseResNet(block, layers, cfg, **kwargs)
    if cfg.BACKBONE.PRETRAINED and cfg.BACKBONE.PRETRAINED_WEIGHTS:
        model.init_weights(cfg.BACKBONE.PRETRAINED_WEIGHTS)                 # no network: model-zoo URLs are not fetched
    return model


for _d in RESNET_SPEC:
    BACKBONES.register("poseR-" + _d, get_pose_net)
    BACKBONES.register("epipolarposeR-" + _d, get_pose_net)


def build_backbone(cfg=None):
    """modeling/backbones/backbone.py:9-13."""
    cfg = cfg if cfg is not None else get_cfg()
    return BACKBONES[cfg.BACKBONE.BODY](cfg)


from . import hourglass  # noqa: E402,F401  (registers HG* / epipolarHG* / simplemultiviewHG*: modeling/backbones/ProHG.py:319-395)
