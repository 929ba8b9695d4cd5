"""`Epipolar` -- drop-in for the reference operator of the same name
(modeling/layers/epipolar.py:11-269): same constructor, same forward signature,
same 4-tuple, same parameter names (`z.weight`, `z.bias`, `bn.*`) so released
checkpoints load.  The Python per-sample loop, the two grid_sample calls and
the K x C x H x W intermediates are replaced by one fused HIP kernel
(csrc/epipolar_kernels.hip) behind the C ABI of include/epipolar_amd.h.

Supported here is the mode every shipped epipolar config runs
(SURVEY.md section 0): ATTENTION avg, SIMILARITY dot, soft-max on or off,
optional 'z' (+BN, +ZRESIDUAL), either normalize convention.  The ablation
branches (theta/phi/g bottleneck, POOLING, PRIOR, ATTENTION max, cosine,
FIND_CORR rgb, reprojection loss, given depth) raise NotImplementedError --
they are rows "next N4" of SURVEY.md section 8f, not silently approximated.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .camera import PairAlgebraCache
from .config import amd_knob, get_cfg


class zeroinitBN(nn.BatchNorm2d):
    """BatchNorm2d whose affine parameters start at zero so the 'z' branch starts
    as the identity residual (modeling/layers/BN.py:12-52).  Same state_dict keys."""

    def reset_parameters(self):
        self.reset_running_stats()
        if self.affine:
            nn.init.zeros_(self.weight)
            nn.init.zeros_(self.bias)


class Epipolar(nn.Module):
    def __init__(self, debug=False, cfg=None):
        super().__init__()
        self.cfg = cfg if cfg is not None else get_cfg()
        cfg = self.cfg
        self.debug = debug
        self.downsample = cfg.BACKBONE.DOWNSAMPLE
        self.feat_h, self.feat_w = cfg.KEYPOINT.HEATMAP_SIZE
        self.sample_size = cfg.EPIPOLAR.SAMPLESIZE
        self.epsilon = 0.001
        nfeats = cfg.KEYPOINT.NFEATS
        if cfg.EPIPOLAR.BOTTLENECK != 1:
            raise NotImplementedError("EPIPOLAR.BOTTLENECK != 1 (theta/phi/g branch) is not on the fused path yet")
        for name in ("theta", "phi", "g"):
            if name in cfg.EPIPOLAR.PARAMETERIZED:
                raise NotImplementedError("EPIPOLAR.PARAMETERIZED %r is not on the fused path yet" % name)
        if cfg.EPIPOLAR.PRIOR:
            raise NotImplementedError("EPIPOLAR.PRIOR is not on the fused path yet")
        if "z" in cfg.EPIPOLAR.PARAMETERIZED:
            self.z = nn.Conv2d(nfeats, nfeats, kernel_size=1, stride=1, padding=0, bias=True)   # epipolar.py:64
            self.bn = zeroinitBN(nfeats)                                                        # epipolar.py:65
        self._spec = None
        self._cams = PairAlgebraCache()

    # ------------------------------------------------------------------ spec
    def layer_spec(self) -> ops.LayerSpec:
        cfg = self.cfg
        mask = (1 if "other1" in cfg.EPIPOLAR.OTHER_GRAD else 0) | (2 if "other2" in cfg.EPIPOLAR.OTHER_GRAD else 0)
        key = (self.feat_h, self.feat_w, self.sample_size, float(cfg.BACKBONE.DOWNSAMPLE),
               float(cfg.DATASETS.IMAGE_RESIZE), float(cfg.DATASETS.PREDICT_RESIZE),
               bool(cfg.EPIPOLAR.USE_CORRECT_NORMALIZE), bool(amd_knob(cfg, "ALIGN_CORNERS", False)),
               float(cfg.EPIPOLAR.SOFTMAXSCALE), bool(cfg.EPIPOLAR.SOFTMAX_ENABLED), mask,
               int(amd_knob(cfg, "VARIANT", 0)))
        if self._spec is None or self._spec[0] != key:
            spec = ops.LayerSpec(H=key[0], W=key[1], K=key[2], downsample=key[3], image_resize=key[4],
                                 predict_resize=key[5], correct_normalize=key[6], align_corners=key[7],
                                 softmax_scale=key[8], softmax_enabled=key[9], eps=self.epsilon,
                                 src_grad_mask=key[10], variant=key[11])
            self._spec = (key, spec)
        return self._spec[1]

    def _check_mode(self, depth, ref1, ref2):
        e = self.cfg.EPIPOLAR
        assert e.ATTENTION in {"avg", "max"}                 # epipolar.py:107
        assert e.SIMILARITY in {"cos", "dot", "prior"}        # epipolar.py:108
        unsupported = []
        if e.ATTENTION != "avg":
            unsupported.append("ATTENTION=%s" % e.ATTENTION)
        if e.SIMILARITY != "dot":
            unsupported.append("SIMILARITY=%s" % e.SIMILARITY)
        if e.FIND_CORR != "feature" or ref1 is not None or ref2 is not None:
            unsupported.append("FIND_CORR=rgb")
        if e.POOLING:
            unsupported.append("POOLING")
        if e.REPROJECT_LOSS_WEIGHT != 0:
            unsupported.append("REPROJECT_LOSS_WEIGHT")
        if depth is not None:
            unsupported.append("externally supplied depth")
        if unsupported:
            raise NotImplementedError("not on the fused MI355X path yet: " + ", ".join(unsupported))

    # --------------------------------------------------------------- forward
    def attend(self, feat1, feat2, P1, P2):
        """The fused kernel only: (out, attn, corr_pos), `out` before the z branch."""
        cam = self._cams.get(P1, P2, feat1.device)
        return ops.EpipolarAttend.apply(feat1, feat2, cam, self.layer_spec())

    def _finalize(self, out, feat1=None):
        """z / bn / ZRESIDUAL (epipolar.py:249-255) and, when feat1 is given, the
        residual fusion of resnet.py:388 in the same pass.  Returns (finalout, x|None)."""
        cfg = self.cfg
        has_z = "z" in cfg.EPIPOLAR.PARAMETERIZED
        grad_mode = torch.is_grad_enabled() and (out.requires_grad or (has_z and self.z.weight.requires_grad))
        fused_ok = bool(amd_knob(cfg, "FUSED_EPILOGUE", True)) and not grad_mode and \
            not (has_z and self.bn.training) and (not has_z or cfg.EPIPOLAR.ZRESIDUAL)
        if not fused_ok:
            # training (batch statistics / autograd): plain torch ops on the kernel's output
            finalout = out
            if has_z:
                finalout = self.bn(self.z(out))
                if cfg.EPIPOLAR.ZRESIDUAL:
                    finalout = finalout + out
            return finalout, (finalout + feat1 if feat1 is not None else None)
        out_l = ops.to_nhwc(out)
        feat_l = ops.to_nhwc(feat1) if feat1 is not None else None
        if has_z:
            y = ops.to_nhwc(self.z(out))                     # 1x1 conv = GEMM (MIOpen / hipBLASLt, MFMA)
            inv = torch.rsqrt(self.bn.running_var + self.bn.eps)
            scale = (self.bn.weight * inv).contiguous()
            shift = (self.bn.bias - self.bn.running_mean * scale).contiguous()
            fin, x = ops.residual_epilogue(feat_l, out_l, y, scale, shift, True, feat_l is not None)
        else:
            fin, x = ops.residual_epilogue(feat_l, out_l, None, None, None, feat_l is None, feat_l is not None)
            fin = fin if fin is not None else out_l
        to_logical = lambda t: t.permute(0, 3, 1, 2) if t is not None else None
        return to_logical(fin), to_logical(x)

    def forward(self, feat1, feat2, P1, P2, depth=None, camera=None, other_camera=None, ref1=None, ref2=None):
        """Same contract as the reference (epipolar.py:82-269):
        feat1, feat2: N x C x H x W ; P1, P2: N x 3 x 4
        returns (finalout, corr_pos[N,H,W,2], depth[N,K,H,W], sample_locs | None)."""
        self._check_mode(depth, ref1, ref2)
        out, attn, corr_pos = self.attend(feat1, feat2, P1, P2)
        finalout, _ = self._finalize(out)
        sample_locs = None
        if self.debug or self.cfg.VIS.EPIPOLAR_LINE:
            cam = self._cams.get(P1, P2, feat1.device)
            sample_locs = ops.sample_locs(self.layer_spec(), cam)            # (K,N,H,W,2)
            if not self.debug:
                sample_locs = sample_locs.transpose(0, 1)                    # epipolar.py:267
        return finalout, corr_pos, attn, sample_locs

    def forward_fused(self, feat1, feat2, P1, P2):
        """forward + `ret + feat` (resnet.py:388) with the adds fused into the
        epilogue kernel.  Returns (x, corr_pos, depth, None)."""
        self._check_mode(None, None, None)
        out, attn, corr_pos = self.attend(feat1, feat2, P1, P2)
        finalout, x = self._finalize(out, feat1)
        return x, corr_pos, attn, None
