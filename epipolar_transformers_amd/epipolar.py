"""`Epipolar` -- drop-in for the reference operator of the same name
(modeling/layers/epipolar.py:11-269): same constructor, same forward signature,
same 4-tuple, same parameter names (`z.weight`, `z.bias`, `bn.*`) so released
checkpoints load.  The Python per-sample loop, the two grid_sample calls and
the K x C x H x W intermediates are replaced by one fused HIP kernel
(csrc/et_forward*.hip) behind the C ABI of include/epipolar_amd.h.

The mode every headline config runs -- ATTENTION avg, SIMILARITY dot, soft-max on or off,
optional 'z' (+BN, +ZRESIDUAL), either normalize convention -- takes the fused HIP kernels.  The operator's
other branches (theta/phi/g bottleneck, POOLING, ATTENTION max, cosine similarity,
PRIOR / PRIORMUL, SIMILARITY prior, FIND_CORR rgb -- e.g. configs/epipolar/keypoint_h36m_param.yaml) run through ONE
general HIP kernel over three tensors (`_attend_general_hip` -> et_epipolar_forward_general) with a HIP backward for every
one of them since ABI 12 (et_epipolar_backward_general: cosine, ATTENTION max, both priors incl. the prior tables' own
gradient).  The torch restatement of the reference's op sequence (`_attend_general_chunk`, chunked over pairs) is what the
tests compare the kernels with; the module only takes it (with an `EpipolarSlowPathWarning`) for shapes outside the
kernel's limits (`_general_kernel_applies`) -- no shipped YAML reaches it.
An externally supplied `depth` (epipolar.py:101-104, 217-218) takes the general kernel too (`_attend_with_depth`).  Only the
reprojection loss raises NotImplementedError (dead for PoseResNet, SURVEY.md a12: its caller unpacks four values,
resnet.py:385-387).
"""
from __future__ import annotations

import contextlib
import threading

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .camera import PairAlgebraCache
from .config import amd_knob, get_cfg


_ANY = object()     # `_general_kernel_applies`: "camera ids not part of the question" (routing queries of the tests)


class EpipolarSlowPathWarning(UserWarning):
    """A call took the chunked torch restatement instead of a HIP kernel (see the module docstring)."""


_warned_slow = False


def _warn_slow_path(cfg):
    global _warned_slow
    if _warned_slow:
        return
    _warned_slow = True
    import warnings

    e = cfg.EPIPOLAR
    warnings.warn("Epipolar: this configuration (ATTENTION %s, SIMILARITY %s, PRIOR %s, POOLING %s, gradients %s) runs the "
                  "chunked torch restatement of the reference's op sequence, not a HIP kernel -- about 5x slower than the "
                  "general kernel (shape outside its limits, or EPIPOLAR_AMD.GENERAL_KERNEL False)" %
                  (e.ATTENTION, e.SIMILARITY, e.PRIOR, e.POOLING, torch.is_grad_enabled()), EpipolarSlowPathWarning,
                  stacklevel=3)


class zeroinitBN(nn.BatchNorm2d):
    """BatchNorm2d whose affine parameters start at zero so the 'z' branch starts
    as the identity residual (modeling/layers/BN.py:12-52).  Same state_dict keys."""

    def reset_parameters(self):
        self.reset_running_stats()
        if self.affine:
            nn.init.zeros_(self.weight)
            nn.init.zeros_(self.bias)


class _TrainEpilogue(torch.autograd.Function):
    """bn(z(out)) [+ out] [+ feat] with BATCH statistics (epipolar.py:250-253, BN.py:59-82 with training = True, resnet.py:388)
    for the 256-channel head as two passes of the hand-written GEMM kernel instead of five stock ops:
      1. `ops.z_batch_stats`: y = z(out) (kept: the batch norm's input, as autograd keeps it in the reference) and its per-channel
         batch mean / variance (per-block centred sums merged pairwise in double: no cancellation, no atomics);
      2. `ops.residual_gemm` with the statistics folded into the weight: Wf = diag(gamma / sigma) Wz [+ I],
         bias = (bz - mean) gamma / sigma + beta -- the same kernel the eval path runs.
    Backward (`ops.z_backward`): one pass over g and the saved y for the batch norm's two sums, then the same GEMM kernel
    once more -- it forms the batch norm's input gradient dy on the fly, writes it and returns d out = dy . Wz + g; only the
    weight gradient dy^T out goes to the library (the convolution's wgrad).  Measured at Config 2 (profiles/r05_train_epilogue.txt)
    against the alternatives kept behind two development switches: aten's own batch-norm / convolution backward on the 4-D
    tensors (slower than autograd's stock path on channels-last memory) and plain 2-D torch ops (`_backward_rows`: the tall
    dy^T out GEMM takes 18 ms in rocBLAS)."""

    ATEN_BACKWARD = False     # development switches (scripts/train_epilogue_time.py): aten's 4-D backward ops ...
    ROWS_BACKWARD = False     # ... or plain 2-D torch ops, instead of the et_z_backward kernels
    LIBRARY_WGRAD = False     # ... the library's 1x1 convolution wgrad instead of et_z_wgrad

    @staticmethod
    def _backward_rows(g4, o4, y4, zw, gamma, mean, invstd, zresidual, need_out):
        """d out, d Wz, d bz, d gamma, d beta of x = gamma (y - mean) invstd + beta [+ out], y = out Wz^T + bz, over rows."""
        c = o4.shape[-1]
        g, o, y = g4.reshape(-1, c), o4.reshape(-1, c), y4.reshape(-1, c)
        m = g.shape[0]
        dbeta = g.sum(0)
        gy = torch.einsum("mc,mc->c", g, y)
        dgamma = (gy - mean * dbeta) * invstd                    # sum g yhat,  yhat = (y - mean) invstd
        k = gamma * invstd
        # dy = k (g - dbeta / m - yhat dgamma / m)  =  a g + b y + c0   per channel
        b = -k * dgamma * invstd / m
        c0 = k * (mean * invstd * dgamma - dbeta) / m
        dy = torch.addcmul(c0, y, b).addcmul_(g, k)
        w2 = zw.reshape(c, c)
        dzw = (dy.t() @ o).reshape(zw.shape)
        dzb = dy.sum(0)
        dout = None
        if need_out:
            dout = (torch.addmm(g, dy, w2) if zresidual else dy @ w2).view(o4.shape)
        return dout, dzw, dzb, dgamma, dbeta


    @staticmethod
    def forward(ctx, out, feat, zw, zb, gamma, beta, eps, zresidual):
        o = ops.to_nhwc(out)
        c = o.shape[-1]
        w2 = zw.detach().reshape(c, c)
        y, mean, var = ops.z_batch_stats(o, ops.residual_gemm_pack(w2), zb.detach().contiguous())
        invstd = torch.rsqrt(var + eps)
        s = gamma.detach() * invstd
        wf = w2 * s[:, None]
        if zresidual:
            wf = wf + torch.eye(c, dtype=wf.dtype, device=wf.device)
        bf = ((zb.detach() - mean) * s + beta.detach()).contiguous()
        x = ops.residual_gemm(o, ops.residual_gemm_pack(wf.contiguous()), bf, None if feat is None else ops.to_nhwc(feat))
        ctx.save_for_backward(o, y, zw, gamma, mean, invstd)
        ctx.eps, ctx.zresidual, ctx.has_feat = eps, zresidual, feat is not None
        ctx.mark_non_differentiable(mean, var)
        return x.permute(0, 3, 1, 2), mean, var

    @staticmethod
    def backward(ctx, gx, _gm, _gv):
        o, y, zw, gamma, mean, invstd = ctx.saved_tensors
        g = gx.contiguous(memory_format=torch.channels_last)
        if not (_TrainEpilogue.ATEN_BACKWARD or _TrainEpilogue.ROWS_BACKWARD):
            # et_z_backward: the batch norm's sums (one pass over g and y), then ONE GEMM kernel that forms dy on the fly, writes
            # it, and returns d out = dy . Wz + g; the weight gradient dy^T out is the library's convolution wgrad
            c = o.shape[-1]
            dout, dy, dgamma, dbeta = ops.z_backward(g.permute(0, 2, 3, 1), y, mean, invstd, gamma.detach().contiguous(),
                                                     ops.residual_gemm_pack(zw.detach().reshape(c, c).t().contiguous()), ctx.zresidual)
            if _TrainEpilogue.LIBRARY_WGRAD:
                _, dzw, dzb = torch.ops.aten.convolution_backward(dy.permute(0, 3, 1, 2), o.permute(0, 3, 1, 2), zw, [c], [1, 1], [0, 0],
                                                                 [1, 1], False, [0, 0], 1, [False, True, True])
            else:
                dzw, dzb = ops.z_wgrad(dy, o)
                dzw = dzw.view(zw.shape)
            return (dout.permute(0, 3, 1, 2) if ctx.needs_input_grad[0] else None, g if ctx.has_feat and ctx.needs_input_grad[1] else None,
                    dzw, dzb, dgamma, dbeta, None, None)
        if not _TrainEpilogue.ATEN_BACKWARD:
            dout, dzw, dzb, dgamma, dbeta = _TrainEpilogue._backward_rows(g.permute(0, 2, 3, 1), o, y, zw, gamma, mean, invstd,
                                                                          ctx.zresidual, bool(ctx.needs_input_grad[0]))
            return (dout.permute(0, 3, 1, 2) if dout is not None else None, g if ctx.has_feat and ctx.needs_input_grad[1] else None,
                    dzw, dzb, dgamma, dbeta, None, None)
        dy, dgamma, dbeta = torch.ops.aten.native_batch_norm_backward(
            g, y.permute(0, 3, 1, 2), gamma, None, None, mean, invstd, True, ctx.eps, [True, True, True])
        dout, dzw, dzb = torch.ops.aten.convolution_backward(
            dy, o.permute(0, 3, 1, 2), zw, [zw.shape[0]], [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
            [bool(ctx.needs_input_grad[0]), True, True])
        if ctx.needs_input_grad[0] and ctx.zresidual:
            dout = dout + g
        return (dout if ctx.needs_input_grad[0] else None, g if ctx.has_feat and ctx.needs_input_grad[1] else None,
                dzw, dzb, dgamma, dbeta, None, None)


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
        bott = cfg.EPIPOLAR.BOTTLENECK
        if bott != 1:                                                        # epipolar.py:56-61
            assert all(k in cfg.EPIPOLAR.PARAMETERIZED for k in ("z", "theta", "phi", "g"))
            assert not cfg.EPIPOLAR.ZRESIDUAL
        if "z" in cfg.EPIPOLAR.PARAMETERIZED:
            self.z = nn.Conv2d(nfeats // bott, nfeats, kernel_size=1, stride=1, padding=0, bias=True)   # epipolar.py:64
            self.bn = zeroinitBN(nfeats)                                                                # epipolar.py:65
        for name in ("theta", "phi", "g"):                                   # epipolar.py:66-71
            if name in cfg.EPIPOLAR.PARAMETERIZED:
                setattr(self, name, nn.Conv2d(nfeats, nfeats // bott, kernel_size=1, stride=1, padding=0, bias=True))
        self.prior = {}
        if cfg.EPIPOLAR.PRIOR:                                               # epipolar.py:73-80: a plain dict of
            for i in cfg.DATASETS.CAMERAS:                                   # Parameters, NOT registered (no state_dict keys)
                for j in cfg.DATASETS.CAMERAS:
                    if j != i:
                        self.prior[(i, j)] = nn.Parameter(torch.empty(self.sample_size, self.feat_h, self.feat_w).uniform_(0, 0.1))
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
        if e.REPROJECT_LOSS_WEIGHT != 0:
            # epipolar.py:257-261, 420-464: a 5-tuple return PoseResNet cannot unpack (resnet.py:385-387 takes four values) --
            # dead code for every backbone on the path; INTEGRATION.md "What raises"
            raise NotImplementedError("not on the MI355X path: REPROJECT_LOSS_WEIGHT != 0 (the reference's reprojection "
                                      "branch, unreachable through PoseResNet)")

    def _fused_mode(self, ref1=None, ref2=None) -> bool:
        """True when the call is the headline mode the fused HIP kernels implement."""
        e = self.cfg.EPIPOLAR
        return (e.ATTENTION == "avg" and e.SIMILARITY == "dot" and e.FIND_CORR == "feature" and ref1 is None and
                ref2 is None and not e.POOLING and not e.PRIOR and e.BOTTLENECK == 1 and
                not any(k in e.PARAMETERIZED for k in ("theta", "phi", "g")))

    def _attend_general(self, feat1, feat2, P1, P2, camera=None, other_camera=None, ref1=None, ref2=None):
        """`_attend_general_chunk` over ranges of pairs, so that the sampled K x C x H x W tensors (what the reference
        materialises per PAIR, epipolar.py:199-213) never exceed ~2 GB at once however large the batch is
        (keypoint_h36m_param.yaml at 32 frames x 4 views would otherwise hold 17 GB per sampled map)."""
        if self._general_kernel_applies(feat1, feat2, ref1, ref2, camera, other_camera):
            return self._attend_general_hip(feat1, feat2, P1, P2, camera, other_camera, ref1, ref2)
        _warn_slow_path(self.cfg)
        N, C, H, W = feat2.shape
        per_pair = 2 * self.sample_size * max(C, feat1.shape[1]) * H * W * 4 * (2 if torch.is_grad_enabled() else 1)
        step = max(1, int(amd_knob(self.cfg, "GENERAL_MODE_BYTES", 2 << 30)) // per_pair)
        if step >= N:
            return self._attend_general_chunk(feat1, feat2, P1, P2, camera, other_camera, ref1, ref2)
        sl = lambda t, a, b: None if t is None else t[a:b]
        parts = [self._attend_general_chunk(feat1[a:a + step], feat2[a:a + step], P1[a:a + step], P2[a:a + step],
                                            sl(camera, a, a + step), sl(other_camera, a, a + step), sl(ref1, a, a + step),
                                            sl(ref2, a, a + step)) for a in range(0, N, step)]
        return tuple(torch.cat([p[i] for p in parts]) for i in range(3))

    def _general_kernel_applies(self, feat1, feat2, ref1=None, ref2=None, camera=_ANY, other_camera=_ANY) -> bool:
        """True when the HIP general kernels compute this call: every branch of a12 / N4 -- theta / phi / g, BOTTLENECK,
        POOLING, PRIOR / PRIORMUL, SIMILARITY cos / prior, ATTENTION max, FIND_CORR rgb --, forward
        (`et_epipolar_forward_general`) and, since ABI 12, backward (`et_epipolar_backward_general`, through
        ops.GeneralAttend) incl. the prior tables' own gradient.  What is left for the chunked torch restatement below:
        EPIPOLAR_AMD.GENERAL_KERNEL False, CPU tensors, more than 512 similarity channels, an odd K with POOLING, prior
        tables whose row count is not K' -- and calls that are ill-formed in the reference too (PRIOR without camera ids,
        SIMILARITY prior without PRIOR), which fail there with the reference's own error."""
        e = self.cfg.EPIPOLAR
        if e.ATTENTION not in ("avg", "max") or (e.ATTENTION == "avg" and e.SIMILARITY not in ("dot", "cos", "prior")):
            return False
        if e.ATTENTION == "avg" and e.SIMILARITY == "prior" and not e.PRIOR:
            return False                               # (epipolar.py:219-224 passes camera ids only with PRIOR: a KeyError there)
        if e.FIND_CORR == "rgb" and (ref1 is None or ref2 is None):
            return False
        if not bool(amd_knob(self.cfg, "GENERAL_KERNEL", True)) or not feat2.is_cuda:
            return False
        if e.POOLING and self.sample_size % 2:
            return False
        if e.PRIOR:
            # the reference builds its prior tables with K rows (epipolar.py:70-80); with POOLING the similarities have K/2
            # (epipolar.py:200-224) and its own shapes disagree.  The kernel takes tables of K' = K/2 rows then; anything else,
            # or a call without camera ids (no table to pick), goes to the restatement and fails there with the reference's
            # own error instead of one from inside ops
            rows = self.sample_size // 2 if e.POOLING else self.sample_size
            if camera is None or other_camera is None or any(t.shape[0] != rows for t in self.prior.values()):
                return False
        c_sim = 3 if e.FIND_CORR == "rgb" else feat1.shape[1] // (e.BOTTLENECK if "theta" in e.PARAMETERIZED else 1)
        return c_sim <= 512

    def _attend_with_depth(self, feat1, feat2, P1, P2, depth):
        """An externally supplied `depth` (epipolar.py:101-104, 217-218): the given (N,K',H,W) weights REPLACE the similarity
        -- no mask, no soft-max --, `out` is their weighted sum of the (pooled) samples of other2 (ATTENTION avg, :243) or the
        arg-max sample (ATTENTION max, :225-235), and the z branch is skipped (:249).  One launch of the general kernel in its
        SIM_PRIOR mode (the weights take the prior's place), with the gradient of both the value map and the weights."""
        e = self.cfg.EPIPOLAR
        w = depth if torch.is_tensor(depth) else torch.stack(list(depth))
        rows = self.sample_size // 2 if e.POOLING else self.sample_size
        if w.dim() != 4 or w.shape[0] != feat2.shape[0] or w.shape[1] != rows or tuple(w.shape[2:]) != (self.feat_h, self.feat_w):
            raise ValueError("depth must be (N, %d, %d, %d) weights, got %s" % (rows, self.feat_h, self.feat_w, tuple(w.shape)))
        other2 = feat2 if "other2" in e.OTHER_GRAD else feat2.detach()                  # :147-150
        m2 = self.g(other2) if "g" in e.PARAMETERIZED else other2                       # :152-153
        with torch.no_grad():
            cam = self._cam(P1, P2, feat2.device)
        w = w.to(feat2)
        if not (feat2.is_cuda and bool(amd_knob(self.cfg, "GENERAL_KERNEL", True)) and not (e.POOLING and self.sample_size % 2)
                and rows <= 256):
            # what the general kernel does not take (the same limits as _general_kernel_applies: CPU tensors are refused by
            # ops, EPIPOLAR_AMD.GENERAL_KERNEL False, an odd K with POOLING, more than 256 weights per pixel): the reference's
            # op sequence in torch, as for every other branch
            return self._attend_with_depth_torch(m2, w, cam)
        unused = feat2.new_zeros((feat2.shape[0], 4, self.feat_h, self.feat_w))        # (q / similarity map: not read in this mode)
        mode = dict(prior_mul=False, cosine=False, attention_max=e.ATTENTION == "max", sim_prior=True)
        out, _, corr_pos = ops.GeneralAttend.apply(unused, unused, m2, cam, self.layer_spec(), bool(e.POOLING),
                                                   w.contiguous(), mode)
        return out, w, corr_pos

    def _attend_with_depth_torch(self, m2, w, cam):
        """The supplied-depth branch as torch ops (epipolar.py:199-213, 225-243 with `sim` = the given weights): the route of
        shapes the general kernel does not take.  `cam`: the per-pair algebra, computed once by the caller."""
        _warn_slow_path(self.cfg)
        e = self.cfg.EPIPOLAR
        K, H, W = self.sample_size, self.feat_h, self.feat_w
        N, c = m2.shape[0], m2.shape[1]
        with torch.no_grad():
            locs = ops.sample_locs(self.layer_spec(), cam)                    # (K,N,H,W,2)
        grid = locs.permute(1, 0, 2, 3, 4).reshape(N, K * H, W, 2)
        s2 = F.grid_sample(m2, grid, mode="bilinear", padding_mode="zeros",
                           align_corners=bool(amd_knob(self.cfg, "ALIGN_CORNERS", False))).view(N, c, K, H, W).permute(0, 2, 1, 3, 4)
        if e.POOLING:
            s2 = s2.reshape(N, 2, K // 2, c, H, W).max(1)[0]
        idx = w.argmax(1)
        with torch.no_grad():
            pos = torch.gather(locs.permute(1, 0, 2, 3, 4), 1, idx.view(N, 1, H, W, 1).expand(-1, -1, -1, -1, 2)).squeeze(1)
            if e.USE_CORRECT_NORMALIZE:
                corr_pos = torch.stack([(pos[..., 0] + 1) * (W - 1) / 2, (pos[..., 1] + 1) * (H - 1) / 2], -1)
            else:
                corr_pos = torch.stack([(pos[..., 0] + 1) * W / 2 - 0.5, (pos[..., 1] + 1) * H / 2 - 0.5], -1)
        if e.ATTENTION == "max":
            out = torch.gather(s2, 1, idx.view(N, 1, 1, H, W).expand(-1, -1, c, -1, -1)).squeeze(1)
        else:
            out = (s2 * w.unsqueeze(2)).sum(1)
        return out, w, corr_pos

    def _attend_general_hip(self, feat1, feat2, P1, P2, camera=None, other_camera=None, ref1=None, ref2=None):
        """The non-headline branches through the HIP general kernels: the 1x1 convolutions act on the maps
        (epipolar.py:138-153: torch / MIOpen GEMMs, with autograd), the kernel samples, pools, masks, soft-maxes and sums
        without materialising a K x C x H x W tensor (`ops.GeneralAttend`; prior / cosine / ATTENTION max:
        `ops.forward_general_nhwc`, forward only)."""
        e = self.cfg.EPIPOLAR
        if e.FIND_CORR == "rgb":                                                        # :131-136
            assert "other1" not in e.OTHER_GRAD and "phi" not in e.PARAMETERIZED
            m1, q = ref2.detach(), ref1
        else:
            other1 = feat2 if "other1" in e.OTHER_GRAD else feat2.detach()              # :138-141
            m1 = self.phi(other1) if "phi" in e.PARAMETERIZED else other1               # :142-143
            q = self.theta(feat1) if "theta" in e.PARAMETERIZED else feat1              # :144-145
        other2 = feat2 if "other2" in e.OTHER_GRAD else feat2.detach()                  # :147-150
        m2 = self.g(other2) if "g" in e.PARAMETERIZED else other2                       # :152-153
        with torch.no_grad():
            cam = self._cam(P1, P2, feat2.device)
        is_max = e.ATTENTION == "max"
        cos = e.ATTENTION == "avg" and e.SIMILARITY == "cos"
        sim_prior = e.ATTENTION == "avg" and e.SIMILARITY == "prior"
        prior = None
        if e.PRIOR and not is_max:                                                      # :288-289, :300-301, :308-309
            # the pairs' (camera, other camera) tables, stacked WITH autograd: the kernel returns d(stack), torch adds the
            # pairs of one camera pair into the table's .grad
            prior = torch.stack([self.prior[(int(a), int(b))].to(q) for a, b in zip(camera, other_camera)])
        mode = dict(prior_mul=bool(prior is not None and e.PRIORMUL and not sim_prior), cosine=cos, attention_max=is_max,
                    sim_prior=sim_prior)
        return ops.GeneralAttend.apply(q, m1, m2, cam, self.layer_spec(), bool(e.POOLING), prior, mode)

    def _attend_general_chunk(self, feat1, feat2, P1, P2, camera=None, other_camera=None, ref1=None, ref2=None):
        """The operator's non-headline branches, restated op for op from epipolar.py:131-247 and
        epipolar_similarity (:272-321) as batched GPU torch ops (autograd included).  Returns (out, attn, corr_pos)
        like `attend`; `out` is what the reference stacks at :247 (before z)."""
        e = self.cfg.EPIPOLAR
        K, H, W = self.sample_size, self.feat_h, self.feat_w
        if e.FIND_CORR == "rgb":                                              # epipolar.py:131-136
            assert ref1 is not None and ref2 is not None
            assert "other1" not in e.OTHER_GRAD and "phi" not in e.PARAMETERIZED
            other1, q = ref2.detach(), ref1
        else:
            other1 = feat2 if "other1" in e.OTHER_GRAD else feat2.detach()    # :138-141
            if "phi" in e.PARAMETERIZED:
                other1 = self.phi(other1)                                     # :142-143
            q = self.theta(feat1) if "theta" in e.PARAMETERIZED else feat1    # :144-145
        other2 = feat2 if "other2" in e.OTHER_GRAD else feat2.detach()        # :147-150
        if "g" in e.PARAMETERIZED:
            other2 = self.g(other2)                                           # :152-153
        N = feat2.shape[0]
        with torch.no_grad():
            cam = self._cam(P1, P2, feat2.device)
            locs = ops.sample_locs(self.layer_spec(), cam)                    # (K,N,H,W,2): the HIP geometry kernel
        align = bool(amd_knob(self.cfg, "ALIGN_CORNERS", False))

        def sample(src):                                                      # :199 / :210, + POOLING :200-202
            c = src.shape[1]
            grid = locs.permute(1, 0, 2, 3, 4).reshape(N, K * H, W, 2)        # one grid_sample per map for all K samples
            s_ = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=align)
            s_ = s_.view(N, c, K, H, W).permute(0, 2, 1, 3, 4)                # (N,K,C,H,W)
            if e.POOLING:
                s_ = s_.reshape(N, 2, K // 2, c, H, W).max(1)[0]              # view(stride, K // stride, ...).max(0)
            return s_

        s1 = sample(other1)
        s2 = s1 if other1 is other2 else sample(other2)
        Ks = s1.shape[1]
        qe = q.unsqueeze(1)
        if e.ATTENTION == "max":                                              # :282-286: always cosine, no mask / soft-max
            sim = F.cosine_similarity(qe.expand(-1, Ks, -1, -1, -1), s1, 2)
        else:
            if e.SIMILARITY == "prior":                                       # :288-289
                sim = torch.stack([self.prior[(int(a), int(b))].to(q) for a, b in zip(camera, other_camera)])
            else:
                if e.SIMILARITY == "cos":
                    sim = F.cosine_similarity(qe.expand(-1, Ks, -1, -1, -1), s1, 2)
                else:
                    sim = (s1 * qe).sum(2)                                    # :294-295
                sim = sim.masked_fill(sim == 0, -1e10)                        # :298
                pr = None
                if e.PRIOR:
                    pr = torch.stack([self.prior[(int(a), int(b))].to(sim) for a, b in zip(camera, other_camera)])
                    if not e.PRIORMUL:
                        sim = sim + pr                                        # :300-301
                if e.SOFTMAX_ENABLED:
                    sim = F.softmax(sim * e.SOFTMAXSCALE, 1)                  # :303-307
                    if e.PRIORMUL:
                        sim = sim * pr                                        # :308-309
                else:
                    sim = sim / Ks                                            # :310-311
        idx = sim.argmax(1)                                                   # (N,H,W)   :225 / :237
        with torch.no_grad():
            gl = locs.permute(1, 0, 2, 3, 4)                                  # (N,K,H,W,2)
            pos = torch.gather(gl, 1, idx.view(N, 1, H, W, 1).expand(-1, -1, -1, -1, 2)).squeeze(1)
            if self.cfg.EPIPOLAR.USE_CORRECT_NORMALIZE:                       # multiview.py:39-57
                corr_pos = torch.stack([(pos[..., 0] + 1) * (W - 1) / 2, (pos[..., 1] + 1) * (H - 1) / 2], -1)
            else:
                corr_pos = torch.stack([(pos[..., 0] + 1) * W / 2 - 0.5, (pos[..., 1] + 1) * H / 2 - 0.5], -1)
        if e.ATTENTION == "max":                                              # :232-235
            c2 = s2.shape[2]
            out = torch.gather(s2, 1, idx.view(N, 1, 1, H, W).expand(-1, -1, c2, -1, -1)).squeeze(1)
        else:
            out = (s2 * sim.unsqueeze(2)).sum(1)                              # :243
        return out, sim, corr_pos

    # ------------------------------------------------------------ debug geometry
    def _debug_geometry(self, cam: torch.Tensor):
        """The intermediates the reference returns with debug=True (epipolar.py:350-407, 416-417): rectangle intersections
        (N,HW,4,2), their validity mask (N,HW,4), the two chosen ones (N,HW,2,2), start (N,HW,2) and vec (1,N,HW,2), from the
        per-pair algebra `cam` (N,27).  Batched torch ops on the device -- visualisation aids: the sample locations
        themselves always come from the HIP geometry kernel (bit-equal to the reference), these are consistent with them
        to float32 rounding.  More than two valid intersections (a line through a corner; the reference raises there):
        the first two in edge order, as the kernels do."""
        spec = self.layer_spec()
        xs, ys, _ = spec.constants(cam.device)
        N, H, W = cam.shape[0], self.feat_h, self.feat_w
        gx, gy = xs.view(1, W).expand(H, W).reshape(-1), ys.view(H, 1).expand(H, W).reshape(-1)
        grid = torch.stack([gx, gy, torch.ones_like(gx)])                       # (3,HW)   epipolar.py:40-44
        X = cam[:, :12].view(N, 4, 3) @ grid                                    # :338
        x2 = cam[:, 12:24].view(N, 3, 4) @ X                                    # :340
        x2 = x2 / x2[:, 2:3]                                                    # :342
        e2 = cam[:, 24:27].view(N, 3, 1)
        l2 = torch.cross(e2.expand_as(x2), x2, dim=1).transpose(1, 2)           # (N,HW,3)  :350-352
        xmin, xmax, ymin, ymax, eps = float(xs[0]), float(xs[-1]), float(ys[0]), float(ys[-1]), self.epsilon
        den1 = torch.sign(l2[..., 1]) * l2[..., 1].abs().clamp_min(eps)
        den0 = torch.sign(l2[..., 0]) * l2[..., 0].abs().clamp_min(eps)
        by1 = -(xmin * l2[..., 0] + l2[..., 2]) / den1                          # :369-373
        by2 = -(xmax * l2[..., 0] + l2[..., 2]) / den1
        bx0 = -(ymin * l2[..., 1] + l2[..., 2]) / den0
        bx3 = -(ymax * l2[..., 1] + l2[..., 2]) / den0
        inter = torch.stack((bx0, by1, by2, bx3), -1).view(N, H * W, 4, 1).repeat(1, 1, 1, 2)   # :375-386
        inter[..., 0, 1], inter[..., 1, 0], inter[..., 2, 0], inter[..., 3, 1] = ymin, xmin, xmax, ymax
        mask = torch.stack(((bx0 >= xmin + eps) & (bx0 < xmax - eps), (by1 > ymin + eps) & (by1 <= ymax - eps),
                            (by2 >= ymin + eps) & (by2 < ymax - eps), (bx3 > xmin + eps) & (bx3 <= xmax - eps)), -1)   # :388-393
        few = mask.sum(-1) < 2
        mask = mask & ~few.unsqueeze(-1)                                        # :397
        rank = mask.cumsum(-1)
        first = (mask & (rank == 1)).float().argmax(-1)                         # first two valid in edge order
        second = (mask & (rank == 2)).float().argmax(-1)
        pick = lambda idx: torch.gather(inter, 2, idx.view(N, H * W, 1, 1).expand(-1, -1, 1, 2)).squeeze(2)
        valid = torch.stack((pick(first), pick(second)), 2)                     # (N,HW,2,2)  :402
        out = inter.new_tensor([xmin - 10000.0, ymin - 10000.0])
        valid = torch.where(few.view(N, H * W, 1, 1), out.view(1, 1, 1, 2), valid)   # :403
        start = valid[..., 0, :]
        vec = (valid[..., 1, :] - start).view(1, N, H * W, 2)                   # :405-407
        return inter, mask, valid, start, vec

    # --------------------------------------------------------------- forward
    # Optional (P_ref_cpu, P_src_cpu) of the batch the CALLING THREAD is about to run, handed over by a launcher that
    # still has the data loader's host copies (spares the device-to-host copy of GPU-resident matrices, camera.py).
    # Thread-local and scoped (`with Epipolar.host_matrices(...)`): nothing outlives the wrapped call, and threads
    # (nn.DataParallel replicas, a second model) never see each other's matrices.
    _host = threading.local()

    @classmethod
    @contextlib.contextmanager
    def host_matrices(cls, P_ref_cpu, P_src_cpu):
        prev = getattr(cls._host, "P", None)
        cls._host.P = (P_ref_cpu, P_src_cpu) if P_ref_cpu is not None and P_src_cpu is not None else None
        try:
            yield
        finally:
            cls._host.P = prev

    def _cam(self, P1, P2, device):
        host = getattr(self._host, "P", None)
        if host is not None and (tuple(host[0].shape) != tuple(P1.shape) or tuple(host[1].shape) != tuple(P2.shape)):
            host = None
        return self._cams.get(P1, P2, device, host=host)

    def attend(self, feat1, feat2, P1, P2):
        """The fused kernel only: (out, attn, corr_pos), `out` before the z branch."""
        cam = self._cam(P1, P2, feat1.device)
        return ops.EpipolarAttend.apply(feat1, feat2, cam, self.layer_spec())

    def _folded_z(self):
        """Eval-mode algebra of epipolar.py:250-253: bn(z(out)) [+ out] == out @ Wf^T + bf with
        Wf = diag(s) W [+ I], bf = s * b + (beta - mean * s), s = gamma / sqrt(var + eps)."""
        w = self.z.weight.view(self.z.out_channels, self.z.in_channels)
        scale = self.bn.weight * torch.rsqrt(self.bn.running_var + self.bn.eps)
        wf = w * scale[:, None]
        if self.cfg.EPIPOLAR.ZRESIDUAL:
            wf = wf + torch.eye(w.shape[0], dtype=w.dtype, device=w.device)
        bf = self.z.bias * scale + (self.bn.bias - self.bn.running_mean * scale)
        return wf.t().contiguous(), bf.contiguous()

    def _packed_z(self):
        """The folded z branch laid out for `ops.residual_gemm` (C == 256): (packed weight, bias), cached until a
        parameter or buffer of the branch changes (their version counters / storage)."""
        ps = (self.z.weight, self.z.bias, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        key = tuple((q.data_ptr(), q._version) for q in ps) + (bool(self.cfg.EPIPOLAR.ZRESIDUAL), float(self.bn.eps))
        hit = getattr(self, "_packed_z_cache", None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                wt, bf = self._folded_z()
                hit = (key, ops.residual_gemm_pack(wt.t().contiguous()), bf)
            self._packed_z_cache = hit
        return hit[1], hit[2]

    def _eval_fast_path(self, tensors):
        cfg = self.cfg
        if not bool(amd_knob(cfg, "FUSED_EPILOGUE", True)):
            return False
        has_z = "z" in cfg.EPIPOLAR.PARAMETERIZED
        # the folded epilogue consumes z / bn inside a HIP kernel and an out= GEMM: no autograd through it, so
        # it is taken only when NO parameter of the branch (and no input) asks for a gradient
        if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or
                                        (has_z and any(q.requires_grad for q in
                                                       list(self.z.parameters()) + list(self.bn.parameters())))):
            return False
        return not (has_z and self.bn.training)

    def _train_epilogue_applies(self, out) -> bool:
        """The z branch in TRAINING mode through the hand-written GEMM kernels (`_TrainEpilogue`): the 256-channel head with
        the layer's own batch norm (a SyncBatchNorm2d spans ranks: its statistics go through an all-reduce, parallel.py)."""
        return ("z" in self.cfg.EPIPOLAR.PARAMETERIZED and self.bn.training and out.is_cuda and out.dtype == torch.float32 and
                out.shape[1] == 256 and self.z.in_channels == 256 and self.z.out_channels == 256 and
                type(self.bn) in (zeroinitBN, nn.BatchNorm2d) and self.bn.affine and out.shape[0] * out.shape[2] * out.shape[3] > 1 and
                bool(amd_knob(self.cfg, "FUSED_TRAIN_EPILOGUE", True)))

    def _train_epilogue(self, out, feat1=None):
        bn = self.bn
        x, mean, var = _TrainEpilogue.apply(out, feat1, self.z.weight, self.z.bias, bn.weight, bn.bias, float(bn.eps),
                                            bool(self.cfg.EPIPOLAR.ZRESIDUAL))
        if bn.track_running_stats:                                   # BN.py:59-70: what F.batch_norm does to the buffers
            with torch.no_grad():
                bn.num_batches_tracked += 1
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                n = out.shape[0] * out.shape[2] * out.shape[3]
                bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(var * (n / (n - 1.0)), alpha=m)
        return x

    def _epilogue_torch(self, out, feat1=None):
        """Training path (batch statistics / autograd): two passes of the GEMM kernel for the 256-channel head
        (`_train_epilogue`), otherwise the reference's own op sequence."""
        if self._train_epilogue_applies(out):
            if feat1 is None:
                return self._train_epilogue(out), None
            # (forward_fused only consumes x = finalout + feat: one kernel adds the feature row as well)
            return None, self._train_epilogue(out, feat1)
        finalout = out
        if "z" in self.cfg.EPIPOLAR.PARAMETERIZED:
            finalout = self.bn(self.z(out))                                  # epipolar.py:250-251
            if self.cfg.EPIPOLAR.ZRESIDUAL:
                finalout = finalout + out                                    # epipolar.py:253
        return finalout, (finalout + feat1 if feat1 is not None else None)  # resnet.py:388

    def forward(self, feat1, feat2, P1, P2, depth=None, camera=None, other_camera=None, ref1=None, ref2=None):
        """Same contract as the reference (epipolar.py:82-269):
        feat1, feat2: N x C x H x W ; P1, P2: N x 3 x 4
        returns (finalout, corr_pos[N,H,W,2], depth[N,K,H,W], sample_locs | None)."""
        self._check_mode(depth, ref1, ref2)
        if depth is not None:
            out, attn, corr_pos = self._attend_with_depth(feat1, feat2, P1, P2, depth)
            sample_locs = None
            if self.debug or self.cfg.VIS.EPIPOLAR_LINE:
                sample_locs = ops.sample_locs(self.layer_spec(), self._cam(P1, P2, feat2.device))
            if self.debug:
                return (out, corr_pos, attn, sample_locs) + self._debug_geometry(self._cam(P1, P2, feat2.device))
            return out, corr_pos, attn, (sample_locs.transpose(0, 1) if sample_locs is not None else None)
        fused = self._fused_mode(ref1, ref2)
        if fused:
            out, attn, corr_pos = self.attend(feat1, feat2, P1, P2)
        else:
            out, attn, corr_pos = self._attend_general(feat1, feat2, P1, P2, camera, other_camera, ref1, ref2)
        if fused and self._eval_fast_path((feat1, feat2)) and "z" in self.cfg.EPIPOLAR.PARAMETERIZED:
            o = ops.to_nhwc(out)
            if o.shape[-1] == 256:    # one HBM-bound kernel: bias + out . Wf^T (split-fp16 MFMA)
                packed, bf = self._packed_z()
                finalout = ops.residual_gemm(o, packed, bf).permute(0, 3, 1, 2)
            else:                     # one library GEMM with the bias in its epilogue
                wt, bf = self._folded_z()
                finalout = torch.addmm(bf, o.reshape(-1, o.shape[-1]), wt).view_as(o).permute(0, 3, 1, 2)
        else:
            finalout, _ = self._epilogue_torch(out)
        sample_locs = None
        if self.debug:
            # epipolar.py:264-265: the 9-tuple of the visualisers -- sample_locs untransposed (K,N,H,W,2) + the geometry
            cam = self._cam(P1, P2, feat1.device)
            return (finalout, corr_pos, attn, ops.sample_locs(self.layer_spec(), cam)) + self._debug_geometry(cam)
        if self.cfg.VIS.EPIPOLAR_LINE:
            cam = self._cam(P1, P2, feat1.device)
            sample_locs = ops.sample_locs(self.layer_spec(), cam).transpose(0, 1)   # (K,N,H,W,2) -> epipolar.py:267
        return finalout, corr_pos, attn, sample_locs

    def forward_fused(self, feat1, feat2, P1, P2, camera=None, other_camera=None):
        """forward + `ret + feat` (resnet.py:388).  In eval mode the whole epilogue is ONE GEMM, x = feat + bf + out @ Wf^T:
        for the 256-channel head on maps up to 96 x 96 (K <= 64) a third GEMM INSIDE the persistent forward kernel
        (`ops.forward_fused_nhwc`: `out` never goes through HBM), for the other 256-channel shapes `ops.residual_gemm` behind
        the forward; other widths: the fused kernel emits feat + bf and a library GEMM accumulates into it.
        Returns (x, corr_pos, depth, None)."""
        self._check_mode(None, None, None)
        if not self._fused_mode():
            fin, corr_pos, attn, _ = self.forward(feat1, feat2, P1, P2, camera=camera, other_camera=other_camera)
            return fin + feat1, corr_pos, attn, None                         # resnet.py:388
        if not self._eval_fast_path((feat1, feat2)):
            out, attn, corr_pos = self.attend(feat1, feat2, P1, P2)
            _, x = self._epilogue_torch(out, feat1)
            return x, corr_pos, attn, None
        cam = self._cam(P1, P2, feat1.device)
        ref, src = ops.to_nhwc(feat1), ops.to_nhwc(feat2)
        if "z" in self.cfg.EPIPOLAR.PARAMETERIZED and ref.shape[-1] == 256:
            # x = feat + bf + out . Wf^T in ONE kernel behind the fused forward (no res_base round trip through HBM)
            packed, bf = self._packed_z()
            spec = self.layer_spec()
            if bool(amd_knob(self.cfg, "FUSED_GEMM3", True)) and ops.fused_layer_applies(spec, 256, ref.shape[0]):
                # ... inside the persistent kernel, as a third GEMM on the tile's out rows (ops.forward_fused_nhwc)
                x, attn, corr_pos = ops.forward_fused_nhwc(spec, ref, src, cam, packed, bf)
            else:
                out, attn, corr_pos = ops.forward_nhwc(spec, ref, src, cam)
                x = ops.residual_gemm(out, packed, bf, ref)
        elif "z" in self.cfg.EPIPOLAR.PARAMETERIZED:
            wt, bf = self._folded_z()
            out, attn, corr_pos, base = ops.forward_nhwc(self.layer_spec(), ref, src, cam, res_bias=bf,
                                                         want_res_base=True)
            c = out.shape[-1]
            x = torch.addmm(base.view(-1, c), out.view(-1, c), wt, out=base.view(-1, c)).view_as(out)
        else:
            out, attn, corr_pos = ops.forward_nhwc(self.layer_spec(), ref, src, cam)
            _, x = ops.residual_epilogue(ref, out, None, None, None, False, True)
        return x.permute(0, 3, 1, 2), corr_pos, attn, None
