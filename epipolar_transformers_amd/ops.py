"""Torch-facing wrappers of the C ABI (torch is plumbing: device memory, the
current HIP stream and autograd bookkeeping; all arithmetic of the path is in
the HIP library).  Feature tensors are logical NCHW like the reference's; their
memory is NHWC (torch.channels_last), converted once with our own kernel when a
caller hands NCHW-contiguous memory."""
from __future__ import annotations

import ctypes
import threading
import weakref
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import EtLayerDesc


@dataclass
class LayerSpec:
    """The static part of an EtLayerDesc + the three small constant arrays the
    reference builds in Epipolar.__init__ (epipolar.py:22-54)."""

    H: int
    W: int
    K: int
    downsample: float = 4.0
    image_resize: float = 1.0
    predict_resize: float = 1.0
    correct_normalize: bool = True
    align_corners: bool = False
    softmax_scale: float = 0.125
    softmax_enabled: bool = True
    eps: float = 0.001
    src_grad_mask: int = 3
    variant: int = 0

    def __post_init__(self):
        ds = self.downsample
        # pix2coord (multiview.py:154-157) * resize factors, same float32 op order as epipolar.py:35-38
        y = torch.arange(0, self.H, dtype=torch.float)
        x = torch.arange(0, self.W, dtype=torch.float)
        y = (y * ds + ds / 2.0 - 0.5) * self.image_resize * self.predict_resize
        x = (x * ds + ds / 2.0 - 0.5) * self.image_resize * self.predict_resize
        self.xs, self.ys = x.contiguous(), y.contiguous()
        # torch.range(0, 1, 1/(K-1)) (epipolar.py:54): start + i*step in double, rounded to float32
        step = 1.0 / (self.K - 1)
        self.steps = torch.from_numpy((np.arange(self.K, dtype=np.float64) * step).astype(np.float32))
        self._dev = {}

    def constants(self, device):
        """xs, ys, steps resident on `device` (uploaded once per device)."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(t.to(device) for t in (self.xs, self.ys, self.steps))
        return self._dev[key]

    def desc(self, N: int, C: int) -> EtLayerDesc:
        return EtLayerDesc(
            N=N, C=C, H=self.H, W=self.W, K=self.K,
            xmin=float(self.xs[0]), ymin=float(self.ys[0]), xmax=float(self.xs[-1]), ymax=float(self.ys[-1]),
            eps=self.eps, downsample=float(self.downsample), image_resize=float(self.image_resize),
            predict_resize=float(self.predict_resize), correct_normalize=int(self.correct_normalize),
            align_corners=int(self.align_corners), softmax_scale=float(self.softmax_scale),
            softmax_enabled=int(self.softmax_enabled), src_grad_mask=int(self.src_grad_mask),
            variant=int(self.variant))


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _lib.EpipolarAmdError(
            "%s is on %s: the epipolar hot path runs on the GPU only (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


POISON_OUTPUTS = False      # tests set this: every output buffer starts as NaN, so that a kernel that leaves part of its
                            # output unwritten cannot hide behind the (correct) values a recycled allocation still holds


def _empty(shape, like=None, device=None, dtype=torch.float32):
    if like is not None:
        shape, device, dtype = like.shape, like.device, like.dtype
    t = torch.empty(tuple(shape), dtype=dtype, device=device)
    if POISON_OUTPUTS and t.is_floating_point():
        t.fill_(float("nan"))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Logical (N,C,H,W) -> contiguous (N,H,W,C) memory.  Zero-copy when x is
    already channels_last; otherwise one pass of et_nchw_to_nhwc."""
    _require_gpu(x, "feature map")
    n, c, h, w = x.shape
    perm = x.permute(0, 2, 3, 1)
    if perm.is_contiguous():
        return perm
    src = x.contiguous()
    dst = _empty((n, h, w, c), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().et_nchw_to_nhwc(n, c, h, w, _ptr(src), _ptr(dst), _stream(x)), "et_nchw_to_nhwc")
    return dst


def to_nchw_contiguous(x_nhwc: torch.Tensor) -> torch.Tensor:
    """(N,H,W,C) memory -> NCHW-contiguous tensor via et_nhwc_to_nchw."""
    _require_gpu(x_nhwc, "feature map")
    n, h, w, c = x_nhwc.shape
    dst = _empty((n, c, h, w), device=x_nhwc.device, dtype=x_nhwc.dtype)
    with torch.cuda.device(x_nhwc.device):
        _lib.check(_lib.load().et_nhwc_to_nchw(n, c, h, w, _ptr(x_nhwc.contiguous()), _ptr(dst), _stream(x_nhwc)),
                   "et_nhwc_to_nchw")
    return dst


def sample_locs(spec: LayerSpec, cam: torch.Tensor) -> torch.Tensor:
    """grid2sample_locs (epipolar.py:323-418): (K,N,H,W,2)."""
    _require_gpu(cam, "cam")
    n = cam.shape[0]
    xs, ys, steps = spec.constants(cam.device)
    out = _empty((spec.K, n, spec.H, spec.W, 2), device=cam.device)
    d = spec.desc(n, 4)
    with torch.cuda.device(cam.device):
        _lib.check(_lib.load().et_sample_locs(ctypes.byref(d), _ptr(xs), _ptr(ys), _ptr(steps), _ptr(cam), _ptr(out),
                                              _stream(cam)), "et_sample_locs")
    return out


def _general_flags(pooling=False, prior_mul=False, cosine=False, attention_max=False, sim_prior=False) -> int:
    return ((_lib.ET_GENERAL_POOLING if pooling else 0) | (_lib.ET_GENERAL_PRIOR_MUL if prior_mul else 0) |
            (_lib.ET_GENERAL_COSINE if cosine else 0) | (_lib.ET_GENERAL_ATTENTION_MAX if attention_max else 0) |
            (_lib.ET_GENERAL_SIM_PRIOR if sim_prior else 0))


def forward_general_nhwc(spec: LayerSpec, q: torch.Tensor, map_sim: torch.Tensor, map_val: torch.Tensor, cam: torch.Tensor,
                         prior: torch.Tensor = None, pooling=False, prior_mul=False, cosine=False, attention_max=False,
                         want_attn=True, want_corr=True, sim_prior=False):
    """The operator's parameterised / pooled / prior branches as ONE kernel (et_epipolar_forward_general; forward only).
    q, map_sim: (N,H,W,Cs); map_val: (N,H,W,Cv), channels last, contiguous; prior: (N,K',H,W) or None, K' = K/2 with
    `pooling` (epipolar.py:200-202), else K.  Returns out (N,H,W,Cv), attn (N,K',H,W)|None, corr_pos (N,H,W,2)|None."""
    for t, nm in ((q, "q"), (map_sim, "map_sim"), (map_val, "map_val"), (cam, "cam")):
        _require_gpu(t, nm)
    n, h, w, cs = q.shape
    cv = map_val.shape[-1]
    if (h, w) != (spec.H, spec.W) or map_sim.shape != q.shape or map_val.shape[:3] != q.shape[:3]:
        raise ValueError("maps %s / %s / %s do not match the layer's %dx%d" %
                         (tuple(q.shape), tuple(map_sim.shape), tuple(map_val.shape), spec.H, spec.W))
    if cam.shape != (n, _lib.ET_CAM_STRIDE) or not cam.is_contiguous():
        raise ValueError("cam must be a contiguous (N,%d) tensor" % _lib.ET_CAM_STRIDE)
    if not (q.is_contiguous() and map_sim.is_contiguous() and map_val.is_contiguous()):
        raise ValueError("q / map_sim / map_val must be contiguous (N,H,W,C) tensors")
    ks = spec.K // 2 if pooling else spec.K
    if prior is not None:
        _require_gpu(prior, "prior")
        if tuple(prior.shape) != (n, ks, h, w) or not prior.is_contiguous():
            raise ValueError("prior must be (N,K',H,W) = %s, got %s" % ((n, ks, h, w), tuple(prior.shape)))
    xs, ys, steps = spec.constants(q.device)
    out = _empty((n, h, w, cv), device=q.device)
    attn = _empty((n, ks, h, w), device=q.device) if want_attn else None
    corr = _empty((n, h, w, 2), device=q.device) if want_corr else None
    flags = _general_flags(pooling, prior_mul, cosine, attention_max, sim_prior)
    d = spec.desc(n, 4)
    with torch.cuda.device(q.device):
        _lib.check(_lib.load().et_epipolar_forward_general(
            ctypes.byref(d), _ptr(xs), _ptr(ys), _ptr(steps), _ptr(cam), _ptr(q), _ptr(map_sim), _ptr(map_val),
            _ptr(prior) if prior is not None else None, cs, cv, flags, _ptr(out), _ptr(attn) if want_attn else None,
            _ptr(corr) if want_corr else None, _stream(q)), "et_epipolar_forward_general")
    return out, attn, corr


def backward_general_nhwc(spec: LayerSpec, q, map_sim, map_val, cam, grad_out, pooling=False, need_sim=True, need_val=True,
                          prior=None, prior_mul=False, cosine=False, attention_max=False, sim_prior=False, need_prior=False):
    """Backward of forward_general_nhwc, every branch: returns (grad_q, grad_map_sim | None, grad_map_val | None,
    grad_prior | None), NHWC maps, grad_prior (N,K',H,W).  The map gradients are accumulated with float atomics
    (reproducible to rounding only)."""
    for t, nm in ((q, "q"), (map_sim, "map_sim"), (map_val, "map_val"), (cam, "cam"), (grad_out, "grad_out")):
        _require_gpu(t, nm)
    n, h, w, cs = q.shape
    cv = map_val.shape[-1]
    if tuple(grad_out.shape) != (n, h, w, cv) or not grad_out.is_contiguous():
        raise ValueError("grad_out must be a contiguous (N,H,W,Cv) tensor")
    if not (q.is_contiguous() and map_sim.is_contiguous() and map_val.is_contiguous()):
        raise ValueError("q / map_sim / map_val must be contiguous (N,H,W,C) tensors")
    ks = spec.K // 2 if pooling else spec.K
    if prior is not None:
        _require_gpu(prior, "prior")
        if tuple(prior.shape) != (n, ks, h, w) or not prior.is_contiguous():
            raise ValueError("prior must be (N,K',H,W) = %s, got %s" % ((n, ks, h, w), tuple(prior.shape)))
    xs, ys, steps = spec.constants(q.device)
    gq = _empty(None, like=q)
    gsim = torch.zeros_like(map_sim) if need_sim else None
    gval = torch.zeros_like(map_val) if need_val else None
    gprior = _empty(None, like=prior) if (need_prior and prior is not None) else None
    d = spec.desc(n, 4)
    flags = _general_flags(pooling, prior_mul, cosine, attention_max, sim_prior)
    with torch.cuda.device(q.device):
        _lib.check(_lib.load().et_epipolar_backward_general(
            ctypes.byref(d), _ptr(xs), _ptr(ys), _ptr(steps), _ptr(cam), _ptr(q), _ptr(map_sim), _ptr(map_val),
            _ptr(prior), _ptr(grad_out), cs, cv, flags, _ptr(gq), _ptr(gsim), _ptr(gval), _ptr(gprior),
            _stream(q)), "et_epipolar_backward_general")
    return gq, gsim, gval, gprior


class GeneralAttend(torch.autograd.Function):
    """The operator's non-headline branches with autograd (every one since ABI 12): logical NCHW in and out, `prior` the
    (N,K',H,W) stack of the pairs' prior tables or None; `attn` and `corr_pos` without gradient (as EpipolarAttend).
    `mode`: dict(pooling, prior_mul, cosine, attention_max, sim_prior) of bools."""

    @staticmethod
    def forward(ctx, q, map_sim, map_val, cam, spec: LayerSpec, pooling, prior=None, mode=None):
        mode = dict(mode or {}, pooling=bool(pooling))
        qn, m1, m2 = to_nhwc(q), to_nhwc(map_sim), to_nhwc(map_val)
        pr = None if prior is None else prior.contiguous()
        out, attn, corr = forward_general_nhwc(spec, qn, m1, m2, cam, prior=pr, **mode)
        ctx.spec, ctx.mode, ctx.has_prior = spec, mode, pr is not None
        ctx.save_for_backward(*((qn, m1, m2, cam) + ((pr,) if pr is not None else ())))
        ctx.mark_non_differentiable(attn, corr)
        return out.permute(0, 3, 1, 2), attn, corr

    @staticmethod
    def backward(ctx, grad_out, _ga, _gc):
        qn, m1, m2, cam = ctx.saved_tensors[:4]
        pr = ctx.saved_tensors[4] if ctx.has_prior else None
        gq, gs, gv, gp = backward_general_nhwc(ctx.spec, qn, m1, m2, cam, to_nhwc(grad_out), need_sim=ctx.needs_input_grad[1],
                                               need_val=ctx.needs_input_grad[2], prior=pr,
                                               need_prior=ctx.has_prior and ctx.needs_input_grad[6], **ctx.mode)
        nchw = lambda t: None if t is None else t.permute(0, 3, 1, 2)
        return nchw(gq) if ctx.needs_input_grad[0] else None, nchw(gs), nchw(gv), None, None, None, gp, None


_TILE_BITS = (_lib.ET_VARIANT_TILE_SPLIT | _lib.ET_VARIANT_TILE_CLASSIC |
              _lib.ET_VARIANT_WS_SETPRIO | _lib.ET_VARIANT_TILE_EXACT | _lib.ET_VARIANT_WS_BAND | _lib.ET_VARIANT_BWD_SPLIT_IN_PLACE)   # bits that tune the tile path instead of leaving it


def forward_nhwc(spec: LayerSpec, ref: torch.Tensor, src: torch.Tensor, cam: torch.Tensor,
                 want_attn=True, want_corr=True, res_bias=None, want_res_base=False, workspace=None):
    """ref/src: (N,H,W,C) contiguous.  Returns out (N,H,W,C), attn (N,K,H,W)|None, corr_pos (N,H,W,2)|None
    [, res_base (N,H,W,C) = ref + res_bias when want_res_base].  `workspace`: a caller-owned uint8 tensor for the
    tile path (see tile_workspace / tile_stats) instead of the cached per-(device, stream) one."""
    for t, nm in ((ref, "feat_ref"), (src, "feat_src"), (cam, "cam")):
        _require_gpu(t, nm)
    n, h, w, c = ref.shape
    if (h, w) != (spec.H, spec.W) or src.shape != ref.shape:
        raise ValueError("feature maps %s / %s do not match the layer's %dx%d" %
                         (tuple(ref.shape), tuple(src.shape), spec.H, spec.W))
    if cam.shape != (n, _lib.ET_CAM_STRIDE) or not cam.is_contiguous():
        raise ValueError("cam must be a contiguous (N,%d) tensor" % _lib.ET_CAM_STRIDE)
    assert ref.is_contiguous() and src.is_contiguous()
    xs, ys, steps = spec.constants(ref.device)
    out = _empty(None, like=ref)
    attn = _empty((n, spec.K, h, w), device=ref.device) if want_attn else None
    corr = _empty((n, h, w, 2), device=ref.device) if want_corr else None
    base = _empty(None, like=ref) if want_res_base else None
    if res_bias is not None:
        assert want_res_base and res_bias.is_cuda and res_bias.numel() == c and res_bias.is_contiguous()
    d = spec.desc(n, c)
    lib = _lib.load()
    # variant 0 (the library default) takes the MFMA tile formulation wherever it applies (C == 256 head);
    # any explicit variant bit selects the per-pixel kernels
    ws_bytes = int(lib.et_epipolar_forward_workspace_bytes(ctypes.byref(d))) \
        if (d.variant & ~_TILE_BITS) == 0 else 0
    with torch.cuda.device(ref.device):
        if ws_bytes > 0:
            ws = workspace if workspace is not None else _workspace(ref.device, ws_bytes, "fwd")
            if ws.numel() < ws_bytes or ws.device != ref.device:
                raise ValueError("workspace of %d bytes on %s: need %d on %s" % (ws.numel(), ws.device, ws_bytes, ref.device))
            _lib.check(lib.et_epipolar_forward_tiled(ctypes.byref(d), _ptr(xs), _ptr(ys), _ptr(steps), _ptr(cam),
                                                     _ptr(ref), _ptr(src), _ptr(out), _ptr(attn), _ptr(corr),
                                                     _ptr(res_bias), _ptr(base), _ptr(ws), ctypes.c_size_t(ws_bytes),
                                                     _stream(ref)),
                       "et_epipolar_forward_tiled")
            if POISON_OUTPUTS:          # (test suite: surface a device-side fault at the call that caused it)
                check_tile_errors(workspace=ws)
        else:
            _lib.check(lib.et_epipolar_forward(ctypes.byref(d), _ptr(xs), _ptr(ys), _ptr(steps), _ptr(cam),
                                               _ptr(ref), _ptr(src), _ptr(out), _ptr(attn), _ptr(corr),
                                               _ptr(res_bias), _ptr(base), _stream(ref)),
                       "et_epipolar_forward")
    if want_res_base:
        return out, attn, corr, base
    return out, attn, corr


def fused_layer_applies(spec: LayerSpec, c: int, n: int = 1) -> bool:
    """True when et_epipolar_forward_fused covers this shape (the warp-specialised tile kernel: C == 256, maps up to
    96 x 96, K <= 64, soft-max on, no variant bit that leaves that kernel)."""
    allowed = _lib.ET_VARIANT_TILE_SPLIT | _lib.ET_VARIANT_WS_SETPRIO | _lib.ET_VARIANT_WS_BAND
    if not (c == 256 and 2 <= spec.W <= 96 and spec.H <= 96 and spec.K <= 64 and spec.softmax_enabled and
            (spec.variant & ~allowed) == 0):
        return False
    d = spec.desc(n, c)
    return int(_lib.load().et_epipolar_forward_workspace_bytes(ctypes.byref(d))) > 0


def forward_fused_nhwc(spec: LayerSpec, ref: torch.Tensor, src: torch.Tensor, cam: torch.Tensor, packed: torch.Tensor,
                       bias: torch.Tensor, want_attn=True, want_corr=True, want_out=False, workspace=None):
    """The eval-mode layer as one data kernel (et_epipolar_forward_fused): returns x = ref + bias + out @ Wf^T (N,H,W,C),
    attn (N,K,H,W)|None, corr_pos (N,H,W,2)|None [, out when want_out].  `packed`: residual_gemm_pack(Wf)."""
    for t, nm in ((ref, "feat_ref"), (src, "feat_src"), (cam, "cam"), (bias, "bias")):
        _require_gpu(t, nm)
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise TypeError("packed must be the uint8 device buffer residual_gemm_pack returns")
    n, h, w, c = ref.shape
    if (h, w) != (spec.H, spec.W) or src.shape != ref.shape or c != 256:
        raise ValueError("feature maps %s / %s do not match the layer's %dx%d x 256" % (tuple(ref.shape), tuple(src.shape), spec.H, spec.W))
    if cam.shape != (n, _lib.ET_CAM_STRIDE) or not cam.is_contiguous():
        raise ValueError("cam must be a contiguous (N,%d) tensor" % _lib.ET_CAM_STRIDE)
    need = int(_lib.load().et_residual_gemm_packed_bytes())
    if packed.numel() < need or not packed.is_contiguous() or packed.device != ref.device:
        raise ValueError("packed holds %d bytes on %s: the kernel reads %d on %s (ops.residual_gemm_pack)" %
                         (packed.numel(), packed.device, need, ref.device))
    if not (ref.is_contiguous() and src.is_contiguous()):
        raise ValueError("feat_ref / feat_src must be contiguous (N,H,W,C) tensors")
    if bias.numel() != c or not bias.is_contiguous() or bias.device != ref.device:
        raise ValueError("bias must be a contiguous float32 vector of %d values on %s" % (c, ref.device))
    if workspace is not None and (workspace.device != ref.device or workspace.dtype != torch.uint8):
        raise ValueError("workspace must be a uint8 tensor on %s" % (ref.device,))
    xs, ys, steps = spec.constants(ref.device)
    x = _empty(None, like=ref)
    # `out`: requested -> a fresh tensor; otherwise scratch for the rows of overflow tiles only (normally none is written),
    # one buffer per (device, stream) like the workspace: all use is stream-ordered
    out = _empty(None, like=ref) if want_out else _workspace(ref.device, ref.numel() * 4, "fwd_out_scratch").view(torch.float32)[:ref.numel()]
    attn = _empty((n, spec.K, h, w), device=ref.device) if want_attn else None
    corr = _empty((n, h, w, 2), device=ref.device) if want_corr else None
    d = spec.desc(n, c)
    lib = _lib.load()
    ws_bytes = int(lib.et_epipolar_forward_workspace_bytes(ctypes.byref(d)))
    with torch.cuda.device(ref.device):
        ws = workspace if workspace is not None else _workspace(ref.device, max(ws_bytes, 1), "fwd")
        _lib.check(lib.et_epipolar_forward_fused(ctypes.byref(d), _ptr(xs), _ptr(ys), _ptr(steps), _ptr(cam), _ptr(ref),
                                                 _ptr(src), _ptr(packed), _ptr(bias), _ptr(x), _ptr(attn), _ptr(corr),
                                                 _ptr(out), 1 if want_out else 0, _ptr(ws), ctypes.c_size_t(ws.numel()),
                                                 _stream(ref)), "et_epipolar_forward_fused")
        if POISON_OUTPUTS:
            check_tile_errors(workspace=ws)
        else:
            _poll_tile_error(ws)
    return (x, attn, corr, out) if want_out else (x, attn, corr)


_workspaces = {}
_workspaces_lock = threading.Lock()


def _prune_dead_threads_locked():
    """Entries of host threads that no longer exist (nn.DataParallel starts new forward threads on every call): dropped, so
    the cache does not grow with the number of calls.  Caller holds _workspaces_lock."""
    alive = {t.ident for t in threading.enumerate()}
    for key in [k for k in _workspaces if k[4] not in alive]:
        del _workspaces[key]


def _workspace(device, nbytes: int, tag: str = "bwd") -> torch.Tensor:
    """Device scratch the library asks for (it allocates nothing itself): the pixel order / overflow list / tile
    statistics of the tile kernels (tag "fwd", 2.3 MB at Config 2) and the coefficient entries of the gather-form
    backward (tag "bwd", 3.8 GB at Config 2 -- sized for 288 GB of HBM).  One buffer per (device, stream, tag, host
    thread), grown on demand: all use is stream-ordered on the stream it is keyed by, and two host threads that launch
    on the SAME device and stream (nn.DataParallel replicas pinned to one GPU) get buffers of their own -- the cache is
    the package's only process-wide mutable state (guarded by a lock; entries of threads that have exited are dropped
    whenever a buffer is created), and no two callers ever write the same entry.  release_workspaces() drops them."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(dev).cuda_stream, tag, threading.get_ident())
    with _workspaces_lock:
        buf = _workspaces.get(key)
        if buf is None or buf.numel() < nbytes:
            _prune_dead_threads_locked()
            # zero-initialised ONCE (include/epipolar_amd.h): the tile forward keeps a sticky error word in it
            _workspaces[key] = buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    return buf


_TILE_ERROR_OFFSET = 4      # bytes: word 1 of the workspace header (et_epipolar_forward_workspace_error_offset: shape-independent)


def _read_tile_error(buf: torch.Tensor) -> int:
    base = (-buf.data_ptr()) % 256                        # the library aligns the base up to 256 bytes
    return int(buf[base + _TILE_ERROR_OFFSET: base + _TILE_ERROR_OFFSET + 4].view(torch.int32).item())


_TILE_ERROR_TEXT = ("the tile forward reported device-side error bits 0x%x (bit 1: a matrix wave of et_epipolar_forward_fused gave "
                    "up at the barrier in front of its third GEMM; the results of that call are invalid)")


def _clear_tile_error(buf: torch.Tensor):
    base = (-buf.data_ptr()) % 256
    buf[base + _TILE_ERROR_OFFSET: base + _TILE_ERROR_OFFSET + 4].zero_()


def check_tile_errors(spec: "LayerSpec" = None, n: int = None, c: int = 256, workspace: torch.Tensor = None, reset: bool = True):
    """Read the sticky error word the tile forward keeps in its workspace (the library never synchronises, so this is
    where a device-side fault surfaces: it synchronises).  Without arguments: every cached forward workspace; with
    `workspace`: that one (spec / n / c are accepted for compatibility; the word sits in the workspace header, at the same
    offset for every shape).  Raises EpipolarAmdError; with `reset` (default) the word is cleared once it has been
    reported, so that later calls are judged on their own.  Called after every tile forward when POISON_OUTPUTS is set
    (the test suite) and once per bench.py run; the product path (forward_fused_nhwc) polls the word without
    synchronising, see _poll_tile_error."""
    if workspace is not None:
        bufs = [workspace]
    else:
        with _workspaces_lock:      # (other threads may insert meanwhile: iterate over a snapshot)
            bufs = [buf for key, buf in list(_workspaces.items()) if key[3] == "fwd"]
    for buf in bufs:
        if buf is None or buf.numel() < 512:
            continue
        word = _read_tile_error(buf)
        if word:
            if reset:
                _clear_tile_error(buf)
            raise _lib.EpipolarAmdError(_TILE_ERROR_TEXT % word)


_TILE_ERROR_POLL_EVERY = 16


class _ErrorProbe:
    """Per-workspace state of the asynchronous poll: calls since the last probe, ONE pinned host word (allocated with the
    probe, not per poll) and the event behind the copy in flight (None: no copy pending)."""
    __slots__ = ("calls", "host", "event")

    def __init__(self):
        self.calls, self.host, self.event = 0, None, None


# The probe hangs on the workspace TENSOR itself (an attribute: a freed workspace takes its probe with it, and a new tensor that
# happens to get the same address starts from a clean state; a dictionary keyed by tensors would compare them element-wise).
_PROBE_ATTR = "_et_error_probe"


def _poll_tile_error(buf: torch.Tensor):
    """The product path's check of the sticky error word WITHOUT a host synchronisation: every 16th one-kernel forward on a
    workspace enqueues a 4-byte device-to-host copy of the word behind the kernel (pinned memory, non-blocking) and an
    event; a later call that finds the event complete reads the host copy and raises (and clears the word) if a wave
    of an EARLIER call gave up at its barrier.  A fault therefore surfaces at most ~32 calls late instead of never;
    check_tile_errors() is the immediate, synchronising form.
    Nothing happens while the stream is being captured into a graph: a copy or an event record would become part of the
    graph, and querying an event is not allowed during a global-mode capture (it invalidates the capture).  A replayed graph
    is therefore not polled -- call check_tile_errors() after a batch of replays."""
    if torch.cuda.is_current_stream_capturing():
        return
    st = getattr(buf, _PROBE_ATTR, None)
    if st is None:
        st = _ErrorProbe()
        setattr(buf, _PROBE_ATTR, st)
    if st.event is not None and st.event.query():
        word = int(st.host.item())
        st.event = None
        if word:
            _clear_tile_error(buf)
            raise _lib.EpipolarAmdError((_TILE_ERROR_TEXT % word) + " -- reported by the asynchronous poll: the faulty call is "
                                        "one of the last %d on this workspace" % (2 * _TILE_ERROR_POLL_EVERY))
    st.calls += 1
    if st.event is None and st.calls >= _TILE_ERROR_POLL_EVERY:
        st.calls = 0
        base = (-buf.data_ptr()) % 256
        if st.host is None:
            st.host = torch.empty(1, dtype=torch.int32).pin_memory()
        st.host.copy_(buf[base + _TILE_ERROR_OFFSET: base + _TILE_ERROR_OFFSET + 4].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(buf.device))
        st.event = ev


def release_workspaces():
    """Drop every cached scratch buffer (e.g. the 3.8 GB of the gather-form backward after a training phase)."""
    with _workspaces_lock:
        _workspaces.clear()
    _last_tile_backward_ws.clear()


def tile_workspace(spec: LayerSpec, n: int, c: int, device) -> torch.Tensor:
    """A caller-owned workspace for forward_nhwc(..., workspace=...) on the tile path (empty tensor if it does not
    apply to this shape)."""
    d = spec.desc(n, c)
    return torch.zeros(int(_lib.load().et_epipolar_forward_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8,
                       device=device)


def tile_stats(spec: LayerSpec, n: int, c: int, workspace: torch.Tensor) -> torch.Tensor:
    """Per-tile statistics the tile forward leaves in its workspace: int32 (N * tiles_per_pair,),
    U | groups << 16 (size of the tile's source-row set, number of pixel groups it was split into)."""
    d = spec.desc(n, c)
    off = int(_lib.load().et_epipolar_forward_workspace_stats_offset(ctypes.byref(d)))
    tiles = n * ((spec.H * spec.W + 31) // 32)
    base = (-workspace.data_ptr()) % 256                     # the library aligns the base up to 256 bytes
    return workspace[base + off: base + off + 4 * tiles].view(torch.int32)


_BWD_EXPLICIT = _lib.ET_VARIANT_BWD_ATOMIC | _lib.ET_VARIANT_BWD_UNSORTED | _lib.ET_VARIANT_NO_TILE


def backward_nhwc(spec: LayerSpec, ref, src, cam, grad_out, use_workspace=True, form=None, attn=None):
    """d(feat_ref), d(feat_src) of forward_nhwc.  Three forms of the same gradient:
      "tile"    MFMA tile formulation, d(feat_src) accumulated with float atomics across tiles: fastest,
                reproducible to rounding only (C == 256, K <= 256);
      "gather"  per-(pixel,row) coefficients -> counting sort -> ordered per-row sums: no float atomics, bit-reproducible;
      "atomic"  bilinear-transpose scatter with float atomics (no workspace).
    form=None picks "tile" where it applies (unless the spec's variant names a backward form or NO_TILE), else
    "gather"; use_workspace=False means "atomic".  `attn`: the attention forward_nhwc returned for the same inputs
    (N,K,H,W) -- the tile form then does not recompute the soft-max (one GEMM of five less); the other forms ignore it."""
    n, h, w, c = ref.shape
    xs, ys, steps = spec.constants(ref.device)
    grad_out = grad_out.contiguous()
    g_ref = _empty(None, like=ref)
    g_src = _empty(None, like=src)
    d = spec.desc(n, c)
    lib = _lib.load()
    tile_bytes = int(lib.et_epipolar_backward_tiled_workspace_bytes(ctypes.byref(d)))
    if form is None:
        if not use_workspace:
            form = "atomic"
        else:
            form = "tile" if tile_bytes > 0 and not (d.variant & _BWD_EXPLICIT) else "gather"
    if form not in ("tile", "gather", "atomic"):
        raise ValueError("unknown backward form %r" % (form,))
    args = (ctypes.byref(d), _ptr(xs), _ptr(ys), _ptr(steps), _ptr(cam), _ptr(ref), _ptr(src), _ptr(grad_out),
            _ptr(g_ref), _ptr(g_src))
    with torch.cuda.device(ref.device):
        if form == "tile":
            if tile_bytes == 0:
                raise _lib.EpipolarAmdError("the tiled backward needs the 256-channel head (got C=%d, K=%d, %dx%d)" % (c, spec.K, h, w))
            ws = _workspace(ref.device, tile_bytes, "fwd")
            _last_tile_backward_ws[(ref.device.index, torch.cuda.current_stream(ref.device).cuda_stream)] = weakref.ref(ws)
            if attn is not None:
                assert attn.is_cuda and attn.dtype == torch.float32 and tuple(attn.shape) == (n, spec.K, h, w) and attn.is_contiguous()
            _lib.check(lib.et_epipolar_backward_tiled_attn(*args[:7], _ptr(attn), *args[7:], _ptr(ws), ctypes.c_size_t(tile_bytes),
                                                           _stream(ref)), "et_epipolar_backward_tiled_attn")
        else:
            ws, ws_bytes = None, 0
            if form == "gather":
                ws_bytes = int(lib.et_epipolar_backward_workspace_bytes(ctypes.byref(d)))
                ws = _workspace(ref.device, ws_bytes)
            _lib.check(lib.et_epipolar_backward(*args, _ptr(ws), ctypes.c_size_t(ws_bytes), _stream(ref)),
                       "et_epipolar_backward")
    return g_ref, g_src


_last_tile_backward_ws = {}     # (device index, stream) -> weak reference to the workspace the last tiled backward ran on


def backward_deferred_tiles(device, header=False, workspace=None):
    """Tiles the most recent tiled backward on `device` (current stream, WHICHEVER host thread launched it: autograd runs the
    backward on a thread of its own) handed from the merged two-array kernel to the one-array kernel because their row set
    exceeds 192 (288) columns -- word 0 of the workspace that call used (or of `workspace`).  Synchronises; a diagnostic
    (tests, profiling).  `header`: the first eight header words, (deferred, error word, four-group tiles met,
    eight-group tiles met early, over-capacity tiles met, ...).  Which tiles
    are deferred depends on the order the blocks of a launch reach them (et_tile_host.h): the count varies from run to run."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    buf = workspace
    if buf is None:
        ref = _last_tile_backward_ws.get((idx, torch.cuda.current_stream(dev).cuda_stream))
        buf = ref() if ref is not None else None
    if buf is None:
        return (0,) * 8 if header else 0
    base = (-buf.data_ptr()) % 256
    words = buf[base:base + 32].view(torch.int32).tolist()       # (eight header words)
    return tuple(words) if header else words[0]


def atomic_probe(device, rows: int = 1 << 18, blocks: int = 2048, iters: int = 256, reps: int = 5) -> dict:
    """Float-atomic request rate of this box (et_debug_atomic_probe: the access pattern of the tiled backward's d(feat_src)
    accumulation -- one buffer_atomic_fadd_f32 wave-instruction = two 128-byte runs in two pixel rows -- on pseudo-random rows of
    a (rows, 256) fp32 array, 1 GB by default, nothing else in the kernel).  A diagnostic for bench.py: synchronises."""
    dev = torch.device(device)
    dst = torch.zeros(rows, 256, device=dev)
    lib = _lib.load()
    ms = []
    with torch.cuda.device(dev):
        for r in range(reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(lib.et_debug_atomic_probe(_ptr(dst), rows, blocks, iters, _stream(dst)), "et_debug_atomic_probe")
            b.record()
            torch.cuda.synchronize(dev)
            if r:
                ms.append(a.elapsed_time(b))
    ms.sort()
    n_instr = blocks * 4 * iters
    return {"array_MB": rows * 1024 / 1e6, "wave_instructions": n_instr, "ms_min": ms[0], "ms_p50": ms[len(ms) // 2], "ms_max": ms[-1],
            "G_wave_atomics_per_s": n_instr / (ms[len(ms) // 2] * 1e-3) / 1e9, "GB_per_s": n_instr * 256 / (ms[len(ms) // 2] * 1e-3) / 1e9}


def residual_epilogue(feat, out, y=None, scale=None, shift=None, want_finalout=True, want_x=True):
    """All (N,H,W,C) contiguous.  finalout = out + y*scale + shift ; x = feat + finalout."""
    n, h, w, c = out.shape
    fin = _empty(None, like=out) if want_finalout else None
    x = _empty(None, like=out) if want_x else None
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().et_residual_epilogue(n * h * w, c, _ptr(feat), _ptr(out), _ptr(y), _ptr(scale),
                                                    _ptr(shift), _ptr(fin), _ptr(x), _stream(out)),
                   "et_residual_epilogue")
    return fin, x


def residual_gemm_pack(wf: torch.Tensor) -> torch.Tensor:
    """Lay the folded (256 out, 256 in) fp32 weight of the z branch out for `residual_gemm` (split fp16 MFMA fragments +
    the scale).  Returns the packed uint8 buffer; repack when the weights change."""
    _require_gpu(wf, "wf")
    assert wf.dtype == torch.float32 and tuple(wf.shape) == (256, 256), "residual_gemm is written for the 256-channel head"
    wf = wf.contiguous()
    lib = _lib.load()
    packed = torch.empty(int(lib.et_residual_gemm_packed_bytes()), dtype=torch.uint8, device=wf.device)
    with torch.cuda.device(wf.device):
        _lib.check(lib.et_residual_gemm_pack(_ptr(wf), _ptr(packed), _stream(wf)), "et_residual_gemm_pack")
    return packed


def residual_gemm(out: torch.Tensor, packed: torch.Tensor, bias: torch.Tensor, feat: torch.Tensor = None) -> torch.Tensor:
    """x = [feat +] bias + out @ Wf^T over the last dimension (256), everything (..., 256) fp32 contiguous: the
    eval-mode `bn(z(out)) [+ out] [+ feat]` (epipolar.py:250-253, resnet.py:388) as one HBM-bound kernel."""
    _require_gpu(out, "out")
    c = out.shape[-1]
    assert out.is_contiguous() and (feat is None or (feat.is_contiguous() and feat.shape == out.shape))
    assert bias.is_cuda and bias.numel() == c and bias.is_contiguous()
    x = _empty(None, like=out)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().et_residual_gemm(out.numel() // c, c, _ptr(out), _ptr(feat), _ptr(packed), _ptr(bias), _ptr(x),
                                                _stream(out)), "et_residual_gemm")
    return x


def z_batch_stats(out: torch.Tensor, packed_wz: torch.Tensor, z_bias: torch.Tensor):
    """First pass of the training-mode epilogue (et_z_batch_stats): y = out @ Wz^T + z_bias over the last dimension (256)
    and its per-channel batch mean / biased variance over all rows.  Returns (y, mean, var)."""
    _require_gpu(out, "out")
    c = out.shape[-1]
    if c != 256 or not out.is_contiguous():
        raise ValueError("out must be a contiguous (..., 256) tensor")
    if not (z_bias.is_cuda and z_bias.numel() == c and z_bias.is_contiguous() and z_bias.dtype == torch.float32):
        raise ValueError("z_bias must be a contiguous float32 vector of 256 values on the GPU")
    lib = _lib.load()
    if packed_wz.numel() < int(lib.et_residual_gemm_packed_bytes()) or packed_wz.dtype != torch.uint8 or not packed_wz.is_cuda:
        raise ValueError("packed_wz must be the buffer residual_gemm_pack returns")
    rows = out.numel() // c
    y = _empty(None, like=out)
    mean = _empty((c,), device=out.device)
    var = _empty((c,), device=out.device)
    with torch.cuda.device(out.device):
        ws_bytes = int(lib.et_z_batch_stats_workspace_bytes(rows))
        ws = _workspace(out.device, ws_bytes, "zstats")
        _lib.check(lib.et_z_batch_stats(rows, c, _ptr(out), _ptr(packed_wz), _ptr(z_bias), _ptr(y), _ptr(mean), _ptr(var), _ptr(ws),
                                        ctypes.c_size_t(ws_bytes), _stream(out)), "et_z_batch_stats")
    return y, mean, var


def z_backward(g: torch.Tensor, y: torch.Tensor, mean: torch.Tensor, invstd: torch.Tensor, gamma: torch.Tensor,
               packed_wzt: torch.Tensor, zresidual: bool):
    """Backward of the training-mode epilogue w.r.t. `out` and the batch norm's affine parameters (et_z_backward): g, y
    (..., 256) contiguous.  Returns (grad_out, grad_y, grad_gamma, grad_beta)."""
    _require_gpu(g, "g")
    _require_gpu(y, "y")
    c = g.shape[-1]
    if c != 256 or g.shape != y.shape or not (g.is_contiguous() and y.is_contiguous()):
        raise ValueError("g and y must be contiguous (..., 256) tensors of one shape")
    for t, nm in ((mean, "mean"), (invstd, "invstd"), (gamma, "gamma")):
        if not (t.is_cuda and t.numel() == c and t.is_contiguous() and t.dtype == torch.float32):
            raise ValueError("%s must be a contiguous float32 vector of 256 values on the GPU" % nm)
    lib = _lib.load()
    if packed_wzt.numel() < int(lib.et_residual_gemm_packed_bytes()) or packed_wzt.dtype != torch.uint8 or not packed_wzt.is_cuda:
        raise ValueError("packed_wzt must be the buffer residual_gemm_pack returns")
    rows = g.numel() // c
    gout, gy = _empty(None, like=g), _empty(None, like=g)
    ggamma, gbeta = _empty((c,), device=g.device), _empty((c,), device=g.device)
    with torch.cuda.device(g.device):
        ws_bytes = int(lib.et_z_backward_workspace_bytes(rows))
        ws = _workspace(g.device, ws_bytes, "zbwd")
        _lib.check(lib.et_z_backward(rows, c, _ptr(g), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(packed_wzt),
                                     1 if zresidual else 0, _ptr(gout), _ptr(gy), _ptr(ggamma), _ptr(gbeta), _ptr(ws),
                                     ctypes.c_size_t(ws_bytes), _stream(g)), "et_z_backward")
    return gout, gy, ggamma, gbeta


def z_wgrad(grad_y: torch.Tensor, out: torch.Tensor):
    """d Wz (256, 256) = grad_y^T @ out and d bz (256) = grad_y.sum(rows) over (..., 256) contiguous tensors (et_z_wgrad:
    three-term bf16 MFMAs with fp32 accumulation, no atomics, bit-reproducible)."""
    _require_gpu(grad_y, "grad_y")
    _require_gpu(out, "out")
    c = out.shape[-1]
    if c != 256 or grad_y.shape != out.shape or not (grad_y.is_contiguous() and out.is_contiguous()):
        raise ValueError("grad_y and out must be contiguous (..., 256) tensors of one shape")
    rows = out.numel() // c
    gw, gb = _empty((c, c), device=out.device), _empty((c,), device=out.device)
    lib = _lib.load()
    with torch.cuda.device(out.device):
        ws_bytes = int(lib.et_z_wgrad_workspace_bytes(rows))     # (sized by the CU count of the CURRENT device: inside the guard)
        ws = _workspace(out.device, ws_bytes, "zwgrad")
        _lib.check(lib.et_z_wgrad(rows, c, _ptr(grad_y), _ptr(out), _ptr(gw), _ptr(gb), _ptr(ws), ctypes.c_size_t(ws_bytes),
                                  _stream(out)), "et_z_wgrad")
    return gw, gb


def heatmap_peaks(heatmaps: torch.Tensor, radius: float, downsample: float, threshold: float = 1e-6,
                  legacy_floor_division: bool = False):
    """find_tensor_peak_batch for a whole batch in ONE kernel: heatmaps (N,J,H,W) -> locations (N,J,2) in image
    coordinates and scores (N,J).  No gradient (the reference's locations are only used for evaluation)."""
    _require_gpu(heatmaps, "heatmaps")
    n, j, h, w = heatmaps.shape
    hm = heatmaps.detach().contiguous()
    locs = _empty((n, j, 2), device=hm.device)
    scores = _empty((n, j), device=hm.device)
    with torch.cuda.device(hm.device):
        _lib.check(_lib.load().et_heatmap_peaks(n * j, h, w, _ptr(hm), float(radius), float(downsample), float(threshold),
                                                int(bool(legacy_floor_division)), _ptr(locs), _ptr(scores), _stream(hm)),
                   "et_heatmap_peaks")
    return locs, scores


class EpipolarAttend(torch.autograd.Function):
    """out = sum_k softmax_k(scale * mask(f_ref . S_k)) S_k with S_k the K bilinear
    samples of f_src on the pixel's epipolar segment (epipolar.py:188-247).
    Inputs/outputs are logical NCHW; attn and corr_pos are returned without
    gradient (the reference never back-propagates through them in the
    configurations of BASELINE.json).  The backward takes backward_nhwc's default form: the MFMA tile kernel for
    the 256-channel head (float atomics across tiles, reproducible to rounding), the bit-reproducible gather form
    otherwise or when the spec's variant carries ET_VARIANT_NO_TILE."""

    @staticmethod
    def forward(ctx, feat_ref, feat_src, cam, spec: LayerSpec):
        ref = to_nhwc(feat_ref)
        src = to_nhwc(feat_src)
        out, attn, corr = forward_nhwc(spec, ref, src, cam)
        ctx.spec = spec
        ctx.save_for_backward(ref, src, cam, attn)       # (the soft-max output, as autograd keeps it in the reference)
        ctx.mark_non_differentiable(attn, corr)
        return out.permute(0, 3, 1, 2), attn, corr

    @staticmethod
    def backward(ctx, grad_out, _ga, _gc):
        ref, src, cam, attn = ctx.saved_tensors
        g = to_nhwc(grad_out)
        g_ref, g_src = backward_nhwc(ctx.spec, ref, src, cam, g, attn=attn)
        need_ref, need_src = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        return (g_ref.permute(0, 3, 1, 2) if need_ref else None,
                g_src.permute(0, 3, 1, 2) if need_src else None, None, None)
