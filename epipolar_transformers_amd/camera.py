"""Per-pair camera algebra of grid2sample_locs, kept on the host in float32.

The reference computes, per (reference, source) pair, a 4x3 pseudo-inverse by
SVD, a 3x3 inverse and the epipole in float32 (modeling/layers/epipolar.py:336,
344-348; vision/multiview.py:16-21).  With cond(P) ~ 1e6 those roundings move
sample locations by up to 4e-3, so parity with the
reference's CPU path requires *these exact* float32 values; every per-pixel step
after them runs on the GPU.  This module produces them batched -- bit-identical
to the reference's per-matrix loop ON THE HOST IT RUNS ON (the last bits depend on
the host's BLAS kernel selection, see `_calibrate`) at ~0.4 ms for 128 pairs
instead of 2-4 ms -- and packs them in the `cam` layout of include/epipolar_amd.h.
"""
from __future__ import annotations

import torch

ET_CAM_STRIDE = 27


def _pinverse_loop(P: torch.Tensor) -> torch.Tensor:
    """The reference's own formulation (epipolar.py:336)."""
    return torch.stack([p.pinverse() for p in P])


def _svd_factors(P):
    # torch.pinverse(A) is (Vh^T * 1/S) @ U^T from the LAPACK SVD (rcond 1e-15); the batched SVD
    # is bit-identical to the per-matrix one, only the tiny 4x3 @ 3x3 product is not necessarily
    U, S, Vh = torch.linalg.svd(P, full_matrices=False)
    cutoff = 1e-15 * S.amax(-1, keepdim=True)
    s_inv = torch.where(S > cutoff, 1.0 / S, torch.zeros_like(S))
    return Vh.transpose(-1, -2) * s_inv.unsqueeze(-2), U.transpose(-1, -2)      # (N,4,3), (N,3,3)


def _pinv_fma_chain(P):
    """mm as an ascending-k FMA chain (what MKL runs on Intel hosts), spelled through float64."""
    A, B = _svd_factors(P)
    A, B = A.double(), B.double()
    acc = (A[..., :, 0:1] * B[..., 0:1, :]).float()
    acc = (A[..., :, 1:2] * B[..., 1:2, :] + acc.double()).float()
    acc = (A[..., :, 2:3] * B[..., 2:3, :] + acc.double()).float()
    return acc


def _pinv_mul_add(P):
    """mm as separately rounded products and sums (what MKL runs on the EPYC hosts of the GPU boxes)."""
    A, B = _svd_factors(P)
    return ((A[..., :, 0:1] * B[..., 0:1, :]) + (A[..., :, 1:2] * B[..., 1:2, :])) + (A[..., :, 2:3] * B[..., 2:3, :])


_BATCHED_CANDIDATES = (("linalg.pinv", torch.linalg.pinv), ("mul_add", _pinv_mul_add), ("fma_chain", _pinv_fma_chain))
_calibrated = None          # name of the batched formulation that reproduces the loop on this host, or "loop"


def _calibrate():
    """The last bits of the reference's pinverse depend on which kernel the host's BLAS picks for a
    4x3 @ 3x3 product, and the layer is discontinuous in them.  Find, once per
    process, a batched formulation that is bit-identical to the reference loop on THIS host."""
    global _calibrated
    g = torch.Generator().manual_seed(1234)
    probe = torch.randn(24, 3, 4, generator=g) * torch.tensor([300.0, 300.0, 300.0, 1.0e6])
    probe[:, 2] = torch.randn(24, 4, generator=g) * torch.tensor([1.0, 1.0, 1.0, 5.0e3])
    want = _pinverse_loop(probe)
    _calibrated = "loop"
    for name, fn in _BATCHED_CANDIDATES:
        try:
            if torch.equal(fn(probe), want):
                _calibrated = name
                break
        except Exception:      # pragma: no cover - a candidate the installed torch cannot run
            continue
    return _calibrated


def batched_pinverse(P: torch.Tensor) -> torch.Tensor:
    """`torch.stack([p.pinverse() for p in P])` (epipolar.py:336) at batched cost (~0.4 ms instead of
    ~4 ms for 128 pairs), bit-identical to that loop on the host it runs on: calibrated once, and
    spot-checked against the loop on two matrices of every call (falls back to the loop on mismatch)."""
    n = P.shape[0]
    method = _calibrated or _calibrate()
    if method == "loop" or n <= 4:
        return _pinverse_loop(P)
    out = dict(_BATCHED_CANDIDATES)[method](P)
    for i in (0, n - 1):
        if not torch.equal(out[i], P[i].pinverse()):
            return _pinverse_loop(P)
    return out


def pair_algebra(P_ref: torch.Tensor, P_src: torch.Tensor) -> torch.Tensor:
    """(N,3,4),(N,3,4) -> (N,27) float32 CPU tensor [P1inv | P2 | e2].

    Inputs may live on any device; they are brought to the CPU (a GPU-resident
    P costs one small synchronising copy -- pass CPU tensors, as the data
    loader produces them, to keep the launch path asynchronous)."""
    P1 = P_ref.detach().to("cpu", torch.float32)
    P2 = P_src.detach().to("cpu", torch.float32)
    if P1.dim() != 3 or P1.shape[1:] != (3, 4) or P2.shape != P1.shape:
        raise ValueError("expected two (N,3,4) projection-matrix batches, got %s and %s"
                         % (tuple(P_ref.shape), tuple(P_src.shape)))
    n = P1.shape[0]
    p1inv = batched_pinverse(P1)
    inv_a = torch.inverse(P1[..., :3])                               # multiview.py:17
    centre = -torch.matmul(inv_a, P1[..., 3, None])                  # multiview.py:18
    hom = torch.ones([n, 4, 1], dtype=torch.float32)                 # multiview.py:19-20
    hom[..., :3, :] = centre
    e2 = torch.matmul(P2, hom).view(n, 3, 1)                         # epipolar.py:346
    e2 = e2 / e2[:, [2], :]                                          # epipolar.py:348
    return torch.cat([p1inv.reshape(n, 12), P2.reshape(n, 12), e2.reshape(n, 3)], 1).contiguous()


class PairAlgebraCache:
    """Small value-keyed cache (camera rigs repeat across frames and steps).
    Keyed on the BYTES of the matrices, never on tensor identity: a data loader frees and re-allocates its batches,
    and a recycled address with different matrices must not hit (SURVEY.md 8b "Ownership").

    A caller that hands GPU-resident matrices -- the reference's Modelbuilder does, after
    `batchdata.to(device)` (model.py:183-195) -- pays one small blocking device-to-host copy per call to form the
    key.  Pass `host=(P_ref_cpu, P_src_cpu)`, the copies the data loader produced anyway, to keep the launch path
    asynchronous (main.py's launcher does).  Parity note: the algebra is float32 LAPACK on the HOST (pinverse /
    inverse), the reference's CPU path; a reference running pinverse on a GPU can differ in the last bits, and the
    layer is discontinuous in them."""

    def __init__(self, max_entries: int = 8):
        self.max_entries = max_entries
        self._store = {}

    def get(self, P_ref: torch.Tensor, P_src: torch.Tensor, device, host=None) -> torch.Tensor:
        src_a, src_b = host if host is not None else (P_ref, P_src)
        a = src_a.detach().to("cpu", torch.float32).contiguous()
        b = src_b.detach().to("cpu", torch.float32).contiguous()
        key = (a.numpy().tobytes(), b.numpy().tobytes(), str(device))
        hit = self._store.get(key)
        if hit is None:
            cam = pair_algebra(a, b)
            if torch.device(device).type == "cuda":
                cam = cam.pin_memory().to(device, non_blocking=True)
            if len(self._store) >= self.max_entries:
                self._store.pop(next(iter(self._store)))
            self._store[key] = hit = cam
        return hit
