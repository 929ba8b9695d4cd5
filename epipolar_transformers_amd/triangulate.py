"""Batched linear triangulation (SVD-DLT) and MPJPE -- SURVEY.md section 8f row N3.

The reference lifts 2-D joints to 3-D on the CPU, joint by joint, through pymvg
(`vision/triangulation.py:400-441` -> `vision/multi_camera_system.py:199-225`,
Hartley & Zisserman 12.2): rows `x * P[2] - P[0]`, `y * P[2] - P[1]` per selected
view, last right-singular vector, de-homogenise.  pymvg is not installable
offline; this restates the same linear method, batched over frames and joints in
float64 torch (runs on the GPU next to the model, no D2H round trip), with the
reference's confidence rule (`cfg.KEYPOINT.CONF_THRES`, lowered in steps of 0.05
until at least two views remain).  `mpjpe` is `EPEmean` of
`modeling/metrics/metrics3d.py:5-46` without its dataset book-keeping.
"""
from __future__ import annotations

import torch


def triangulate_dlt(points_2d: torch.Tensor, proj: torch.Tensor, conf: torch.Tensor = None,
                    conf_thres: float = 0.05) -> torch.Tensor:
    """points_2d: (F, V, J, 2) image coordinates; proj: (F, V, 3, 4); conf: (F, V, J) or None.
    Returns (F, J, 3) world coordinates."""
    F_, V, J, _ = points_2d.shape
    pts = points_2d.to(torch.float64)
    P = proj.to(torch.float64)
    x, y = pts[..., 0], pts[..., 1]                                   # (F,V,J)
    row2 = P[:, :, None, 2, :]                                        # (F,V,1,4)
    a0 = x[..., None] * row2 - P[:, :, None, 0, :]                    # (F,V,J,4)
    a1 = y[..., None] * row2 - P[:, :, None, 1, :]
    if conf is not None:
        # triangulation.py:425-435: `conf > confthresh` compares the float32 scores with a Python float, i.e. IN FLOAT32
        # (a score equal to float32(threshold) is not above it); the threshold itself steps down in float64
        c = conf.to(torch.float32)
        thres = torch.full((F_, 1, J), conf_thres, dtype=torch.float64, device=pts.device)
        for _ in range(64):
            few = ((c > thres.to(torch.float32)).sum(1, keepdim=True) <= 1) & (thres >= -1)
            if not bool(few.any()):
                break
            thres = torch.where(few, thres - 0.05, thres)
        keep = (c > thres.to(torch.float32)).to(torch.float64)[..., None]   # (F,V,J,1)
        a0, a1 = a0 * keep, a1 * keep                                 # a dropped view contributes zero rows
    A = torch.cat([a0, a1], 1).permute(0, 2, 1, 3)                    # (F,J,2V,4)
    _, _, vt = torch.linalg.svd(A, full_matrices=False)
    X = vt[..., -1, :]
    return X[..., :3] / X[..., 3:4]


def mpjpe(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Mean per-joint position error (same units as the inputs, mm for H36M)."""
    return (pred.to(torch.float64) - target.to(torch.float64)).norm(dim=-1).mean()
