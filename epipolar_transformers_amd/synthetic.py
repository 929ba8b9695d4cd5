"""Synthetic H36M-like inputs: a ring of look-at cameras
and post-ReLU feature maps.  No dataset or checkpoint exists offline, so tests,
fixtures and bench.py all draw their inputs from here with fixed seeds.

Camera model mirrors what the reference data loader hands to the model
(data/datasets/joints_dataset.py:239-248,334-336): KRT = A_crop . K . [R | -R C]
computed in float64 and cast to float32 (modeling/model.py:185,195).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def look_at_camera(centre, target, f=1145.0, c=(512.0, 515.0)):
    """Return K (3x3) and [R|t] (3x4), float64, for a camera at `centre` looking
    at `target` with world +z up."""
    centre = np.asarray(centre, np.float64)
    target = np.asarray(target, np.float64)
    fwd = target - centre
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd])          # rows: camera x, y, z axes in world
    t = -R @ centre
    K = np.array([[f, 0.0, c[0]], [0.0, f, c[1]], [0.0, 0.0, 1.0]])
    return K, np.concatenate([R, t[:, None]], 1)


def ring_cameras(num_views=4, image_size=256, radius=5000.0, height=1500.0,
                 target=(0.0, 0.0, 900.0), sensor=1000.0, jitter=None, rng=None):
    """(V,3,4) float64 projection matrices of V cameras on a ring.

    jitter: optional (scale_sigma, shift_sigma_px) to mimic per-frame crops
    (joints_dataset.py:334-336); rng: np.random.Generator for it."""
    mats = []
    for i in range(num_views):
        ang = (2 * i + 0.5) * math.pi / num_views
        centre = (radius * math.cos(ang), radius * math.sin(ang), height)
        K, RT = look_at_camera(centre, target)
        s = image_size / sensor
        A = np.array([[s, 0.0, 0.0], [0.0, s, 0.0], [0.0, 0.0, 1.0]])
        if jitter is not None:
            ds = 1.0 + jitter[0] * rng.standard_normal()
            sh = jitter[1] * rng.standard_normal(2)
            A = np.array([[s * ds, 0.0, sh[0]], [0.0, s * ds, sh[1]], [0.0, 0.0, 1.0]])
        mats.append(A @ K @ RT)
    return np.stack(mats)


def make_pairs(num_frames, num_views=4, image_size=256, seed=0, jitter=None):
    """Test-time pairing (data/datasets/multiview_h36m.py:231-238): every view
    of every frame is the reference once, its ring neighbour is the source.
    Returns P_ref, P_src as float32 tensors (N,3,4), N = frames*views, ordered
    frame-major."""
    rng = np.random.default_rng(seed)
    p_ref, p_src = [], []
    for _ in range(num_frames):
        cams = ring_cameras(num_views, image_size, jitter=jitter, rng=rng)
        for v in range(num_views):
            p_ref.append(cams[v])
            p_src.append(cams[(v + 1) % num_views])
    to32 = lambda a: torch.from_numpy(np.stack(a)).float()
    return to32(p_ref), to32(p_src)


def make_features(N, C, H, W, seed=0, relu=True, smooth=False):
    """Post-ReLU random feature maps (NCHW float32).  relu=True gives the
    statistics of resnet.py:359 outputs and exercises the exact-zero mask."""
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(N, C, H, W, generator=g)
    f2 = torch.randn(N, C, H, W, generator=g)
    if smooth:
        k = torch.ones(1, 1, 5, 5) / 25.0
        f1 = torch.nn.functional.conv2d(f1.view(N * C, 1, H, W), k, padding=2).view(N, C, H, W)
        f2 = torch.nn.functional.conv2d(f2.view(N * C, 1, H, W), k, padding=2).view(N, C, H, W)
    if relu:
        f1, f2 = f1.relu(), f2.relu()
    return f1.contiguous(), f2.contiguous()
