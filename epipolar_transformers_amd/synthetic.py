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


# ---- camera rigs beyond the ring (the geometry cases of tests/golden/make_golden.py and tests/test_gpu_rigs.py) ------------
# The ring above keeps the epipole far outside the map: every epipolar line crosses the whole image and the fan of lines
# is narrow.  The rigs below cover what the reference's geometry code (epipolar.py:340-407) does elsewhere: a fan through
# 360 degrees (epipole inside the map), an epipole on the rectangle's edge, parallel lines (epipole at infinity, the
# sign(0) = 0 clamp of epipolar.py:369-373), no usable line at all (identical cameras: every pixel takes the "< 2 valid"
# placeholder of epipolar.py:395-403), and an H36M-like room rig paired by the reference's nearest-neighbour rule.
RIGS = ("ring", "epipole_inside", "epipole_border", "rectified_x", "near_rectified_x", "near_rectified_y", "identical", "h36m_room")

# camera centres (mm) of a Human3.6M-like capture room: four cameras near the corners of a ~4 x 10 m floor at ~1.5 m,
# deliberately not on a circle and not evenly spaced
_H36M_ROOM_CENTRES = ((1841.0, 4955.0, 1563.0), (1761.0, -5078.0, 1606.0), (-1846.0, 5215.0, 1491.0), (-1794.0, -3722.0, 1574.0))


def _projection(K, RT, image_size, sensor=1000.0, jitter=None, rng=None):
    s = image_size / sensor
    A = np.array([[s, 0.0, 0.0], [0.0, s, 0.0], [0.0, 0.0, 1.0]])
    if jitter is not None:
        ds = 1.0 + jitter[0] * rng.standard_normal()
        sh = jitter[1] * rng.standard_normal(2)
        A = np.array([[s * ds, 0.0, sh[0]], [0.0, s * ds, sh[1]], [0.0, 0.0, 1.0]])
    return A @ K @ RT


def nearest_neighbour_pairs(mats):
    """The reference's test-time pairing (vision/multiview.py:59-83 + data/datasets/multiview_h36m.py:231-238): every
    camera is the reference once, the camera whose centre is nearest is its source.  mats: (V,3,4) float64."""
    centres = [-np.linalg.inv(m[:, :3]) @ m[:, 3] for m in mats]
    src = []
    for i, c in enumerate(centres):
        d = [np.linalg.norm(c - o) if j != i else np.inf for j, o in enumerate(centres)]
        src.append(int(np.argmin(d)))
    return src


# four cameras on an uneven arc: nearest neighbours that are NOT mutual -- cameras 0 and 2 both take camera 1 as their source,
# camera 3 is nobody's -- what the reference's pairing rule gives for a rig like this and a ring pairing cannot express
_UNEVEN_ARC_ANGLES = (0.10, 0.55, 1.25, 2.60)


def rig_cameras(rig, num_frames=1, image_size=256, seed=0, jitter=None):
    """The cameras themselves, (num_frames, V, 3, 4) float64, of a rig whose views can be sharded one camera per rank:
    "ring" (as make_pairs draws them), "h36m_room" (as rig_pairs draws them: the same matrices for the same seed) and
    "uneven_arc" (four cameras whose nearest neighbours are not mutual).  Pair them with source_table(rig)."""
    rng = np.random.default_rng(seed)
    target = np.array([0.0, 0.0, 900.0])
    out = []
    for _ in range(num_frames):
        if rig == "ring":
            cams = ring_cameras(4, image_size, jitter=jitter, rng=rng)
        elif rig == "h36m_room":
            cams = [_projection(*look_at_camera(c, target + rng.normal(0, 50.0, 3)), image_size, jitter=jitter, rng=rng)
                    for c in _H36M_ROOM_CENTRES]
        elif rig == "uneven_arc":
            cams = [_projection(*look_at_camera(np.array([5000.0 * math.cos(a), 5000.0 * math.sin(a), 1500.0]), target), image_size,
                                jitter=jitter, rng=rng) for a in _UNEVEN_ARC_ANGLES]
        else:
            raise ValueError("rig_cameras: no multi-camera rig named %r" % (rig,))
        out.append(np.stack(cams))
    return np.stack(out)


def source_table(rig, num_views=4):
    """source_of[v] = the camera whose map reference camera v samples: the ring neighbour for "ring" (make_pairs), the
    reference's nearest-neighbour rule (nearest_neighbour_pairs: vision/multiview.py:59-83 + multiview_h36m.py:231-238) for the
    rigs with fixed camera centres.  Not a permutation in general: two views may share a source, a view may be nobody's."""
    if rig == "ring":
        return [(v + 1) % num_views for v in range(num_views)]
    return nearest_neighbour_pairs(list(rig_cameras(rig, 1, 256, seed=0)[0]))


def rig_pairs(rig, num_frames=1, image_size=256, seed=0, jitter=None):
    """P_ref, P_src (N,3,4) float32 of a named rig (RIGS); N = 4 * num_frames except for the two-camera rigs, which
    yield both orderings of the pair per frame (N = 2 * num_frames)."""
    if rig == "ring":
        return make_pairs(num_frames, 4, image_size, seed, jitter)
    if rig == "uneven_arc":         # (not in RIGS: a pairing-table case of the view-sharded exchange, not a geometry case)
        cams = rig_cameras(rig, num_frames, image_size, seed, jitter)
        table = source_table(rig)
        ref = torch.from_numpy(cams.reshape(-1, 3, 4)).float()
        src = torch.from_numpy(np.stack([cams[f, table[v]] for f in range(num_frames) for v in range(4)])).float()
        return ref, src
    rng = np.random.default_rng(seed)
    target = np.array([0.0, 0.0, 900.0])
    p_ref, p_src = [], []
    for _ in range(num_frames):
        if rig == "h36m_room":
            cams = [_projection(*look_at_camera(c, target + rng.normal(0, 50.0, 3)), image_size, jitter=jitter, rng=rng)
                    for c in _H36M_ROOM_CENTRES]
            src = nearest_neighbour_pairs(cams)
            for v in range(4):
                p_ref.append(cams[v])
                p_src.append(cams[src[v]])
            continue
        # two cameras A, B sharing one rotation (look-at from A)
        a = np.array([5000.0 * math.cos(0.3), 5000.0 * math.sin(0.3), 1500.0])
        K, RT = look_at_camera(a, target)
        R = RT[:, :3]
        right, down, fwd = R[0], R[1], R[2]
        if rig == "epipole_inside":
            # B stands in front of A, a little off its axis: A's centre projects INSIDE B's image and vice versa is
            # behind B (negative depth, still a finite epipole inside the image after the division)
            b = a + 1200.0 * fwd + 90.0 * right - 60.0 * down
        elif rig == "epipole_border":
            # the offset that puts the epipole on the rectangle's left edge: x_e = s (cx + f dx / dz) = xmin = 1.5
            s = image_size / 1000.0
            dz = 1500.0
            dx = (1.5 / s - K[0, 2]) * dz / K[0, 0]
            b = a - dz * fwd - dx * right + 40.0 * down       # A is dz in FRONT of B, at camera-x = dx, camera-y = -40 mm
        elif rig == "rectified_x":
            # pure sideways baseline: the epipole's third coordinate is exactly 0 in float32, the division of
            # epipolar.py:348 yields (+-inf, inf|nan) and every comparison of :388-393 fails -- all pixels invalid
            b = a + 400.0 * right
        elif rig == "near_rectified_x":
            b = a + 400.0 * right + 0.4 * fwd                  # epipole ~1e6 px away: horizontal lines, |l2.x| tiny
        elif rig == "near_rectified_y":
            b = a + 400.0 * down + 0.4 * fwd                   # the same with vertical lines (y-major tiles)
        elif rig == "identical":
            b = a.copy()
        else:
            raise ValueError("unknown rig %r (have %s)" % (rig, ", ".join(RIGS)))
        RTb = np.concatenate([R, (-R @ b)[:, None]], 1)
        pa = _projection(K, RT, image_size, jitter=jitter, rng=rng)
        if rig in ("rectified_x", "near_rectified_x", "near_rectified_y", "identical"):
            pa, pb = _projection(K, RT, image_size), _projection(K, RTb, image_size)   # (a rectified pair shares its crop)
        else:
            pb = _projection(K, RTb, image_size, jitter=jitter, rng=rng)
        p_ref += [pa, pb]
        p_src += [pb, pa]
    to32 = lambda x: torch.from_numpy(np.stack(x)).float()
    return to32(p_ref), to32(p_src)
