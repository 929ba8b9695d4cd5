"""Generate tests/golden/triangulation.npz from the REAL reference lifting code (row N3 of SURVEY.md section 8f).

Runs only in the build container (needs /root/reference, imported read-only through oracle/ref_harness.py, and the
pymvg stand-in oracle/pymvg_stub.py -- pymvg itself is not installable offline; the stand-in's header says which three
helpers it restates).  What is executed is the reference's own code:

    vision/triangulation.py:400-441   triangulate_pymvg: the confidence rule (cfg.KEYPOINT.CONF_THRES, lowered in
                                      steps of 0.05 until two views remain) and the per-joint loop
    vision/triangulation.py:350-379   build_multi_camera_system (CameraModel._from_parts per view)
    vision/multi_camera_system.py:199-225   find3d: undistort, rows x * P[2] - P[0], y * P[2] - P[1], SVD, de-homogenise

    python tests/golden/make_triangulation_golden.py

Stored per scene: K (V,3,3), RT (V,3,4) float32 as Modelbuilder hands them over (model.py:186-191), detections (V,J,2),
scores (V,J) float32, the threshold, and the reference's (J,3) float64 points.
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pymvg_stub, ref_harness as rh  # noqa: E402


def reference_triangulate_pymvg():
    """The reference function, importable under numpy 2 / without pymvg."""
    rh.install()
    pymvg_stub.install()
    # names the vendored pymvg files use at import time (numpy < 1.20); this generator process only
    for name, val in (("float", float), ("alltrue", np.all)):
        if not hasattr(np, name):
            setattr(np, name, val)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import vision.camera_model as cm
        import vision.triangulation as tri
    cm.np = pymvg_stub.numpy_legacy_proxy()          # np.array(x, copy=False) == "copy only if needed" (numpy < 2)
    return tri


def look_at_rig(V, seed):
    """V cameras on a ring (SURVEY.md 8d synthetic rig), float64 K, RT with x_cam = R x + t."""
    rng = np.random.default_rng(seed)
    Ks, RTs = [], []
    for i in range(V):
        ang = (2 * i + 0.5) * np.pi / V + rng.normal(0, 0.05)
        c = np.array([5000 * np.cos(ang), 5000 * np.sin(ang), 1500.0 + rng.normal(0, 50)])
        fwd = np.array([0, 0, 900.0]) - c
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, [0, 0, 1.0])
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        roll = rng.uniform(0.05, 0.2) * (1 if i % 2 else -1)      # generic rotations (no exactly-zero entries: the
        right, down = (np.cos(roll) * right + np.sin(roll) * down,  # reference asserts a float32 R survives its
                       np.cos(roll) * down - np.sin(roll) * right)  # quaternion round trip to 1e-8 + 1e-5 |r|)
        R = np.stack([right, down, fwd])
        f = 1145.0 + rng.normal(0, 5)
        Ks.append(np.array([[f, 0, 512.0 + rng.normal(0, 3)], [0, f, 515.0 + rng.normal(0, 3)], [0, 0, 1]]))
        RTs.append(np.concatenate([R, (-R @ c)[:, None]], 1))
    return np.stack(Ks), np.stack(RTs)


def scene(V, J, seed, noise):
    rng = np.random.default_rng(seed)
    K, RT = look_at_rig(V, seed)
    X = np.array([0, 0, 900.0]) + rng.normal(0, 300, (J, 3))
    xh = np.einsum("vij,vjk,nk->vni", K, RT, np.concatenate([X, np.ones((J, 1))], 1))
    pts = xh[..., :2] / xh[..., 2:] + rng.normal(0, noise, (V, J, 2))
    return K.astype(np.float32), RT.astype(np.float32), pts.astype(np.float32), X


def main():
    tri = reference_triangulate_pymvg()
    cases = {}
    J = 17
    specs = [
        # name, V, threshold, how the scores are made
        ("all_views_default_thres", 4, 0.05, "high"),
        ("thres085_drops_low_views", 4, 0.85, "mixed"),
        ("thres085_lowered_until_two_views", 4, 0.85, "one_high"),
        ("thres_boundary_float32", 4, 0.85, "boundary"),
        ("eight_views", 8, 0.85, "mixed"),
        ("all_scores_tiny", 4, 0.05, "tiny"),
    ]
    for ci, (name, V, thres, kind) in enumerate(specs):
        K, RT, pts, X = scene(V, J, 100 + ci, noise=1.5)
        rng = np.random.default_rng(7 + ci)
        if kind == "high":
            conf = rng.uniform(0.3, 1.0, (V, J))
        elif kind == "mixed":
            conf = rng.uniform(0.6, 1.0, (V, J))
            pts[conf < thres] += rng.normal(0, 40, pts[conf < thres].shape).astype(np.float32)   # dropped views are wrong
        elif kind == "one_high":
            conf = rng.uniform(0.2, 0.8, (V, J))
            conf[0] = 0.95                                   # one view above 0.85: the rule lowers the threshold
            conf[1, :5] = 0.849                              # ... by one step for these joints
        elif kind == "boundary":
            conf = rng.uniform(0.86, 1.0, (V, J))
            conf[2] = np.float32(0.85)                       # == float32(threshold): NOT above it (compared in float32)
            conf[3, ::2] = np.nextafter(np.float32(0.85), np.float32(1))
        else:
            conf = rng.uniform(0.0, 0.04, (V, J))            # nothing above 0.05: lowered to 0 and below
            conf[:, 3] = 0.0                                 # all-zero scores: threshold ends negative
        conf = conf.astype(np.float32)
        cfg = rh.load_cfg(None, ["KEYPOINT.CONF_THRES", str(thres)])
        assert abs(float(cfg.KEYPOINT.CONF_THRES) - thres) < 1e-12
        import io
        import contextlib

        with contextlib.redirect_stdout(io.StringIO()):     # the reference prints every threshold step
            got = tri.triangulate_pymvg(torch.from_numpy(pts), K, RT, torch.from_numpy(conf))
        got = np.asarray(got, dtype=np.float64)
        err = np.linalg.norm(got - X, axis=1)
        print("%-36s V=%d thres %.2f: reference 3-D error vs planted joints mean %.2f mm max %.2f mm" % (name, V, thres, err.mean(), err.max()))
        for key, val in (("K", K), ("RT", RT), ("pts", pts), ("conf", conf), ("thres", np.float64(thres)), ("X_ref", got), ("X_true", X)):
            cases["%s.%s" % (name, key)] = val
    out = os.path.join(ROOT, "tests", "golden", "triangulation.npz")
    np.savez_compressed(out, **cases)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
