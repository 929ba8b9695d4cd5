"""Batched SVD-DLT triangulation (row N3) on CPU: exact recovery of planted points, the
reference's confidence rule, and the frozen MPJPE scene's reference detections."""
import os

import numpy as np
import torch

from conftest import GOLDEN_DIR
from epipolar_transformers_amd import synthetic as syn
from epipolar_transformers_amd.triangulate import mpjpe, triangulate_dlt


def _project(P, X):
    Xh = torch.cat([X, torch.ones(X.shape[0], 1, dtype=X.dtype)], 1)
    p = P @ Xh.T
    return (p[:, :2] / p[:, 2:3]).permute(0, 2, 1)          # (V,J,2)


def test_dlt_recovers_planted_points():
    P = torch.from_numpy(syn.ring_cameras(4, 256))
    X = torch.tensor([0.0, 0.0, 900.0], dtype=torch.float64) + torch.randn(17, 3, dtype=torch.float64) * 300
    uv = _project(P, X)
    got = triangulate_dlt(uv[None], P[None])
    assert mpjpe(got[0], X).item() < 1e-6                   # mm


def test_dlt_confidence_rule_drops_bad_views():
    P = torch.from_numpy(syn.ring_cameras(4, 256))
    X = torch.tensor([[100.0, -50.0, 1000.0]], dtype=torch.float64)
    uv = _project(P, X)
    uv[2] += 40.0                                            # one corrupted view with low confidence
    conf = torch.tensor([0.9, 0.8, 0.01, 0.7]).view(1, 4, 1)
    assert mpjpe(triangulate_dlt(uv[None], P[None], conf)[0], X).item() < 1e-6
    assert mpjpe(triangulate_dlt(uv[None], P[None])[0], X).item() > 10.0
    # fewer than two confident views: the threshold is lowered in steps of 0.05 (triangulation.py:427-435)
    conf = torch.tensor([0.04, 0.03, 0.01, 0.02]).view(1, 4, 1)
    got = triangulate_dlt(uv[None], P[None], conf)
    assert torch.isfinite(got).all()


def test_mpjpe_scene_reference_detections_triangulate_near_ground_truth():
    d = np.load(os.path.join(GOLDEN_DIR, "mpjpe_scene.npz"))
    P = torch.from_numpy(d["P"]).double()
    locs = torch.from_numpy(d["ref_locs"]).double()          # (V,J,2) from the reference pipeline
    X = triangulate_dlt(locs[None], P[None], torch.from_numpy(d["ref_scores"])[None])
    err = mpjpe(X[0], torch.from_numpy(d["joints"]))
    assert err.item() < 150.0                                # mm: 16x16 heat-maps, ~70 mm per image pixel


def _triangulation_cases():
    d = np.load(os.path.join(GOLDEN_DIR, "triangulation.npz"))
    return d, sorted({k.split(".")[0] for k in d.files})


def test_dlt_matches_the_reference_find3d_and_confidence_rule():
    """Row N3 pinned to reference CODE: tests/golden/triangulation.npz holds what the reference's own
    `triangulate_pymvg` -> `build_multi_camera_system` -> `MultiCameraSystem.find3d` (vision/triangulation.py:350-441,
    vision/multi_camera_system.py:199-225) returned for these detections (tests/golden/make_triangulation_golden.py),
    including joints whose threshold was lowered, views dropped at the float32 boundary and all-zero scores."""
    d, names = _triangulation_cases()
    assert len(names) >= 6
    for name in names:
        g = lambda k: d["%s.%s" % (name, k)]
        P = torch.from_numpy(g("K")).double() @ torch.from_numpy(g("RT")).double()          # (V,3,4), as CameraModel.get_M
        got = triangulate_dlt(torch.from_numpy(g("pts"))[None], P[None], torch.from_numpy(g("conf"))[None],
                              conf_thres=float(g("thres")))[0].numpy()
        want = g("X_ref")
        err = np.linalg.norm(got - want, axis=1)
        # the reference camera keeps R as a quaternion: its M differs from K @ RT by ~2e-8 relative (float32 R is not
        # exactly orthonormal), i.e. ~1e-4 mm at 5 m; joints whose selected views nearly agree are worse conditioned
        scale = np.maximum(1.0, np.linalg.norm(want - g("X_true"), axis=1))
        assert (err <= 2e-3 * scale).all(), (name, float(err.max()), int(err.argmax()))
