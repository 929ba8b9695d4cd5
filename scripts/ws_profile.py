#!/usr/bin/env python
"""Where the cycles of the warp-specialised forward go (development tool).

Needs a profiling build of the library:  ET_EXTRA_HIPCC_FLAGS=-DET_WS_PROFILE python -m epipolar_transformers_amd.build
(force a rebuild of et_forward_tile.hip).  The kernel then accumulates s_memtime deltas per pipeline segment and
wave; this script prints their per-tile means on the Config-2 batch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn  # noqa: E402

lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(raw, "et_dev_ws_profile"):
    raise SystemExit("not a profiling build")
dev = torch.device("cuda:0")
variant = int(os.environ.get("PROF_VARIANT", 0))
nv = 8
H, C, K = 64, 256, 64
P1, P2 = syn.make_pairs(32, 4, H * 4, seed=1000, jitter=(0.05, 8.0))
g = torch.Generator(device=dev).manual_seed(0)
ref = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
src = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
cam = camera.pair_algebra(P1, P2).to(dev)
spec = ops.LayerSpec(H=H, W=H, K=K, variant=variant)
bias = torch.randn(C, device=dev)
for _ in range(3):
    ops.forward_nhwc(spec, ref, src, cam, res_bias=bias, want_res_base=True)
prof = torch.zeros(256 * (4 + nv) * 12, dtype=torch.int64, device=dev)
raw.et_dev_ws_profile(ctypes.c_void_p(prof.data_ptr()))
ops.forward_nhwc(spec, ref, src, cam, res_bias=bias, want_res_base=True)
torch.cuda.synchronize()
raw.et_dev_ws_profile(None)
p = prof.cpu().numpy().reshape(256, 4 + nv, 12).astype(np.float64) / 64.0      # cycles per tile (64 tiles per block)
m, v = p[:, :4], p[:, 4:]
names_m = ["G2 prefetch", "wait A", "G2", "G1 prefetch", "wait B", "copy load", "G1", "matrix barrier", "copy finish"]
names_v = ["SM attn store", "S1", "arrive", "S2 (last wave)", "wait A", "copy finish", "wait B", "SM tail (B rows, phase A)", "SM front", "SM softmax", "SM corr"]
print("cycles per tile (mean over blocks; per wave index)")
if os.environ.get("WS_PROFILE_LIGHT"):
    ma = m[:, :, 5] + m[:, :, 6] + m[:, :, 0]; mb = m[:, :, 3]
    va = v[:, :, 1] + v[:, :, 3] + v[:, :, 7]; vb = v[:, :, 0] + v[:, :, 5]
    r = lambda a: np.round(a.mean(0)).astype(int).tolist()
    print("matrix waves: phase A work %s  wait A %s  phase B work %s  wait B %s" % (r(ma), r(m[:, :, 1]), r(mb), r(m[:, :, 4])))
    print("vector waves: phase A work %s  wait A %s  phase B work %s  wait B %s" % (r(va), r(v[:, :, 4]), r(vb), r(v[:, :, 6])))
    print("  total per tile: %s" % r(m[:, :, :9].sum(2)))
    raise SystemExit(0)
print("matrix waves: " + "  ".join("%s %s" % (n, np.round(m[:, :, k].mean(0)).astype(int).tolist()) for k, n in enumerate(names_m)))
print("  total per tile: %s" % np.round(m[:, :, :9].sum(2).mean(0)).astype(int).tolist())
print("vector waves: " + "  ".join("%s %s" % (n, np.round(v[:, :, k].mean(0)).astype(int).tolist()) for k, n in enumerate(names_v)))
print("  total per tile: %s" % np.round(v.sum(2).mean(0)).astype(int).tolist())
