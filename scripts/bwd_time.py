#!/usr/bin/env python
"""Time the tiled backward on the Config-2 batch (development): python scripts/bwd_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolar_transformers_amd import camera, ops, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
H, C, K = int(os.environ.get("PROF_HW", 64)), 256, int(os.environ.get("PROF_K", 64))
P1, P2 = syn.make_pairs(32, 4, H * 4, seed=1000, jitter=(0.05, 8.0))
g = torch.Generator(device=dev).manual_seed(0)
ref = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
src = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
go = torch.randn(128, H, H, C, device=dev, generator=g)
cam = camera.pair_algebra(P1, P2).to(dev)
spec = ops.LayerSpec(H=H, W=H, K=K)
attn = ops.forward_nhwc(spec, ref, src, cam)[1]


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("%s: backward tile %.3f ms, with the forward's attention %.3f ms" % (
    os.environ.get("EPIPOLAR_AMD_LIB", "product library"),
    timed(lambda: ops.backward_nhwc(spec, ref, src, cam, go, form="tile")),
    timed(lambda: ops.backward_nhwc(spec, ref, src, cam, go, form="tile", attn=attn))))
