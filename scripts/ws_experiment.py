#!/usr/bin/env python
"""Which role bounds the warp-specialised forward (development tool).  Needs the profiling build
(`python -m epipolar_transformers_amd.build --profile`, run with EPIPOLAR_AMD_LIB=.../libepipolar_amd_prof.so): its host
wrapper reads ET_WS_EXPERIMENT per call -- bit 128: no G1, 64: no G2, 32: no S1, 16: no A-stage copy, 8: no SM (results are wrong, timing only), 2: tiles assigned
statically (block jx of an XCD takes tiles jx, jx + nbx, ..) instead of drawn from the XCD's counter (results stay right).
Round 5 used it to settle what the "skeleton" time is: see profiles/r05_ws_role_experiment.txt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolar_transformers_amd import camera, ops, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
H, C, K = 64, 256, 64
P1, P2 = syn.make_pairs(32, 4, H * 4, seed=1000, jitter=(0.05, 8.0))
g = torch.Generator(device=dev).manual_seed(0)
ref = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
src = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
cam = camera.pair_algebra(P1, P2).to(dev)
spec = ops.LayerSpec(H=H, W=H, K=K)


def events_ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(t) / len(t), t[0]


for name, bits in (("everything", 0), ("no G1", 128), ("no G2", 64), ("vector waves only (no G1, no G2)", 192),
                   ("no SM", 8), ("no SM, no G2", 72), ("S1 + S2 + copy only", 200), ("S1 + S2 only", 216), ("copy only", 232),
                   ("skeleton (barriers, tile walk, operand prefetches)", 248),
                   ("skeleton, tiles assigned statically (no atomic draw)", 250),
                   ("everything, tiles assigned statically", 2)):
    # (round 4 also ran "no S1" (32) and "no copy" (16): without S1 the pixel ids G2 stores through are whatever LDS held --
    #  a wild store, a memory fault in round 5 -- so they are gone)
    os.environ["ET_WS_EXPERIMENT"] = str(bits)
    m, lo = events_ms(lambda: ops.forward_nhwc(spec, ref, src, cam))
    print("%-36s forward call %.3f ms (min %.3f)" % (name, m, lo), flush=True)
