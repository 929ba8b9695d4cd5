#!/usr/bin/env python
"""Time the fused forward kernel (and optionally backward) for a list of
EtLayerDesc.variant values on the Config-2 workload; verifies each variant
against variant 0.  Usage: python scripts/sweep_variants.py 0 4 8 24 ..."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epipolar_transformers_amd import camera, ops, synthetic as syn  # noqa: E402


def main():
    variants = [int(v) for v in sys.argv[1:] if not v.startswith("-")] or [0]
    H = int(os.environ.get("SWEEP_HW", 64)); C = int(os.environ.get("SWEEP_C", 256)); K = int(os.environ.get("SWEEP_K", 64))
    frames = int(os.environ.get("SWEEP_FRAMES", 32)); views = int(os.environ.get("SWEEP_VIEWS", 4))
    dev = torch.device("cuda:0")
    P1, P2 = syn.make_pairs(frames, views, H * 4, seed=1000, jitter=(0.05, 8.0))
    n = P1.shape[0]
    g = torch.Generator(device=dev).manual_seed(0)
    ref = torch.randn(n, H, H, C, device=dev, generator=g).relu_()
    src = torch.randn(n, H, H, C, device=dev, generator=g).relu_()
    cam = camera.pair_algebra(P1, P2).to(dev)
    base = None
    rows = []
    for v in variants:
        spec = ops.LayerSpec(H=H, W=H, K=K, variant=v)
        out, attn, corr = ops.forward_nhwc(spec, ref, src, cam)
        torch.cuda.synchronize()
        if base is None:
            base = (out.clone(), attn.clone())
        err = ((out - base[0]).abs().max().item(), (attn - base[1]).abs().max().item())
        for _ in range(3):
            ops.forward_nhwc(spec, ref, src, cam)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record(); ops.forward_nhwc(spec, ref, src, cam); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        # attn / corr outputs off: how much do the side outputs cost?
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev2:
            a.record(); ops.forward_nhwc(spec, ref, src, cam, want_attn=False, want_corr=False); b.record()
        torch.cuda.synchronize()
        t2 = sorted(a.elapsed_time(b) for a, b in ev2)
        rows.append(dict(variant=v, ms_med=ts[len(ts) // 2], ms_min=ts[0], ms_noattn=t2[len(t2) // 2], max_err_out=err[0], max_err_attn=err[1]))
        print("variant %3d  fwd %.3f ms (min %.3f)  no-attn %.3f ms  |d out| %.2e |d attn| %.2e" %
              (v, ts[len(ts) // 2], ts[0], t2[len(t2) // 2], err[0], err[1]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "sweep_%s.json" % os.environ.get("SWEEP_TAG", "latest")), "w") as fh:
        json.dump(dict(H=H, C=C, K=K, pairs=n, rows=rows), fh, indent=1)


if __name__ == "__main__":
    main()
