#!/usr/bin/env python
"""GPU check and tuning report of the MFMA tile kernels (C = 256 head).

  1. forward: tile kernels (warp-specialised persistent + one-block-per-tile) vs the per-pixel kernels on a few
     shapes (max differences; the two tile kernels must agree bit for bit where both apply);
  2. on the Config-2 batch: timings of every tile-kernel variant and of the per-pixel kernels, the row-set
     statistics of the tiles, the tiled backward vs the gather form.

usage: python scripts/tile_check.py
"""
import ctypes
import dataclasses
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolar_transformers_amd import _lib, camera, ops, synthetic  # noqa: E402


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


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


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    CL, PRIO = _lib.ET_VARIANT_TILE_CLASSIC, _lib.ET_VARIANT_WS_SETPRIO
    cases = [(2, 64, 64, True, 4), (4, 64, 64, False, 4), (3, 32, 128, True, 4), (2, 48, 33, True, 4),
             (4, 16, 16, True, 8), (4, 96, 64, True, 4), (1, 64, 64, True, 4), (5, 64, 20, True, 4), (128, 64, 64, True, 4)]
    for (N, hw, K, sm, views) in cases:
        spec = ops.LayerSpec(H=hw, W=hw, K=K, softmax_enabled=sm)
        spec_pp = dataclasses.replace(spec, variant=_lib.ET_VARIANT_NO_TILE)
        spec_cl = dataclasses.replace(spec, variant=CL)
        P1, P2 = synthetic.make_pairs(max(1, (N + views - 1) // views), views, image_size=hw * 4, seed=3, jitter=(0.05, 8.0))
        P1, P2 = P1[:N], P2[:N]
        N = P1.shape[0]
        f1, f2 = synthetic.make_features(N, 256, hw, hw, seed=5)
        ref = f1.permute(0, 2, 3, 1).contiguous().to(dev)
        src = f2.permute(0, 2, 3, 1).contiguous().to(dev)
        cam = camera.pair_algebra(P1, P2).to(dev)
        bias = torch.randn(256).to(dev)
        o1, a1, c1, b1 = ops.forward_nhwc(spec_pp, ref, src, cam, res_bias=bias, want_res_base=True)
        o2, a2, c2, b2 = ops.forward_nhwc(spec, ref, src, cam, res_bias=bias, want_res_base=True)
        torch.cuda.synchronize()
        print("N=%d %dx%d K=%d softmax=%s: tile vs per-pixel  out %.2e (scale %.2e)  attn %.2e  corr_pos mismatches %.4f"
              "  res_base %.1e" % (N, hw, hw, K, sm, (o1 - o2).abs().max().item(), o1.abs().max().item(),
                                   (a1 - a2).abs().max().item(), (c1 != c2).any(-1).float().mean().item(),
                                   (b1 - b2).abs().max().item()), flush=True)
        for nm, v in (("classic", CL),):
            o3, a3, c3, b3 = ops.forward_nhwc(dataclasses.replace(spec, variant=v), ref, src, cam, res_bias=bias,
                                              want_res_base=True)
            print("    default vs %-8s out %.2e attn %.2e corr %.4f base %.1e  (bit-equal: %s)" % (
                nm, (o3 - o2).abs().max().item(), (a3 - a2).abs().max().item(), (c3 != c2).any(-1).float().mean().item(),
                (b3 - b2).abs().max().item(), bool(torch.equal(o3, o2) and torch.equal(a3, a2) and torch.equal(c3, c2))),
                flush=True)
        if N < 64:
            continue
        for nm, v in (("per-pixel", _lib.ET_VARIANT_NO_TILE), ("classic", CL), ("ws (default)", 0),
                      ("ws prio", PRIO)):
            sp = dataclasses.replace(spec, variant=v)
            m1, lo1 = events_ms(lambda: ops.forward_nhwc(sp, ref, src, cam, res_bias=bias, want_res_base=True))
            m2, lo2 = events_ms(lambda: ops.forward_nhwc(sp, ref, src, cam))
            print("  forward %-12s with res_base %.3f ms (min %.3f)   out/attn/corr only %.3f ms (min %.3f)" % (
                nm, m1, lo1, m2, lo2), flush=True)
        ws = ops.tile_workspace(spec, N, 256, dev)
        ops.forward_nhwc(spec, ref, src, cam, workspace=ws)
        torch.cuda.synchronize()
        st = ops.tile_stats(spec, N, 256, ws).cpu().numpy()
        rows, groups = st & 0xFFFF, st >> 16
        print("  %d tiles: source rows per tile mean %.1f p50 %d p90 %d max %d; tiles split into groups: %d; row blocks: %s" % (
            len(st), rows.mean(), np.percentile(rows, 50), np.percentile(rows, 90), rows.max(), int((groups > 1).sum()),
            np.bincount((rows + 31) // 32).tolist()))
        go = torch.randn_like(ref)
        gr_t, gs_t = ops.backward_nhwc(spec, ref, src, cam, go, form="tile")
        gr_g, gs_g = ops.backward_nhwc(spec, ref, src, cam, go, form="gather")
        print("  backward tile vs gather: grad_ref %.2e (scale %.2e)  grad_src %.2e (scale %.2e)" % (
            (gr_t - gr_g).abs().max().item(), gr_g.abs().max().item(), (gs_t - gs_g).abs().max().item(),
            gs_g.abs().max().item()), flush=True)
        attn = ops.forward_nhwc(spec, ref, src, cam)[1]
        gr_a, gs_a = ops.backward_nhwc(spec, ref, src, cam, go, form="tile", attn=attn)
        print("  backward tile with the forward's attention vs gather: grad_ref %.2e  grad_src %.2e" % (
            (gr_a - gr_g).abs().max().item(), (gs_a - gs_g).abs().max().item()), flush=True)
        print("  backward: tile %.3f ms, tile with attention %.3f ms, gather %.3f ms" % (
            timed(lambda: ops.backward_nhwc(spec, ref, src, cam, go, form="tile"), 5, 2),
            timed(lambda: ops.backward_nhwc(spec, ref, src, cam, go, form="tile", attn=attn), 5, 2),
            timed(lambda: ops.backward_nhwc(spec, ref, src, cam, go, form="gather"), 5, 2)), flush=True)


if __name__ == "__main__":
    main()
