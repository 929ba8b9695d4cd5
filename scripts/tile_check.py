#!/usr/bin/env python
"""GPU check and tuning report of the MFMA tile kernels (C = 256 head).

  1. forward: tile kernel vs the per-pixel kernels on a few shapes (max differences);
  2. on the Config-2 batch: timings of both, the row-set statistics of the tiles, the tiled backward vs the
     gather form, and timing ablations of the tile kernels' phases (et_debug_tile_ablate: wrong results by
     construction, used only to see where the time goes).

usage: python scripts/tile_check.py [--ablate]
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


def main():
    ablate = "--ablate" in sys.argv
    dev = torch.device("cuda:0")
    lib = _lib.load()
    cases = [(2, 64, 64, True, 4), (4, 64, 64, False, 4), (3, 32, 128, True, 4), (2, 48, 33, True, 4),
             (4, 16, 16, True, 8), (4, 96, 64, True, 4), (128, 64, 64, True, 4)]
    for (N, hw, K, sm, views) in cases:
        spec = ops.LayerSpec(H=hw, W=hw, K=K, softmax_enabled=sm)
        spec_pp = dataclasses.replace(spec, variant=_lib.ET_VARIANT_NO_TILE)
        P1, P2 = synthetic.make_pairs(max(1, N // views), views, image_size=hw * 4, seed=3, jitter=(0.05, 8.0))
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
        if N < 64:
            continue
        print("  forward: per-pixel %.3f ms, tile %.3f ms (with res_base), tile %.3f ms (out/attn/corr only)" % (
            timed(lambda: ops.forward_nhwc(spec_pp, ref, src, cam, res_bias=bias, want_res_base=True)),
            timed(lambda: ops.forward_nhwc(spec, ref, src, cam, res_bias=bias, want_res_base=True)),
            timed(lambda: ops.forward_nhwc(spec, ref, src, cam))), flush=True)
        tiles = N * ((hw * hw + 31) // 32)
        stats = torch.zeros(tiles, dtype=torch.int32, device=dev)
        lib.et_debug_tile_stats(ctypes.c_void_p(stats.data_ptr()))
        ops.forward_nhwc(spec, ref, src, cam)
        torch.cuda.synchronize()
        lib.et_debug_tile_stats(None)
        st = stats.cpu().numpy()
        rows, groups = st & 0xFFFF, st >> 16
        print("  %d tiles: source rows per tile mean %.1f p50 %d p90 %d max %d; tiles split into groups: %d" % (
            tiles, rows.mean(), np.percentile(rows, 50), np.percentile(rows, 90), rows.max(), int((groups > 1).sum())))
        go = torch.randn_like(ref)
        gr_t, gs_t = ops.backward_nhwc(spec, ref, src, cam, go, form="tile")
        gr_g, gs_g = ops.backward_nhwc(spec, ref, src, cam, go, form="gather")
        print("  backward tile vs gather: grad_ref %.2e (scale %.2e)  grad_src %.2e (scale %.2e)" % (
            (gr_t - gr_g).abs().max().item(), gr_g.abs().max().item(), (gs_t - gs_g).abs().max().item(),
            gs_g.abs().max().item()), flush=True)
        print("  backward: tile %.3f ms, gather %.3f ms" % (
            timed(lambda: ops.backward_nhwc(spec, ref, src, cam, go, form="tile"), 5, 2),
            timed(lambda: ops.backward_nhwc(spec, ref, src, cam, go, form="gather"), 5, 2)), flush=True)
        if ablate:
            try:
                for bits, what in ((1, "no first GEMM"), (2, "no second GEMM"), (4, "no soft-max phase"),
                                   (7, "set-up + stores only"), (16, "ordering kernel only")):
                    lib.et_debug_tile_ablate(bits, 0)
                    print("  forward ablation %-22s %.3f ms" % (what, timed(lambda: ops.forward_nhwc(spec, ref, src, cam))))
                for bits, what in ((1, "no grad_src atomics"), (2, "no transposed GEMMs"), (6, "no transposed GEMMs, no B rows")):
                    lib.et_debug_tile_ablate(0, bits)
                    print("  backward ablation %-30s %.3f ms" % (
                        what, timed(lambda: ops.backward_nhwc(spec, ref, src, cam, go, form="tile"), 5, 2)))
            finally:
                lib.et_debug_tile_ablate(0, 0)


if __name__ == "__main__":
    main()
