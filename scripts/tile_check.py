"""GPU check of the MFMA tile forward against the per-pixel kernels (and timing of both)."""
import dataclasses
import sys
import time

import torch

sys.path.insert(0, ".")
from epipolar_transformers_amd import _lib, camera, ops, synthetic

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    cases = [(2, 64, 64, True, 4), (4, 64, 64, False, 4), (3, 32, 128, True, 4), (2, 48, 33, True, 4),
             (4, 16, 16, True, 8), (4, 96, 64, True, 4), (128, 64, 64, True, 4)]
    for (frames_views, hw, K, sm, views) in cases:
        N = frames_views
        spec = ops.LayerSpec(H=hw, W=hw, K=K, softmax_enabled=sm)
        frames = max(1, N // views)
        P1, P2 = synthetic.make_pairs(frames, views, image_size=hw * 4, seed=3, jitter=(0.05, 8.0))
        P1, P2 = P1[:N], P2[:N]
        N = P1.shape[0]
        f1, f2 = synthetic.make_features(N, 256, hw, hw, seed=5)
        ref = f1.permute(0, 2, 3, 1).contiguous().to(dev)
        src = f2.permute(0, 2, 3, 1).contiguous().to(dev)
        cam = camera.pair_algebra(P1, P2).to(dev)
        spec_pp = dataclasses.replace(spec, variant=_lib.ET_VARIANT_NO_TILE)
        bias = torch.randn(256).to(dev)
        o1, a1, c1, b1 = ops.forward_nhwc(spec_pp, ref, src, cam, res_bias=bias, want_res_base=True)
        o2, a2, c2, b2 = ops.forward_nhwc(spec, ref, src, cam, res_bias=bias, want_res_base=True)
        torch.cuda.synchronize()
        print("N=%d HW=%d K=%d softmax=%s: out maxdiff %.3e (scale %.3e) attn %.3e corr %.3e base %.3e nan %d" % (
            N, hw, K, sm, (o1 - o2).abs().max().item(), o1.abs().max().item(), (a1 - a2).abs().max().item(),
            (c1 - c2).abs().max().item(), (b1 - b2).abs().max().item(), int(torch.isnan(o2).sum())), flush=True)
        if N >= 64:
            for name, sp in (("per-pixel", spec_pp), ("tile", spec)):
                for _ in range(3):
                    ops.forward_nhwc(sp, ref, src, cam, res_bias=bias, want_res_base=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    ops.forward_nhwc(sp, ref, src, cam, res_bias=bias, want_res_base=True)
                torch.cuda.synchronize()
                print("  %s: %.3f ms" % (name, (time.perf_counter() - t0) * 100), flush=True)
            # tuning aids: row-set statistics and phase ablations (wrong results by construction)
            import ctypes, os
            lib = _lib.load()
            tiles = N * ((hw * hw + 31) // 32)
            stats = torch.zeros(tiles, dtype=torch.int32, device=dev)
            lib.et_debug_tile_stats(ctypes.c_void_p(stats.data_ptr()))
            ops.forward_nhwc(spec, ref, src, cam)
            torch.cuda.synchronize()
            lib.et_debug_tile_stats(None)
            st = stats.cpu().numpy()
            U, ng = st & 0xFFFF, st >> 16
            import numpy as np
            print("  tiles %d: U mean %.1f p50 %d p90 %d max %d; groups>1: %d tiles (max %d)" % (
                tiles, U.mean(), np.percentile(U, 50), np.percentile(U, 90), U.max(), int((ng > 1).sum()), ng.max()))
            for wb in (True, False):
                for _ in range(3):
                    ops.forward_nhwc(spec, ref, src, cam, res_bias=bias if wb else None, want_res_base=wb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    ops.forward_nhwc(spec, ref, src, cam, res_bias=bias if wb else None, want_res_base=wb)
                torch.cuda.synchronize()
                print("  tile, res_base %s: %.3f ms" % (wb, (time.perf_counter() - t0) * 50), flush=True)
            go = torch.randn_like(ref)
            gr_t, gs_t = ops.backward_nhwc(spec, ref, src, cam, go, form="tile")
            gr_g, gs_g = ops.backward_nhwc(spec, ref, src, cam, go, form="gather")
            print("  bwd tile vs gather: grad_ref %.3e (scale %.3e) grad_src %.3e (scale %.3e)" % (
                (gr_t - gr_g).abs().max().item(), gr_g.abs().max().item(), (gs_t - gs_g).abs().max().item(),
                gs_g.abs().max().item()), flush=True)
            for ab in ():
                os.environ["ET_BTILE_ABLATE"] = str(ab)
                for _ in range(2):
                    ops.backward_nhwc(spec, ref, src, cam, go, form="tile")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    ops.backward_nhwc(spec, ref, src, cam, go, form="tile")
                torch.cuda.synchronize()
                print("  bwd tile ablate %d: %.3f ms" % (ab, (time.perf_counter() - t0) * 200), flush=True)
            os.environ["ET_BTILE_ABLATE"] = "0"
            for form in ("tile", "gather"):
                for _ in range(2):
                    ops.backward_nhwc(spec, ref, src, cam, go, form=form)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    ops.backward_nhwc(spec, ref, src, cam, go, form=form)
                torch.cuda.synchronize()
                print("  bwd %s: %.3f ms" % (form, (time.perf_counter() - t0) * 200), flush=True)
            for ab, sg in ((0, 0),):
                os.environ["ET_TILE_ABLATE"] = str(ab)
                os.environ["ET_TILE_STAGGER"] = str(sg)
                for _ in range(2):
                    ops.forward_nhwc(spec, ref, src, cam, res_bias=bias, want_res_base=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    ops.forward_nhwc(spec, ref, src, cam, res_bias=bias, want_res_base=True)
                torch.cuda.synchronize()
                print("  ablate %2d stagger %d: %.3f ms" % (ab, sg, (time.perf_counter() - t0) * 100), flush=True)
            os.environ["ET_TILE_ABLATE"] = "0"
