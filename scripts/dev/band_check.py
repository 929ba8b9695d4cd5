#!/usr/bin/env python
"""Development: the band-table instance of the persistent kernel (maps above 64 x 64; ET_VARIANT_WS_BAND forces it on small maps).
  1. 64 x 64: the forced band instance against the default one -- bit for bit (same row order, same arithmetic);
  2. 96 x 96: the band instance (default there) against the one-block-per-tile kernel and the per-pixel kernels;
  3. timing at the bench batch (128 pairs, 96 x 96, K = 64): persistent against one-block-per-tile, forward and one-kernel layer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn

dev = torch.device("cuda:0")
C = 256
ops.POISON_OUTPUTS = True


def inputs(n, h, seed):
    P1, P2 = syn.make_pairs((n + 3) // 4, 4, 4 * h, seed=seed, jitter=(0.05, 8.0))
    g = torch.Generator(device=dev).manual_seed(seed)
    ref = torch.randn(n, h, h, C, device=dev, generator=g).relu_()
    src = torch.randn(n, h, h, C, device=dev, generator=g).relu_()
    return ref, src, camera.pair_algebra(P1[:n], P2[:n]).to(dev)


def ovf(ws):
    base = (-ws.data_ptr()) % 256
    return int(ws[base:base + 4].view(torch.int32).item())


def cmp(a, b):
    return tuple(("%.2e" % (x - y).abs().max().item()) for x, y in zip(a, b))


# 1. small maps, forced
for (n, h, k) in ((8, 64, 64), (5, 48, 33), (4, 16, 16)):
    ref, src, cam = inputs(n, h, 100 + h)
    s0, s1 = ops.LayerSpec(H=h, W=h, K=k), ops.LayerSpec(H=h, W=h, K=k, variant=_lib.ET_VARIANT_WS_BAND)
    ws = ops.tile_workspace(s1, n, C, dev)
    a = ops.forward_nhwc(s0, ref, src, cam)
    b = ops.forward_nhwc(s1, ref, src, cam, workspace=ws)
    torch.cuda.synchronize()
    print("%dx%d K=%d forced band instance vs default: equal %s  max diff (out, attn, corr) %s  overflow tiles %d"
          % (h, h, k, all(torch.equal(x, y) for x, y in zip(a, b)), cmp(a, b), ovf(ws)), flush=True)
# 2. 96 x 96
for (n, h, k) in ((4, 96, 64), (3, 80, 40)):
    ref, src, cam = inputs(n, h, 200 + h)
    s0 = ops.LayerSpec(H=h, W=h, K=k)
    ws = ops.tile_workspace(s0, n, C, dev)
    a = ops.forward_nhwc(s0, ref, src, cam, workspace=ws)
    b = ops.forward_nhwc(ops.LayerSpec(H=h, W=h, K=k, variant=_lib.ET_VARIANT_TILE_CLASSIC), ref, src, cam)
    c = ops.forward_nhwc(ops.LayerSpec(H=h, W=h, K=k, variant=_lib.ET_VARIANT_NO_TILE), ref, src, cam)
    torch.cuda.synchronize()
    st = ops.tile_stats(s0, n, C, ws)
    print("%dx%d K=%d persistent vs one-block-per-tile %s  vs per-pixel %s  corr mismatch %.5f  overflow tiles %d  U max %d"
          % (h, h, k, cmp(a, b), cmp(a, c), (a[2] != c[2]).any(-1).float().mean().item(), ovf(ws), int((st & 0xffff).max())), flush=True)
    if ops.fused_layer_applies(s0, C, n):
        g = torch.Generator(device=dev).manual_seed(5)
        wf = torch.randn(C, C, device=dev, generator=g) * 0.05 + torch.eye(C, device=dev)
        bias = torch.randn(C, device=dev, generator=g)
        packed = ops.residual_gemm_pack(wf)
        x2 = ops.residual_gemm(a[0], packed, bias, ref)
        x1, a1, c1, o1 = ops.forward_fused_nhwc(s0, ref, src, cam, packed, bias, want_out=True)
        torch.cuda.synchronize()
        print("   one-kernel layer: attn/corr/out equal %s  x vs two kernels %.2e" % (torch.equal(a1, a[1]) and torch.equal(c1, a[2]) and torch.equal(o1, a[0]), (x1 - x2).abs().max().item()), flush=True)
ops.check_tile_errors()
# 3. timing
if os.environ.get("BAND_TIME", "1") == "1":
    ops.POISON_OUTPUTS = False
    n, h, k = 128, 96, 64
    ref, src, cam = inputs(n, h, 1000)
    g = torch.Generator(device=dev).manual_seed(5)
    packed = ops.residual_gemm_pack(torch.randn(C, C, device=dev, generator=g) * 0.05 + torch.eye(C, device=dev))
    bias = torch.randn(C, device=dev, generator=g)

    def timed(f, reps=15):
        for _ in range(3):
            f()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        return sum(t) / len(t), t[0]
    s0 = ops.LayerSpec(H=h, W=h, K=k)
    s1 = ops.LayerSpec(H=h, W=h, K=k, variant=_lib.ET_VARIANT_TILE_CLASSIC)
    ws = ops.tile_workspace(s0, n, C, dev)
    print("96x96 K=64, 128 pairs: persistent %.3f ms (min %.3f) | one-block-per-tile %.3f ms (min %.3f)"
          % (timed(lambda: ops.forward_nhwc(s0, ref, src, cam, workspace=ws)) + timed(lambda: ops.forward_nhwc(s1, ref, src, cam))), flush=True)
    print("   overflow tiles %d" % ovf(ws))
    def two():
        o = ops.forward_nhwc(s0, ref, src, cam, workspace=ws)[0]
        return ops.residual_gemm(o, packed, bias, ref)
    print("   one-kernel layer %.3f ms (min %.3f) | persistent + residual GEMM %.3f ms (min %.3f)"
          % (timed(lambda: ops.forward_fused_nhwc(s0, ref, src, cam, packed, bias, workspace=ws)) + timed(two)), flush=True)
    ops.check_tile_errors()
