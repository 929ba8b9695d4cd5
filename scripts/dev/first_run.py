import sys, torch
sys.path.insert(0, ".")
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn
dev = torch.device("cuda:0")
H, K, N = 96, 64, 8
P1, P2 = syn.make_pairs(2, 4, H * 4, seed=11, jitter=(0.05, 8.0)); P1, P2 = P1[:N], P2[:N]
f1, f2 = syn.make_features(N, 256, H, H, seed=5)
ref = f1.permute(0, 2, 3, 1).contiguous().to(dev); src = f2.permute(0, 2, 3, 1).contiguous().to(dev)
cam = camera.pair_algebra(P1, P2).to(dev)
o0, a0, c0 = ops.forward_nhwc(ops.LayerSpec(H=H, W=H, K=K, variant=_lib.ET_VARIANT_NO_TILE), ref, src, cam)
def bad_of(o, a):
    return ((a - a0).abs().amax(1) > 1e-5) | ((o - o0).abs().amax(-1) > 1e-4 * max(1.0, o0.abs().max().item()))
for label, variant, fresh in (("exact, cached ws", 65536 | 524288, False), ("split, cached ws", 0, False), ("split, cached ws", 0, False),
                              ("split, FRESH ws", 0, True), ("split, FRESH ws", 0, True), ("split, garbage ws", 0, "garbage"), ("split, cached", 0, False)):
    spec = ops.LayerSpec(H=H, W=H, K=K, variant=variant)
    ws = None
    if fresh:
        ws = ops.tile_workspace(spec, N, 256, dev)
        if fresh == "garbage":
            ws.view(torch.int32)[: ws.numel() // 4 - 64].fill_(0x7f7f7f7f)   # everything but the tail
    o, a, c = ops.forward_nhwc(spec, ref, src, cam, workspace=ws)
    bad = bad_of(o, a)
    info = ""
    if bad.any():
        n, y, x = bad.nonzero()[0].tolist()
        info = " first bad (%d,%d,%d): attn sum %.4f, attn[:4] %s ref %s; out err %.3e" % (n, y, x, a[n, :, y, x].sum().item(), [round(v, 4) for v in a[n, :4, y, x].tolist()], [round(v, 4) for v in a0[n, :4, y, x].tolist()], (o - o0)[n, y, x].abs().max().item())
        info += " ; bad per pair %s" % bad.flatten(1).sum(1).tolist()
    print("%-20s bad pixels %d%s" % (label, int(bad.sum()), info))
