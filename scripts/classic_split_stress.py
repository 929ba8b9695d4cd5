"""Development / acceptance: repeated runs of the one-block-per-tile forward with split-fp16 GEMMs (ET_VARIANT_TILE_CLASSIC, and
the shapes that take it by default) against the per-pixel kernels -- the intermittent fault of round 3 showed as a handful of
wrong attention rows per run (scripts/dev/README.md).  Prints the number of bad pixels of every run; all must be 0."""
import sys, torch
sys.path.insert(0, ".")
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn
dev = torch.device("cuda:0")
ops.POISON_OUTPUTS = True            # outputs start as NaN: unwritten pixels cannot hide behind a recycled allocation
total_bad = 0
for H, K, N, variant, reps in ((64, 64, 16, 65536, 12), (64, 64, 128, 65536, 3), (96, 64, 8, 0, 8), (96, 64, 32, 0, 2), (128, 128, 4, 0, 4), (64, 33, 8, 65536, 6)):
    P1, P2 = syn.make_pairs((N + 3) // 4, 4, H * 4, seed=3 + N, jitter=(0.05, 8.0)); P1, P2 = P1[:N], P2[:N]
    f1, f2 = syn.make_features(N, 256, H, H, seed=5)
    ref = f1.permute(0, 2, 3, 1).contiguous().to(dev); src = f2.permute(0, 2, 3, 1).contiguous().to(dev)
    cam = camera.pair_algebra(P1, P2).to(dev)
    o0, a0, c0 = ops.forward_nhwc(ops.LayerSpec(H=H, W=H, K=K, variant=_lib.ET_VARIANT_NO_TILE), ref, src, cam)
    cnt = []
    for rep in range(reps):
        o, a, c = ops.forward_nhwc(ops.LayerSpec(H=H, W=H, K=K, variant=variant), ref, src, cam)
        bad = ~(((a - a0).abs().amax(1) <= 1e-5) & ((o - o0).abs().amax(-1) <= 1e-4 * max(1.0, o0.abs().max().item())) & ~torch.isnan(c).any(-1))
        cnt.append(int(bad.sum()))
    total_bad += sum(cnt)
    print("%3dx%-3d K=%-3d N=%-3d variant %-6d bad pixels per run %s   max |attn diff| %.2e  max |out diff| %.2e" % (H, H, K, N, variant, cnt, (a - a0).abs().max().item(), (o - o0).abs().max().item()))
print("TOTAL BAD", total_bad)
sys.exit(1 if total_bad else 0)
