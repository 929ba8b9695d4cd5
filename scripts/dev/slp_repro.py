"""Development: the parked split-fp16 one-block-per-tile forward (scripts/dev/classic_split_fp16.diff) built with and without
SLP vectorisation -- does the intermittent fault follow the packed-fp32 instructions of the tap arithmetic?"""
import sys, torch
sys.path.insert(0, ".")
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn
dev = torch.device("cuda:0")
H, K, N = 64, 64, 4
P1, P2 = syn.make_pairs(1, 4, H * 4, seed=3, jitter=(0.05, 8.0)); P1, P2 = P1[:N], P2[:N]
f1, f2 = syn.make_features(N, 256, H, H, seed=5)
ref = f1.permute(0, 2, 3, 1).contiguous().to(dev); src = f2.permute(0, 2, 3, 1).contiguous().to(dev)
cam = camera.pair_algebra(P1, P2).to(dev)
o0, a0, c0 = ops.forward_nhwc(ops.LayerSpec(H=H, W=H, K=K, variant=_lib.ET_VARIANT_NO_TILE), ref, src, cam)
cnt = []
for rep in range(10):
    o, a, c = ops.forward_nhwc(ops.LayerSpec(H=H, W=H, K=K, variant=65536), ref, src, cam)
    cnt.append(int(((a - a0).abs().amax(1) > 1e-4).sum()))
print(sys.argv[1] if len(sys.argv) > 1 else "", "bad pixels per run", cnt)
