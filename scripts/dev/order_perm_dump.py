"""Development: dump the tile ordering (perm of every pair) of a few shapes to an .npz -- run once per library build
(EPIPOLAR_AMD_LIB) and compare the files: the radix sort (round 6) must give the bitonic network's permutation bit for bit
(keys are unique).    python scripts/dev/order_perm_dump.py OUT.npz    |    python scripts/dev/order_perm_dump.py --compare A.npz B.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("compared %d arrays: %s" % (len(a.files), "ALL EQUAL" if not bad else "DIFFER: %s" % bad))
    sys.exit(1 if bad else 0)
import torch
from epipolar_transformers_amd import camera, ops, synthetic as syn
out = {}
for rig in ("ring", "epipole_inside", "h36m_room"):
    for (n, h, w, k) in [(8, 64, 64, 64), (4, 16, 16, 16), (4, 15, 15, 16), (4, 20, 20, 16), (4, 40, 40, 32), (4, 48, 64, 48), (4, 33, 20, 20), (2, 96, 96, 64)]:
        if rig == "ring":
            P1, P2 = syn.make_pairs((n + 3) // 4, 4, 4 * max(h, w), seed=7 + h, jitter=(0.05, 8.0))
        else:
            P1, P2 = syn.rig_pairs(rig, n // (4 if rig == "h36m_room" else 2), 4 * max(h, w), seed=7 + h, jitter=(0.05, 8.0))
        P1, P2 = P1[:n], P2[:n]
        g = torch.Generator().manual_seed(h)
        f1 = torch.randn(n, h, w, 256, generator=g).relu_().cuda()
        f2 = torch.randn(n, h, w, 256, generator=g).relu_().cuda()
        cam = camera.pair_algebra(P1, P2).cuda()
        spec = ops.LayerSpec(H=h, W=w, K=k)
        ws = ops.tile_workspace(spec, n, 256, f1.device)
        ops.forward_nhwc(spec, f1, f2, cam, workspace=ws)
        torch.cuda.synchronize()
        tiles = n * ((h * w + 31) // 32)
        base = (-ws.data_ptr()) % 256
        words = ws[base:base + (ws.numel() - base) // 4 * 4].view(torch.int32)
        out["%s_%dx%d" % (rig, h, w)] = words[64:64 + tiles * 32].cpu().numpy().copy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
