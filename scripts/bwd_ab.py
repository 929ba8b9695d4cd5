#!/usr/bin/env python
"""Lean A/B timing of the tiled backward at Config 2 (development): one process per library variant.
    EPIPOLAR_AMD_LIB=.../libepipolar_amd_X.so python scripts/bwd_ab.py [label]
Prints the call (memset + tile order + kernel) with the forward's attention, and the error against the gather form on 8 pairs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("EPIPOLAR_AMD_LIB", "default"))
dev = torch.device("cuda:0")
H, C, K = (int(os.environ.get(k, v)) for k, v in (("AB_H", 64), ("AB_C", 256), ("AB_K", 64)))
N = int(os.environ.get("AB_N", 128))
RIG = os.environ.get("AB_RIG", "ring")           # (ring | h36m_room | epipole_inside | epipole_border | near_rectified_y: synthetic.rig_pairs)
if RIG == "ring":
    P1, P2 = syn.make_pairs(N // 4, 4, H * 4, seed=1000, jitter=(0.05, 8.0))
else:
    P1, P2 = syn.rig_pairs(RIG, N // (4 if RIG == "h36m_room" else 2), 4 * H, seed=1000, jitter=None if RIG == "epipole_border" else (0.05, 8.0))
label += " [%s]" % RIG
g = torch.Generator(device=dev).manual_seed(0)
ref = torch.randn(N, H, H, C, device=dev, generator=g).relu_()
src = torch.randn(N, H, H, C, device=dev, generator=g).relu_()
gout = torch.randn(N, H, H, C, device=dev, generator=g)
cam = camera.pair_algebra(P1, P2).to(dev)
spec = ops.LayerSpec(H=H, W=H, K=K, variant=int(os.environ.get("AB_VARIANT", "0")))      # (2097152: split every over-capacity tile in place)
attn = ops.forward_nhwc(spec, ref, src, cam)[1]
bwd = lambda: ops.backward_nhwc(spec, ref, src, cam, gout, attn=attn)
for _ in range(3):
    bwd()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
torch.cuda.synchronize()
for a, b in ev:
    a.record()
    bwd()
    b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
hdr = ops.backward_deferred_tiles(dev, header=True)
gr, gs = ops.backward_nhwc(spec, ref[:8], src[:8], cam[:8], gout[:8], attn=attn[:8].contiguous())
gr2, gs2 = ops.backward_nhwc(spec, ref[:8], src[:8], cam[:8], gout[:8], form="gather")
print("%-28s backward call %.4f ms (min %.4f, p90 %.4f) | vs gather form: d_ref %.2e of %.2e, d_src %.2e of %.2e | header %s"
      % (label, sum(t) / len(t), t[0], t[int(0.9 * len(t))], (gr - gr2).abs().max().item(), gr2.abs().max().item(),
         (gs - gs2).abs().max().item(), gs2.abs().max().item(), hdr), flush=True)
