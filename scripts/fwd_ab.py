#!/usr/bin/env python
"""Lean A/B timing of the fused forward at Config 2 (development): one process per library variant.
    EPIPOLAR_AMD_LIB=.../libepipolar_amd_X.so python scripts/fwd_ab.py [label]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("EPIPOLAR_AMD_LIB", "default"))
dev = torch.device("cuda:0")
H, C, K = int(os.environ.get("AB_HW", "64")), 256, int(os.environ.get("AB_K", "64"))      # (Config 5: AB_HW=128 AB_K=128 AB_PAIRS=64 AB_VIEWS=8)
NP, V = int(os.environ.get("AB_PAIRS", "128")), int(os.environ.get("AB_VIEWS", "4"))
RIG = os.environ.get("AB_RIG", "ring")           # (ring | h36m_room | epipole_inside | epipole_border | near_rectified_y: synthetic.rig_pairs)
if RIG == "ring":
    P1, P2 = syn.make_pairs(NP // V, V, H * 4, seed=1000, jitter=(0.05, 8.0))
else:
    P1, P2 = syn.rig_pairs(RIG, NP // (4 if RIG == "h36m_room" else 2), 4 * H, seed=1000, jitter=None if RIG == "epipole_border" else (0.05, 8.0))
    label += " [%s]" % RIG
g = torch.Generator(device=dev).manual_seed(0)
ref = torch.randn(NP, H, H, C, device=dev, generator=g).relu_()
src = torch.randn(NP, H, H, C, device=dev, generator=g).relu_()
cam = camera.pair_algebra(P1, P2).to(dev)
spec = ops.LayerSpec(H=H, W=H, K=K, variant=int(os.environ.get("AB_VARIANT", "0")))
ws = ops.tile_workspace(spec, NP, C, dev)
fused = os.environ.get("AB_FUSED") == "1"
if fused:
    packed = ops.residual_gemm_pack(torch.randn(C, C, device=dev, generator=g) * 0.05 + torch.eye(C, device=dev))
    bias = torch.randn(C, device=dev, generator=g)
    fwd = lambda: ops.forward_fused_nhwc(spec, ref, src, cam, packed, bias, workspace=ws)
else:
    fwd = lambda: ops.forward_nhwc(spec, ref, src, cam, workspace=ws)
for _ in range(5):
    fwd()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
torch.cuda.synchronize()
for a, b in ev:
    a.record()
    fwd()
    b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
base = (-ws.data_ptr()) % 256
ovf = int(ws[base:base + 4].view(torch.int32).item())
o, a_, c_ = ops.forward_nhwc(spec, ref[:8], src[:8], cam[:8])
o2, a2, c2 = ops.forward_nhwc(ops.LayerSpec(H=H, W=H, K=K, variant=_lib.ET_VARIANT_NO_TILE), ref[:8], src[:8], cam[:8])
if fused:       # x against the two-kernel path
    xo = ops.residual_gemm(ops.forward_nhwc(spec, ref[:8], src[:8], cam[:8])[0], packed, bias, ref[:8])
    xf = ops.forward_fused_nhwc(spec, ref[:8], src[:8], cam[:8], packed, bias)[0]
    label += "  (fused; x vs two kernels %.1e)" % (xf - xo).abs().max().item()
print("%-28s forward call %.4f ms (min %.4f, p90 %.4f)  overflow tiles %d  | vs per-pixel: out %.2e attn %.2e corr mismatch %.5f"
      % (label, sum(t) / len(t), t[0], t[int(0.9 * len(t))], ovf, (o - o2).abs().max().item(), (a_ - a2).abs().max().item(),
         (c_ != c2).any(-1).float().mean().item()), flush=True)
