#!/usr/bin/env python
"""Launch the fused forward (default), the one-kernel layer (PROF_KERNEL=fused) or the backward kernel a few times on the
Config-2 workload -- the target of rocprofv3 --pmc passes (scripts/gpu_pmc.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epipolar_transformers_amd import camera, ops, synthetic as syn  # noqa: E402

variant = int(os.environ.get("PROF_VARIANT", 0))
which = os.environ.get("PROF_KERNEL", "fwd")
H, C, K = int(os.environ.get("PROF_HW", 64)), 256, int(os.environ.get("PROF_K", 64))
dev = torch.device("cuda:0")
NP, V = int(os.environ.get("PROF_PAIRS", 128)), int(os.environ.get("PROF_VIEWS", 4))      # (Config 5: PROF_PAIRS=64 PROF_VIEWS=8)
RIG = os.environ.get("PROF_RIG", "ring")                                                   # (as scripts/bwd_ab.py's AB_RIG)
if RIG == "ring":
    P1, P2 = syn.make_pairs(NP // V, V, H * 4, seed=1000, jitter=(0.05, 8.0))
else:
    P1, P2 = syn.rig_pairs(RIG, NP // (4 if RIG == "h36m_room" else 2), 4 * H, seed=1000, jitter=None if RIG == "epipole_border" else (0.05, 8.0))
g = torch.Generator(device=dev).manual_seed(0)
ref = torch.randn(NP, H, H, C, device=dev, generator=g).relu_()
src = torch.randn(NP, H, H, C, device=dev, generator=g).relu_()
cam = camera.pair_algebra(P1, P2).to(dev)
spec = ops.LayerSpec(H=H, W=H, K=K, variant=variant)
attn = ops.forward_nhwc(spec, ref, src, cam)[1] if which == "bwd" else None      # what autograd hands the backward
if which == "fused":       # the one-kernel layer: sampling + attention + z / BN / residual GEMM
    packed = ops.residual_gemm_pack(torch.randn(C, C, device=dev, generator=g) * 0.05 + torch.eye(C, device=dev))
    bias = torch.randn(C, device=dev, generator=g)
for _ in range(int(os.environ.get("PROF_REPS", 3))):
    if which == "fwd":
        ops.forward_nhwc(spec, ref, src, cam)
    elif which == "fused":
        ops.forward_fused_nhwc(spec, ref, src, cam, packed, bias)
    else:
        ops.backward_nhwc(spec, ref, src, cam, ref, attn=attn)
torch.cuda.synchronize()
