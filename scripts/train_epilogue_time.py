#!/usr/bin/env python
"""Development: the TRAINING-mode epilogue bn(z(out)) + out + feat at Config 2 (128 x 64 x 64 rows of 256 channels) --
forward and backward of (a) the stock torch ops, (b) the GEMM-kernel forward with aten's batch-norm / convolution backward on
the 4-D tensors, (c) the GEMM-kernel forward with the 2-D rows backward (three library GEMMs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolar_transformers_amd import default_cfg  # noqa: E402
from epipolar_transformers_amd import epipolar as ep  # noqa: E402

dev = torch.device("cuda:0")
N, H, C = int(os.environ.get("TE_PAIRS", "128")), int(os.environ.get("TE_HW", "64")), 256
g = torch.Generator(device=dev).manual_seed(0)
out = torch.randn(N, H, H, C, device=dev, generator=g).relu_().permute(0, 3, 1, 2)
feat = torch.randn(N, H, H, C, device=dev, generator=g).relu_().permute(0, 3, 1, 2)
gx = torch.randn(N, H, H, C, device=dev, generator=g).permute(0, 3, 1, 2)


def timed(fn, reps=8):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / reps


for label, fused, aten, rows, libw in (("stock torch ops", False, False, False, False), ("GEMM kernels + aten 4-D backward", True, True, False, False),
                                       ("GEMM kernels + 2-D rows backward", True, False, True, False),
                                       ("GEMM kernels forward AND backward, library wgrad", True, False, False, True),
                                       ("GEMM kernels forward AND backward + et_z_wgrad (product)", True, False, False, False)):
    cfg = default_cfg()
    cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (H, H), "KEYPOINT.NFEATS", C, "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True,
                         "EPIPOLAR_AMD.FUSED_TRAIN_EPILOGUE", fused])
    torch.manual_seed(0)
    mod = ep.Epipolar(cfg=cfg).to(dev).train()
    with torch.no_grad():
        mod.bn.weight.normal_(1, 0.1)
    ep._TrainEpilogue.ATEN_BACKWARD, ep._TrainEpilogue.ROWS_BACKWARD, ep._TrainEpilogue.LIBRARY_WGRAD = aten, rows, libw
    o = out.detach().requires_grad_(True)
    f = feat.detach().requires_grad_(True)

    def fwd():
        return mod._epilogue_torch(o, f)[1]

    def step():
        o.grad = f.grad = None
        mod.zero_grad(set_to_none=True)
        fwd().backward(gx)

    t_f, t_s = timed(fwd), timed(step)
    print("%-60s forward %.3f ms   forward + backward %.3f ms   (backward %.3f)" % (label, t_f, t_s, t_s - t_f), flush=True)
ep._TrainEpilogue.ATEN_BACKWARD = ep._TrainEpilogue.ROWS_BACKWARD = ep._TrainEpilogue.LIBRARY_WGRAD = False
