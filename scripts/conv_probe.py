#!/usr/bin/env python
"""Which convolutions of the fp32 pose trunk are slow at the larger BASELINE image sizes, and under which settings
(development tool; VERDICT r3 item 9: at 384 x 384 / 512 x 512 MIOpen's `naive_conv_ab_nonpacked_*` kernels took 80 % of
the end-to-end leg).  Times every Conv2d / ConvTranspose2d of the trunk with HIP events, per setting:

    python scripts/conv_probe.py [--body epipolarposeR-50] [--image 384] [--batch 32]

settings: memory format (channels_last / contiguous) x torch.backends.cudnn.benchmark (MIOpen find on / off) x batch.
"""
import argparse
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolar_transformers_amd import default_cfg  # noqa: E402
from epipolar_transformers_amd.model import MultiViewPoseModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--body", default="epipolarposeR-50")
ap.add_argument("--image", type=int, default=384)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--top", type=int, default=8)
ap.add_argument("--settings", default="cl0,cl1,nchw0,nchw1")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = default_cfg()
hs = args.image // 4
cfg.merge_from_list(["BACKBONE.BODY", args.body, "BACKBONE.PRETRAINED", False, "KEYPOINT.HEATMAP_SIZE", (hs, hs),
                     "KEYPOINT.NUM_PTS", 17, "KEYPOINT.NFEATS", 256, "DATASETS.IMAGE_SIZE", (args.image, args.image),
                     "EPIPOLAR.MERGE", "late", "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True,
                     "EPIPOLAR.SHARE_WEIGHTS", True])
net = MultiViewPoseModel(cfg).to(dev).eval().reference

records = {}


def hook_all(model):
    hs_ = []
    for name, m in model.named_modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            def pre(mod, inp, name=name):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                records.setdefault(name, []).append([e, None, tuple(inp[0].shape), mod])

            def post(mod, inp, out, name=name):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                records[name][-1][1] = e
            hs_ += [m.register_forward_pre_hook(pre), m.register_forward_hook(post)]
    return hs_


def run(setting):
    cl, bench = setting.startswith("cl"), setting.endswith("1")
    torch.backends.cudnn.benchmark = bench
    fmt = torch.channels_last if cl else torch.contiguous_format
    net.to(memory_format=fmt)
    img = torch.randn(args.batch, 3, args.image, args.image, device=dev).contiguous(memory_format=fmt)

    def trunk(x):       # (PoseResNet.trunk forces channels_last: restated here so that the format is the setting's)
        x = net.layer1(net.maxpool(net.relu(net.bn1(net.conv1(x)))))
        return net.deconv_layers(net.layer4(net.layer3(net.layer2(x))))

    with torch.no_grad():
        t0 = time.perf_counter()
        trunk(img)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        trunk(img)
        torch.cuda.synchronize()
        records.clear()
        hooks = hook_all(net)
        t0 = time.perf_counter()
        trunk(img)
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) * 1e3
        for h in hooks:
            h.remove()
    rows = []
    for name, recs in records.items():
        a, b, shape, mod = recs[-1]
        rows.append((a.elapsed_time(b), name, shape, mod))
    rows.sort(key=lambda r: -r[0])
    conv_ms = sum(r[0] for r in rows)
    print("== %s  (%s, cudnn.benchmark=%s, batch %d, %dx%d): trunk %.1f ms (convs %.1f ms over %d modules; first call %.1f s)"
          % (setting, "channels_last" if cl else "NCHW", bench, args.batch, args.image, args.image, total, conv_ms, len(rows), first),
          flush=True)
    for ms, name, shape, mod in rows[:args.top]:
        print("   %8.2f ms  %-28s in %s  k%s s%s %d->%d" % (ms, name, shape, tuple(mod.kernel_size), tuple(mod.stride),
                                                          mod.in_channels, mod.out_channels), flush=True)
    return total


for s in args.settings.split(","):
    run(s)
