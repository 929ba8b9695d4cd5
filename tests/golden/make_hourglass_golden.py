"""Golden vectors for the stacked-hourglass callers of the layer, from the REAL reference's
modeling/backbones/ProHG.py (HourGlassNet behind registry names epipolarHG1 / epipolarHG).

    python tests/golden/make_hourglass_golden.py

Writes tests/golden/hourglass_<case>.npz: images of 4 (reference, source) pairs, the projection matrices, the reference's per-pair
algebra on this machine, and what `net(img, other_inputs=[other_features, other_KRT, None, KRT, camera, other_camera, other_img])`
returns in eval mode -- `other_features` being `net(other_img)[0]`, the list of per-stack maps, as modeling/model.py:241-247 obtains
them.  Weights: tests/golden/model_weights.py (rebuilt from the parameter names on both sides).  Runs only in the build container."""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from model_weights import deterministic_state_dict  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402
from epipolar_transformers_amd import synthetic as syn  # noqa: E402

SIZE, HS, K, J, N = 64, 16, 16, 7, 2
CASES = [dict(name="hg1_late", body="epipolarHG1", merge="late"),          # one stack, fusion behind the stack
         dict(name="hg3_early", body="epipolarHG", merge="early"),         # three stacks, fusion in front of every stack
         dict(name="hg11_both", body="epipolarHG11", merge="both")]        # depth-1 hourglass, both fusion points


def calm(net, img):
    """The generic deterministic weights let activations grow through 30+ pre-activation modules (feature maps of 1e2-1e4, a
    saturated soft-max).  Rescale, from measurements of this very net: the stem's output and every stack's output to O(1), the
    cross-stack connections to a fraction of that."""
    sd = net.state_dict()
    scales = []
    with torch.no_grad():
        for k in sd:
            if k.startswith(("trsfeas.", "trstmps.")):
                sd[k] *= 0.1
        x = net.ress(net.conv(img))
        s0 = 2.0 / float(x.abs().max())
        scales.append(s0)
        for k in ("ress.3.conv_C.2.weight", "ress.3.conv_C.2.bias", "ress.3.branch.2.weight", "ress.3.branch.2.bias"):
            sd[k] *= s0
        x = x * s0
        for i in range(net.nStack):
            f = net.features[i](x)
            si = 2.0 / float(f.abs().max())
            scales.append(si)
            sd["features.%d.3.weight" % i] *= si
            sd["features.%d.3.bias" % i] *= si
    return scales


def apply_weight_scales(sd, scales, stacks):
    """what calm() did to deterministic_state_dict's tensors, from the stored factors (tests/test_gpu_hourglass.py)"""
    for k in sd:
        if k.startswith(("trsfeas.", "trstmps.")):
            sd[k] *= 0.1
    for k in ("ress.3.conv_C.2.weight", "ress.3.conv_C.2.bias", "ress.3.branch.2.weight", "ress.3.branch.2.bias"):
        sd[k] *= float(scales[0])
    for i in range(stacks):
        sd["features.%d.3.weight" % i] *= float(scales[1 + i])
        sd["features.%d.3.bias" % i] *= float(scales[1 + i])
    return sd


def run_case(c):
    ov = ["BACKBONE.BODY", c["body"], "BACKBONE.PRETRAINED", "False", "EPIPOLAR.PRETRAINED", "False", "EPIPOLAR.MERGE", c["merge"],
          "KEYPOINT.HEATMAP_SIZE", "(%d, %d)" % (HS, HS), "KEYPOINT.NUM_PTS", str(J), "KEYPOINT.SIGMA", "2.0", "KEYPOINT.NFEATS", "256",
          "DATASETS.IMAGE_SIZE", "(%d, %d)" % (SIZE, SIZE), "DEVICE", "cpu", "EPIPOLAR.SAMPLESIZE", str(K)]
    cfg = rh.load_cfg("configs/epipolar/keypoint_h36m_zresidual_fixed.yaml", ov)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from modeling import registry
        import modeling.backbones.ProHG  # noqa: F401  (registers the hourglass names)

        net = registry.BACKBONES[cfg.BACKBONE.BODY](cfg)
    net.load_state_dict(deterministic_state_dict(net.state_dict()))
    net.eval()
    g = torch.Generator().manual_seed(sum(map(ord, c["name"])))
    img = torch.randn(N, 3, SIZE, SIZE, generator=g)
    scales = calm(net, img)                                                  # (in place: state_dict() tensors alias the parameters)
    src = torch.arange(N).roll(-1)
    P_ref, P_src = syn.make_pairs(1, N, SIZE, seed=31, jitter=(0.03, 2.0))
    assert torch.equal(P_src, P_ref[src])
    cam = np.concatenate([a.reshape(N, -1) for a in orc.camera_algebra(P_ref, P_src)], 1).astype(np.float32)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        own = net(img)[0]                                                    # per-stack maps of every view (no fusion)
        other_features = [f[src] for f in own]
        out = net(img, other_inputs=[other_features, P_src, None, P_ref, None, None, img[src]])
    features, heatmaps, locs, scos, corr_pos, depth, _, _ = out
    npf = lambda t: t.detach().numpy().astype(np.float32)
    data = dict(meta=np.array([SIZE, HS, K, J, N]), body=np.array(c["body"]), merge=np.array(c["merge"]), img=npf(img), src=src.numpy(),
                KRT=npf(P_ref), cam=cam, locs=npf(locs), scos=npf(scos), corr_pos=npf(corr_pos), depth=npf(depth),
                n_features=np.array(len(features)), n_heatmaps=np.array(len(heatmaps)), n_own=np.array(len(own)))
    data["feature_last"] = npf(features[-1])                                  # (one fused map is enough; every heat map is kept)
    data["feature_first_checksum"] = np.array([float(f.double().abs().sum()) for f in features])
    for i, h in enumerate(heatmaps):
        data["heatmap%d" % i] = npf(h)
    data["own_last"] = npf(own[-1])                                           # (the trunk alone, last stack: one map is enough)
    data["weight_scales"] = np.array(scales, np.float64)                     # (apply_weight_scales below re-applies them)
    print("%-12s %d parameters tensors, feature scale %.2f, heat-map scale %.3f" %
          (c["name"], len(net.state_dict()), float(features[-1].abs().max()), float(heatmaps[-1].abs().max())))
    return data


def main():
    torch.set_num_threads(4)
    only = set(sys.argv[1:])
    for c in CASES:
        if only and c["name"] not in only:
            continue
        path = os.path.join(HERE, "hourglass_%s.npz" % c["name"])
        np.savez_compressed(path, **run_case(c))
        print("  -> %s %.0f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
