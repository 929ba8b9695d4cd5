"""The stacked-hourglass callers of the layer (reference modeling/backbones/ProHG.py:120-316, registry names
epipolarHG / epipolarHG1 / epipolarHG11) on the GPU against outputs of the REAL reference (tests/golden/hourglass_*.npz, made by
tests/golden/make_hourglass_golden.py): same state_dict keys, the per-stack heat maps, the fused feature map, detections, corr_pos
and the attention of the last fusion point -- one stack with late fusion, three stacks with early fusion, MERGE both."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, assert_corr_pos

pytestmark = pytest.mark.gpu

sys.path.insert(0, GOLDEN_DIR)


def _build(d):
    from epipolar_transformers_amd import backbones, default_cfg
    from make_hourglass_golden import apply_weight_scales
    from model_weights import deterministic_state_dict

    size, hs, k, j, n = [int(v) for v in d["meta"]]
    cfg = default_cfg()
    cfg.merge_from_list(["BACKBONE.BODY", str(d["body"]), "BACKBONE.PRETRAINED", False, "EPIPOLAR.MERGE", str(d["merge"]),
                         "KEYPOINT.HEATMAP_SIZE", (hs, hs), "KEYPOINT.NUM_PTS", j, "KEYPOINT.SIGMA", 2.0, "KEYPOINT.NFEATS", 256,
                         "DATASETS.IMAGE_SIZE", (size, size), "EPIPOLAR.SAMPLESIZE", k,
                         # configs/epipolar/keypoint_h36m_zresidual_fixed.yaml:27-35, the fixture's base
                         "EPIPOLAR.ATTENTION", "avg", "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True,
                         "EPIPOLAR.USE_CORRECT_NORMALIZE", True, "EPIPOLAR.SHARE_WEIGHTS", True])
    net = backbones.build_backbone(cfg)
    sd = deterministic_state_dict(net.state_dict())
    net.load_state_dict(apply_weight_scales(sd, d["weight_scales"], net.nStack))
    return net.cuda().eval(), cfg


@pytest.mark.parametrize("case", ["hg1_late", "hg3_early", "hg11_both"])
def test_hourglass_caller_vs_reference(case):
    d = np.load(os.path.join(GOLDEN_DIR, "hourglass_%s.npz" % case))
    net, cfg = _build(d)
    size, hs, k, j, n = [int(v) for v in d["meta"]]
    img = torch.from_numpy(d["img"]).cuda()
    src = torch.from_numpy(d["src"]).cuda()
    KRT = torch.from_numpy(d["KRT"])
    cam = torch.from_numpy(d["cam"]).cuda()
    if hasattr(net, "epipolar_sampler"):
        net.epipolar_sampler._cams.get = lambda *a, **kw: cam            # the algebra the reference computed for the fixture
    with torch.no_grad():
        own = net(img)[0]
        assert len(own) == int(d["n_own"])
        scale = max(1.0, float(np.abs(d["own_last"]).max()))
        assert np.abs(own[-1].cpu().numpy() - d["own_last"]).max() <= 2e-4 * scale          # the trunk alone (MIOpen fp32 vs CPU)
        other = [f[src] for f in own]
        features, heatmaps, locs, scos, corr_pos, depth, _, warped = net(
            img, other_inputs=[other, KRT[d["src"]], None, KRT, None, None, img[src]])
    assert warped is None and len(features) == int(d["n_features"]) and len(heatmaps) == int(d["n_heatmaps"])
    for i, h in enumerate(heatmaps):
        want = d["heatmap%d" % i]
        assert np.abs(h.cpu().numpy() - want).max() <= 5e-4 * max(1.0, float(np.abs(want).max())), (case, i)
    want = d["feature_last"]
    assert np.abs(features[-1].cpu().numpy() - want).max() <= 5e-4 * max(1.0, float(np.abs(want).max()))
    for f, chk in zip(features, d["feature_first_checksum"]):
        assert abs(float(f.double().abs().sum()) - chk) <= 1e-3 * chk
    assert np.abs(depth.cpu().numpy() - d["depth"]).max() <= 1e-4          # the attention of the last fusion point (inputs differ by the trunk's rounding)
    from epipolar_transformers_amd import ops
    locs_all = ops.sample_locs(net.epipolar_sampler.layer_spec(), cam).cpu().numpy()
    assert_corr_pos(locs_all, corr_pos.cpu().numpy(), d["corr_pos"], depth.cpu().numpy(), True, tie=2e-4, max_frac=5e-2)
    assert np.abs(scos.cpu().numpy() - d["scos"]).max() <= 5e-4 * max(1.0, float(np.abs(d["scos"]).max()))
    # detections: the arg-max cell may flip between near-equal heat-map values; where the peak cell agrees the sub-pixel location does too
    dl = np.abs(locs.cpu().numpy() - d["locs"]).max(-1)
    assert (dl <= 0.05).mean() >= 0.9, dl


def test_hourglass_registry_names_and_state_dict_keys():
    """every registry name of ProHG.py:319-395 that does not need the Meta layer, and the parameter names a reference checkpoint
    carries (the fixture test loads weights rebuilt from those names on both sides)"""
    from epipolar_transformers_amd import backbones, default_cfg

    for name, stacks in (("HG", 3), ("HG1", 1), ("HG11", 1), ("epipolarHG", 3), ("epipolarHG1", 1), ("epipolarHG11", 1),
                         ("simplemultiviewHG", 3), ("simplemultiviewHG1", 1), ("simplemultiviewHG11", 1)):
        assert name in backbones.BACKBONES
    cfg = default_cfg()
    cfg.merge_from_list(["BACKBONE.BODY", "epipolarHG", "BACKBONE.PRETRAINED", False, "KEYPOINT.HEATMAP_SIZE", (16, 16),
                         "DATASETS.IMAGE_SIZE", (64, 64), "KEYPOINT.NUM_PTS", 7, "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True])
    net = backbones.build_backbone(cfg)
    keys = set(net.state_dict())
    for k in ("conv.0.weight", "conv.7.running_var", "ress.0.branch.2.bias", "ress.3.conv_C.2.weight", "features.2.0.mid.mid.mid.0.conv_B.2.weight",
              "features.0.0.down.1.conv_A.0.weight", "features.1.2.weight", "features.1.3.running_mean", "tmpOuts.2.bias", "trsfeas.1.weight",
              "trstmps.0.weight", "epipolar_sampler.z.weight", "epipolar_sampler.bn.running_var"):
        assert k in keys, k
    assert len(keys) == 833                                                  # (what the reference's epipolarHG reports)
    cfg.merge_from_list(["BACKBONE.BODY", "metaHG"])                         # registered like the reference's; the Meta layer is not built
    with pytest.raises(NotImplementedError):
        backbones.build_backbone(cfg)
