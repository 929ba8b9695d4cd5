"""The drop-in boundary above the operator, on CPU: registry names, constructor
surface, checkpoint-key compatibility with the reference, single-view forward
(`other_features=None`) equal to the reference's, and the launcher swap.
Tests that need the read-only reference tree are skipped where it is absent
(the GPU box)."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

from epipolar_transformers_amd import default_cfg
from epipolar_transformers_amd import backbones

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "modeling")), reason="reference tree not mounted")


def _cfg(body="epipolarposeR-18", size=64):
    cfg = default_cfg()
    cfg.merge_from_list(["BACKBONE.BODY", body, "BACKBONE.PRETRAINED", False, "KEYPOINT.HEATMAP_SIZE", (size // 4, size // 4),
                         "KEYPOINT.NUM_PTS", 17, "KEYPOINT.SIGMA", 2.0, "DATASETS.IMAGE_SIZE", (size, size),
                         "EPIPOLAR.MERGE", "late", "EPIPOLAR.ATTENTION", "avg", "EPIPOLAR.PARAMETERIZED", ("z",),
                         "EPIPOLAR.ZRESIDUAL", True, "EPIPOLAR.USE_CORRECT_NORMALIZE", True, "EPIPOLAR.SHARE_WEIGHTS", True])
    return cfg


def test_registry_names_and_uniqueness():
    for d in ("18", "34", "50", "101", "152"):
        assert "poseR-" + d in backbones.BACKBONES and "epipolarposeR-" + d in backbones.BACKBONES
    with pytest.raises(KeyError):
        backbones.BACKBONES.register("poseR-50", lambda cfg: None)


def test_reference_yaml_parses_unchanged():
    if not os.path.isdir(REF):
        pytest.skip("reference tree not mounted")
    import glob

    for path in sorted(glob.glob(os.path.join(REF, "configs", "epipolar", "*.yaml")) +
                       glob.glob(os.path.join(REF, "configs", "benchmark", "*.yaml"))):
        cfg = default_cfg()
        cfg.merge_from_file(path)
        assert cfg.BACKBONE.BODY
    cfg = default_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/epipolar/keypoint_h36m_zresidual_fixed.yaml"))
    assert cfg.EPIPOLAR.PARAMETERIZED == ("z",) and cfg.KEYPOINT.HEATMAP_SIZE == (64, 64)
    assert cfg.EPIPOLAR.SOFTMAXSCALE == 0.125 and cfg.EPIPOLAR.USE_CORRECT_NORMALIZE is True


def test_single_view_forward_shapes_cpu():
    cfg = _cfg()
    net = backbones.build_backbone(cfg).eval()
    assert hasattr(net, "epipolar_sampler") and set(k.split(".")[0] for k in net.epipolar_sampler.state_dict()) == {"z", "bn"}
    with torch.no_grad():
        feat, heat, locs, scos, corr, depth, sl, _ = net(torch.randn(2, 3, 64, 64))
    assert tuple(feat.shape) == (2, 256, 16, 16) and tuple(heat[0].shape) == (2, 17, 16, 16)
    assert tuple(locs.shape) == (2, 17, 2) and tuple(scos.shape) == (2, 17) and corr is None and depth is None


def test_trunk_dtype_knob_cpu():
    """EPIPOLAR_AMD.TRUNK_DTYPE: fp32 is the default and the reference's arithmetic; bf16 runs the stock convolutions under autocast
    and still hands the layer an fp32 map; anything else is refused."""
    cfg = _cfg()
    assert cfg.EPIPOLAR_AMD.TRUNK_DTYPE == "fp32"
    net = backbones.build_backbone(cfg).eval()
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        full = net.trunk(x)
        cfg.merge_from_list(["EPIPOLAR_AMD.TRUNK_DTYPE", "bf16"])
        low = net.trunk(x)
        cfg.merge_from_list(["EPIPOLAR_AMD.TRUNK_DTYPE", "int8"])
        with pytest.raises(ValueError):
            net.trunk(x)
    assert low.dtype == torch.float32 and low.shape == full.shape
    assert 0 < (low - full).abs().max().item() <= 0.1 * full.abs().max().item()


@needs_ref
def test_checkpoint_keys_and_single_view_parity_with_reference(tmp_path):
    from oracle import ref_harness as rh

    ov = ["FOLDER_NAME", str(tmp_path), "BACKBONE.BODY", "epipolarposeR-18", "BACKBONE.PRETRAINED", "False", "KEYPOINT.HEATMAP_SIZE", "(16, 16)",
          "KEYPOINT.NUM_PTS", "17", "KEYPOINT.SIGMA", "2.0", "DATASETS.IMAGE_SIZE", "(64, 64)", "DEVICE", "cpu",
          "KEYPOINT.NFEATS", "256"]
    rcfg = rh.load_cfg("configs/epipolar/keypoint_h36m_zresidual_fixed.yaml", ov)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from modeling import registry as ref_registry
        import modeling.backbones.resnet  # noqa: F401  (registers the names)

        torch.manual_seed(0)
        ref_net = ref_registry.BACKBONES["epipolarposeR-18"](rcfg).eval()
    ours = backbones.build_backbone(_cfg()).eval()
    ref_sd = ref_net.state_dict()
    assert sorted(ref_sd) == sorted(ours.state_dict())                   # released checkpoints load by name
    ours.load_state_dict(ref_sd)
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = ref_net(x)
        o = ours(x)
    assert torch.allclose(o[0], r[0], atol=1e-5) and torch.allclose(o[1][0], r[1][0], atol=1e-5)
    assert torch.allclose(o[2], r[2], atol=1e-3) and torch.allclose(o[3], r[3], atol=1e-6)   # peaks, scores


@needs_ref
def test_launcher_swaps_operator_into_reference_backbone(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    monkeypatch.chdir(tmp_path)                      # the reference logs into ./outs by default
    import main as launcher

    ref_cfg = launcher.install(REF)
    from epipolar_transformers_amd.epipolar import Epipolar
    import modeling.backbones.resnet as ref_resnet

    assert ref_resnet.Epipolar is Epipolar
    ref_cfg.defrost() if hasattr(ref_cfg, "defrost") else None
    ref_cfg.merge_from_file(os.path.join(REF, "configs/epipolar/keypoint_h36m_zresidual_fixed.yaml"))
    ref_cfg.merge_from_list(["BACKBONE.PRETRAINED", False, "BACKBONE.BODY", "epipolarposeR-18"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = ref_resnet.get_pose_net(ref_cfg)
    assert isinstance(net.epipolar_sampler, Epipolar)
    assert sorted(k for k in net.state_dict() if k.startswith("epipolar_sampler")) == sorted(
        "epipolar_sampler." + k for k in ("z.weight", "z.bias", "bn.weight", "bn.bias", "bn.running_mean",
                                          "bn.running_var", "bn.num_batches_tracked"))


def test_backbone_trunk_equals_the_reference_modelbuilder_fixture():
    """Row N1 on the CPU: the pre-fusion features of our `epipolarposeR-18` (single-view call, resnet.py:381-383,406)
    equal what the REAL reference `Modelbuilder` produced for the same images and the same (name-derived) weights --
    tests/golden/model_r18.npz, made by tests/golden/make_model_golden.py.  (The multi-view outputs of the fixture need
    the GPU layer: tests/test_gpu_model.py.)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from model_weights import deterministic_state_dict

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_r18.npz"))
    frames, V, size, hs, K, J = [int(v) for v in d["meta"]]
    cfg = _cfg(size=size)
    cfg.merge_from_list(["EPIPOLAR.SAMPLESIZE", K, "KEYPOINT.NUM_PTS", J])
    net = backbones.build_backbone(cfg).eval()
    net.load_state_dict(deterministic_state_dict(net.state_dict()))
    with torch.no_grad():
        feat = net(torch.from_numpy(d["img"]))[0]
    assert abs(feat.abs().max().item() - float(d["feat_norm"][0])) <= 1e-4 * float(d["feat_norm"][0])
    assert abs(feat.norm().item() - float(d["feat_norm"][1])) <= 1e-5 * float(d["feat_norm"][1])
    assert np.abs(feat[:, :8].numpy() - d["feat_slice"]).max() <= 2e-5
