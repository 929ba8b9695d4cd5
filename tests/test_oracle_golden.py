"""Pin the CPU oracle (oracle/epipolar_oracle.c) against the golden vectors that
were generated from the real reference (tests/golden/make_golden.py).

Tolerances: sample_locs bit-exact (measured), attn 1e-6,
out 5e-6 absolute, corr_pos exact, grads 1e-4 relative to the tensor scale.
"""
import numpy as np
import pytest
import torch

from conftest import assert_corr_pos, golden_cases, load_golden


def _spec(orc, d):
    m = d["dims"]
    return orc.LayerSpec(m["H"], m["W"], m["K"], downsample=float(d["downsample"]),
                         correct_normalize=m["correct"], softmax_scale=float(d["softmax_scale"]),
                         softmax_enabled=m["softmax"], align_corners=False)


@pytest.mark.parametrize("case", golden_cases())
def test_sample_locs_bit_exact(oracle_mod, case):
    d = load_golden(case)
    spec = _spec(oracle_mod, d)
    locs = oracle_mod.sample_locs(spec, None, None, cam=d["cam"])
    got = locs[:, :, d["rows"]]
    assert got.shape == d["sample_locs"].shape
    assert np.array_equal(got, d["sample_locs"]), "max|d|=%g" % np.abs(got - d["sample_locs"]).max()


@pytest.mark.parametrize("case", golden_cases())
def test_forward_matches_reference(oracle_mod, case):
    d = load_golden(case)
    spec = _spec(oracle_mod, d)
    r = oracle_mod.forward(spec, d["feat1"], d["feat2"], None, None, cam=d["cam"])
    attn = r["attn"][:, :, d["rows"]]
    # relative term only matters for the softmax-off case, where a masked
    # sample keeps its -1e10/K logit as a weight (epipolar.py:298,311)
    assert np.allclose(attn, d["attn"], rtol=2e-6, atol=1e-6)
    assert np.allclose(r["out"], d["out"], rtol=2e-6, atol=5e-6)
    # rows of attention sum to one when the softmax is on
    if d["dims"]["softmax"]:
        assert np.abs(r["attn"].sum(1) - 1.0).max() < 1e-5
    # corr_pos is an arg-max: it may differ from the reference's only at PROVEN ties of the oracle's own attention
    assert_corr_pos(r["sample_locs"], r["corr_pos"], d["corr_pos"], r["attn"], d["dims"]["correct"], max_frac=2e-3)


@pytest.mark.parametrize("case", golden_cases())
def test_zero_feature_pixel_gives_uniform_attention(oracle_mod, case):
    d = load_golden(case)
    if not (d["feat1"][0, :, 3, 5] == 0).all() or not d["dims"]["softmax"]:
        pytest.skip("case has no all-zero reference pixel")
    spec = _spec(oracle_mod, d)
    r = oracle_mod.forward(spec, d["feat1"], d["feat2"], None, None, cam=d["cam"])
    K = d["dims"]["K"]
    assert np.allclose(r["attn"][0, :, 3, 5], 1.0 / K, atol=1e-7)       # epipolar.py:298 (H3)


@pytest.mark.parametrize("case", golden_cases())
def test_epilogue_matches_reference(oracle_mod, case):
    d = load_golden(case)
    fin, fused = oracle_mod.epilogue(d["out"], d["feat1"], d["z_weight"], d["z_bias"], d["bn_weight"],
                                     d["bn_bias"], d["bn_running_mean"], d["bn_running_var"], training=False)
    assert np.abs(fin.numpy() - d["finalout_eval"]).max() <= 2e-6
    fin_t, _ = oracle_mod.epilogue(d["out"], d["feat1"], d["z_weight"], d["z_bias"], d["bn_weight"],
                                   d["bn_bias"], d["bn_running_mean"], d["bn_running_var"], training=True)
    assert np.abs(fin_t.numpy() - d["finalout_train"]).max() <= 2e-5


@pytest.mark.parametrize("case", golden_cases())
def test_backward_matches_reference_autograd(oracle_mod, case):
    d = load_golden(case)
    spec = _spec(oracle_mod, d)
    locs = oracle_mod.sample_locs(spec, None, None, cam=d["cam"])
    g1, g2 = oracle_mod.backward(spec, d["feat1"], d["feat2"], locs, d["grad_out"])
    for got, want in ((g1, d["grad_feat1"]), (g2, d["grad_feat2"])):
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 1e-4 * scale, (np.abs(got - want).max(), scale)


@pytest.mark.parametrize("case", golden_cases())
def test_torch_op_sequence_vs_reference(case):
    """oracle/torch_ref_path.py (the reference's executed op sequence, used as bench.py's `cpu_baseline`, implementation "reference-op-sequence"
    CPU baseline) reproduces the real reference's outputs on the fixtures."""
    import torch

    from oracle import torch_ref_path as trp

    d = load_golden(case)
    m = d["dims"]
    from oracle import oracle as orc
    orc.build()
    locs = orc.sample_locs(_spec(orc, d), None, None, cam=d["cam"])
    out, attn, corr = trp.forward(torch.from_numpy(d["feat1"]), torch.from_numpy(d["feat2"]), torch.from_numpy(locs),
                                  softmax_scale=float(d["softmax_scale"]), softmax_enabled=m["softmax"],
                                  correct_normalize=m["correct"])
    assert np.abs(out.numpy() - d["out"]).max() <= 5e-6 * max(1.0, float(np.abs(d["out"]).max()))
    assert np.abs(attn.numpy()[:, :, d["rows"]] - d["attn"]).max() <= 1e-6 * max(1.0, float(np.abs(d["attn"]).max()))
    assert_corr_pos(locs, corr.numpy(), d["corr_pos"], attn.numpy(), m["correct"], max_frac=1e-3)
