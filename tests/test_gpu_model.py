"""Model level on the GPU (SURVEY.md 8f rows N1 / N3 and EPIPOLAR.MULTITEST): trunk once per view, GPU lifting."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(**over):
    from epipolar_transformers_amd import default_cfg

    size, hs = 64, 16
    cfg = default_cfg()
    cfg.merge_from_list(["BACKBONE.BODY", "epipolarposeR-18", "BACKBONE.PRETRAINED", False, "DATASETS.TASK", "multiview_keypoint",
                         "KEYPOINT.HEATMAP_SIZE", (hs, hs), "KEYPOINT.NUM_PTS", 17, "KEYPOINT.SIGMA", 2.0,
                         "DATASETS.IMAGE_SIZE", (size, size), "EPIPOLAR.MERGE", "late", "EPIPOLAR.ATTENTION", "avg",
                         "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True, "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                         "EPIPOLAR.SAMPLESIZE", 16, "VIS.MULTIVIEW", True])
    for k, v in over.items():
        cfg.merge_from_list([k, v])
    return cfg, size, hs


def _model(cfg):
    from epipolar_transformers_amd.model import MultiViewPoseModel

    torch.manual_seed(11)
    m = MultiViewPoseModel(cfg).cuda().eval()
    with torch.no_grad():
        m.reference.epipolar_sampler.bn.weight.normal_(1, 0.1)
        m.reference.epipolar_sampler.bn.bias.normal_(0, 0.1)
    return m


def test_trunk_once_per_view_equals_the_two_pass_reference_form():
    """N1: forward_views (trunk once per view) gives exactly what the reference's two passes give in eval mode:
    backbone(other_img) for the source features, then reference(img, [other_features, ...]) (model.py:241-247)."""
    from epipolar_transformers_amd import synthetic as syn
    from epipolar_transformers_amd.model import ring_sources

    cfg, size, hs = _cfg()
    m = _model(cfg)
    frames, V = 2, 4
    P = torch.from_numpy(syn.ring_cameras(V, size)).float().repeat(frames, 1, 1)            # frame-major (F*V,3,4)
    img = torch.randn(frames * V, 3, size, size, device="cuda")
    src = ring_sources(frames, V, "cuda")
    with torch.no_grad():
        one = m.forward_views(img, P, src)
        other_features = m.backbone(img[src])[0]
        two = m.reference(img, [other_features, P[src.cpu()], None, P, None, None, None])
    # (MIOpen may pick another convolution algorithm for another batch: the trunk is equal to rounding, not to the bit)
    assert (one[0] - two[0]).abs().max().item() <= 1e-5 * max(1.0, two[0].abs().max().item())     # pre-fusion features
    assert (one[1][0] - two[1][0]).abs().max().item() <= 1e-4            # heat maps
    assert (one[2] - two[2]).abs().max().item() <= 1e-2                  # detections, image pixels
    assert (one[5] - two[5]).abs().max().item() <= 1e-5 and (one[4] != two[4]).any(-1).float().mean().item() <= 1e-2
    # dict form = Modelbuilder.forward: both spellings of the batch, plus lifting and MPJPE on the device
    gt = torch.randn(frames, V, 17, 3, device="cuda").double()
    _, metrics, out = m({"img": img, "KRT": P, "other_index": src, "points-3d": gt, "num_views": V}, is_train=False)
    _, metrics2, out2 = m({"img": img, "KRT": P, "other_img": img[src], "other_KRT": P[src.cpu()], "num_views": V}, is_train=False)
    assert out["points-3d"].is_cuda and tuple(out["points-3d"].shape) == (frames, 17, 3) and "MPJPE" in metrics
    assert (out["batch_locs"] - out2["batch_locs"]).abs().max().item() <= 1e-2
    # training: the reference's loss entry, gradients reach the trunk and the layer
    m.train()
    loss, _ = m({"img": img, "KRT": P, "other_index": src, "heatmap": torch.rand(frames * V, 17, hs, hs, device="cuda"),
                 "visibility": torch.ones(frames * V, 17, 1, device="cuda"), "num_views": V}, is_train=True)
    loss["loss"].backward()
    assert m.reference.conv1.weight.grad is not None and m.reference.epipolar_sampler.z.weight.grad is not None


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_reduced_precision_trunk_option(dt):
    """EPIPOLAR_AMD.TRUNK_DTYPE: the stock trunk under autocast, the fused layer in fp32 on the map it leaves (channels_last, no
    copy); detections stay within a fraction of a heat-map cell of the fp32 trunk's.  An option: the default is the reference's fp32."""
    from epipolar_transformers_amd import synthetic as syn
    from epipolar_transformers_amd.model import ring_sources

    cfg, size, hs = _cfg()
    m = _model(cfg)
    frames, V = 2, 4
    P = torch.from_numpy(syn.ring_cameras(V, size)).float().repeat(frames, 1, 1)
    img = torch.randn(frames * V, 3, size, size, device="cuda")
    src = ring_sources(frames, V, "cuda")
    with torch.no_grad():
        full = m.forward_views(img, P, src)
        cfg.merge_from_list(["EPIPOLAR_AMD.TRUNK_DTYPE", dt])
        low = m.forward_views(img, P, src)
    assert low[0].dtype == torch.float32 and low[1][0].dtype == torch.float32
    scale = full[0].abs().max().item()
    assert 0 < (low[0] - full[0]).abs().max().item() <= (0.05 if dt == "bf16" else 0.01) * scale
    assert torch.isfinite(low[2]).all()


@pytest.mark.parametrize("share", [True, False])
def test_multitest_picks_the_best_source_per_joint(share):
    """EPIPOLAR.MULTITEST (model.py:213-239): every other view as the source (features from `self.backbone`: the same
    network only with SHARE_WEIGHTS); per joint the highest score wins."""
    from epipolar_transformers_amd import synthetic as syn

    cfg, size, hs = _cfg(**{"EPIPOLAR.MULTITEST": True, "EPIPOLAR.SHARE_WEIGHTS": share})
    m = _model(cfg)
    frames, V = 1, 4
    P = torch.from_numpy(syn.ring_cameras(V, size)).float()
    img = torch.randn(V, 3, size, size, device="cuda")
    with torch.no_grad():
        m.reference.final_layer.weight.mul_(30.0)        # (a random-initialised head is nearly flat: give it contrast)
        locs, scos = m.forward_multitest(img, P, V)
        all_s = []
        for shift in range(1, V):                                           # the reference's loop over other views
            idx = torch.arange(V, device="cuda").roll(-shift)
            of = m.backbone(img[idx])[0]
            all_s.append(m.reference(img, [of, P[idx.cpu()], None, P, None, None, None])[3])
        best, _ = torch.stack(all_s).max(0)
        # scores against the reference's two-pass loop (MIOpen's fp32 trunk differs by ~1e-4 relative between two
        # differently composed batches; the wrong network for the sources would differ by O(1))
        assert (best - scos).abs().max().item() <= 5e-3 * max(1.0, best.abs().max().item())
        # Locations: the arg-max of a heat map can flip on such differences, so the per-joint selection is checked
        # with both trunks and the 1x1 head frozen: single-source passes and the one-launch form then see identical
        # features (the fused layer is batch-invariant bit for bit) and must agree exactly.
        feat = m.reference.trunk(img)
        m.reference.trunk = lambda x: feat
        if m.backbone is not m.reference:
            feat_b = m.backbone(img)[0]
            m.backbone.forward = lambda x, *a, **k: (feat_b,)
        fl = m.reference.final_layer
        w64, b64 = fl.weight.double().flatten(1), fl.bias.double()

        class Head64(torch.nn.Module):                   # (MIOpen also picks its algorithm for the head by batch size)
            def forward(self, t):
                return (torch.einsum("nchw,jc->njhw", t.double(), w64) + b64[None, :, None, None]).float()

        m.reference.final_layer = Head64()
        locs, scos = m.forward_multitest(img, P, V)
        own_l, own_s = [], []
        for shift in range(1, V):
            r = m.forward_views(img, P, torch.arange(V, device="cuda").roll(-shift))
            own_l.append(r[2])
            own_s.append(r[3])
        own_best, which = torch.stack(own_s).max(0)
        want = torch.gather(torch.stack(own_l), 0, which[None, ..., None].expand(-1, -1, -1, 2)).squeeze(0)
    assert (own_best - scos).abs().max().item() <= 1e-5 * max(1.0, own_best.abs().max().item())
    assert (want - locs).abs().max().item() <= 1e-3


def test_lifting_on_device_matches_the_reference_linear_triangulation():
    """N3: `lift` (batched float64 SVD-DLT on the GPU) against a per-joint numpy restatement of the reference's
    find3d (vision/multi_camera_system.py:199-225) with the confidence rule of triangulation.py:427-435."""
    from epipolar_transformers_amd import synthetic as syn
    from epipolar_transformers_amd.triangulate import triangulate_dlt

    V, J, F_ = 4, 17, 3
    P = torch.from_numpy(syn.ring_cameras(V, 256))                       # (V,3,4) float64
    g = torch.Generator().manual_seed(2)
    X = torch.tensor([0.0, 0.0, 900.0], dtype=torch.float64) + torch.randn(F_, J, 3, generator=g, dtype=torch.float64) * 300
    Xh = torch.cat([X, torch.ones(F_, J, 1, dtype=torch.float64)], -1)
    uvw = torch.einsum("vij,fkj->fvki", P, Xh)
    uv = uvw[..., :2] / uvw[..., 2:3] + torch.randn(F_, V, J, 2, generator=g, dtype=torch.float64) * 0.5   # noisy detections
    conf = torch.rand(F_, V, J, generator=g)
    conf[0, :, 0] = torch.tensor([0.9, 0.01, 0.8, 0.02])                # one joint with only two confident views
    conf[1, :, 1] = 0.01                                                 # and one below the threshold everywhere
    got = triangulate_dlt(uv.cuda(), P[None].expand(F_, -1, -1, -1).cuda(), conf.cuda(), conf_thres=0.05).cpu().numpy()
    Pn, uvn, cn = P.numpy(), uv.numpy(), conf.double().numpy()
    for f in range(F_):
        for k in range(J):
            thr = 0.05
            while True:
                sel = np.where(cn[f, :, k] > thr)[0]
                if thr < -1 or len(sel) > 1:
                    break
                thr -= 0.05
            A = []
            for v in sel:
                A.append(uvn[f, v, k, 0] * Pn[v, 2] - Pn[v, 0])
                A.append(uvn[f, v, k, 1] * Pn[v, 2] - Pn[v, 1])
            _, _, vt = np.linalg.svd(np.array(A))
            want = vt[-1, :3] / vt[-1, 3]
            assert np.abs(got[f, k] - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (f, k)


def test_model_equals_the_reference_modelbuilder_fixture():
    """Row N1 pinned to reference CODE: tests/golden/model_r18.npz holds what the REAL reference `Modelbuilder`
    (modeling/model.py:160-302, two backbone passes per pair, CPU) produced for 2 frames x 4 views of epipolarposeR-18 --
    heat maps, detections, corr_pos, depth in eval mode; the loss and three gradient tensors of one training step.
    `MultiViewPoseModel` with the same (name-derived) weights must reproduce them on the GPU, with the trunk run once per
    view (`other_index`) and in the reference's own two-pass spelling (`other_img` / `other_KRT`)."""
    import os
    import sys

    from conftest import GOLDEN_DIR, assert_corr_pos
    from epipolar_transformers_amd import ops
    from epipolar_transformers_amd.model import MultiViewPoseModel

    sys.path.insert(0, GOLDEN_DIR)
    from model_weights import deterministic_state_dict

    d = np.load(os.path.join(GOLDEN_DIR, "model_r18.npz"))
    frames, V, size, hs, K, J = [int(v) for v in d["meta"]]
    cfg, _, _ = _cfg(**{"VIS.MULTIVIEW": False, "EPIPOLAR.SAMPLESIZE": K, "KEYPOINT.NUM_PTS": J,
                        "EPIPOLAR.SHARE_WEIGHTS": True})        # (keypoint_h36m_zresidual_fixed.yaml, as the fixture)
    m = MultiViewPoseModel(cfg)
    m.reference.load_state_dict(deterministic_state_dict(m.reference.state_dict()))
    m = m.cuda().eval()
    cam = torch.from_numpy(d["cam"]).cuda()
    m.reference.epipolar_sampler._cams.get = lambda *a, **k: cam       # the algebra the reference computed for the fixture
    img = torch.from_numpy(d["img"]).cuda()
    src = torch.from_numpy(d["src"]).cuda()
    KRT = torch.from_numpy(d["KRT"])
    batches = {"once per view": {"img": img, "KRT": KRT, "other_index": src, "num_views": V},
               "two passes": {"img": img, "KRT": KRT, "other_img": img[src], "other_KRT": KRT[src.cpu()], "num_views": V}}
    hscale = float(np.abs(d["heat_eval"]).max())
    for name, batch in batches.items():
        with torch.no_grad():
            _, _, out = m(dict(batch), is_train=False)
        heat = out["heatmaps"].cpu().numpy()
        assert np.abs(heat - d["heat_eval"]).max() <= 2e-4 * hscale, (name, np.abs(heat - d["heat_eval"]).max(), hscale)
        assert np.abs(out["depth"].cpu().numpy() - d["depth"]).max() <= 2e-5, name
        locs = ops.sample_locs(m.reference.epipolar_sampler.layer_spec(), cam).cpu().numpy()
        assert_corr_pos(locs, out["corr_pos"].cpu().numpy(), d["corr_pos"], out["depth"].cpu().numpy(), True, tie=5e-6)
        assert np.abs(out["batch_scos"].cpu().numpy() - d["scores_eval"]).max() <= 2e-4 * hscale, name
        # detections: equal to a hundredth of an image pixel, except where the heat map's arg-max has a proven tie
        got, want = out["batch_locs"].cpu().numpy(), d["locs_eval"]
        far = np.abs(got - want).max(-1) > 1e-2
        for n, j in zip(*np.nonzero(far)):
            hm = heat[n, j]
            top = np.sort(hm.ravel())[-2:]
            assert top[1] - top[0] <= 4e-4 * hscale, (name, n, j, got[n, j], want[n, j], top)
        assert far.mean() <= 0.05, name
    # one training step: loss and gradients (BN batch statistics: the two reference passes see the same eight images)
    m.train()
    for name, batch in batches.items():
        m.zero_grad()
        b = dict(batch, heatmap=torch.from_numpy(d["target"]).cuda(), visibility=torch.from_numpy(d["vis"]).cuda())
        loss, _ = m(b, is_train=True)
        assert abs(loss["loss"].item() - float(d["loss"][0])) <= 2e-5 * float(d["loss"][0]), (name, loss["loss"].item(), d["loss"])
        loss["loss"].backward()
        net = m.reference
        for got, want in ((net.conv1.weight.grad, d["grad_conv1"]), (net.epipolar_sampler.z.weight.grad[:16], d["grad_z_rows"]),
                          (net.final_layer.weight.grad, d["grad_final"])):
            # (a whole-network gradient, CPU vs GPU: rounding differences of ~1e-6 in the activations flip a few ReLU
            #  gates, each a discrete local change -- bound the difference in the mean (relative L2) and in the maximum)
            diff = got.cpu().numpy() - want
            scale = float(np.abs(want).max())
            assert np.linalg.norm(diff) <= 2e-2 * np.linalg.norm(want), (name, np.linalg.norm(diff), np.linalg.norm(want))
            assert np.abs(diff).max() <= 5e-2 * scale, (name, np.abs(diff).max(), scale)
        norms = {k: p.grad.norm().item() for k, p in net.named_parameters() if p.grad is not None}
        assert sorted(norms) == [str(k) for k in d["grad_keys"]]
        # (a bias in front of a batch norm in training mode -- conv biases, z.bias -- has a mathematically zero gradient:
        #  what is computed is rounding noise; norms are compared relative to the network's largest one)
        floor = 1e-3 * float(np.max(d["grad_norms"]))
        for k, w in zip(d["grad_keys"], d["grad_norms"]):
            assert abs(norms[str(k)] - float(w)) <= 2e-2 * max(float(w), floor), (name, str(k), norms[str(k)], float(w))
