"""The operator's non-headline branches on the GPU against outputs of the REAL reference
(tests/golden/modes/*.npz, made by tests/golden/make_golden_modes.py): theta/phi/g bottleneck + POOLING
(configs/epipolar/keypoint_h36m_param.yaml), ATTENTION max, cosine similarity, PRIOR (+PRIORMUL), FIND_CORR rgb."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, assert_corr_pos

pytestmark = pytest.mark.gpu

MODES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "modes", "*.npz")))


def _call_kwargs(d, name, dev, depth_grad=False):
    """camera ids (+ rgb references, + the externally supplied depth weights) of a mode fixture"""
    kw = dict(camera=torch.from_numpy(d["camera"]), other_camera=torch.from_numpy(d["other_camera"]))
    if "rgb" in name:
        kw.update(ref1=dev("rgb1"), ref2=dev("rgb2"))
    if "depth_given" in d.files:
        kw.update(depth=dev("depth_given").requires_grad_(depth_grad))
    return kw


def _module(d):
    from epipolar_transformers_amd import default_cfg
    from epipolar_transformers_amd.epipolar import Epipolar

    H, C, K, N, image = [int(v) for v in d["meta"]]
    cfg = default_cfg()
    cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (H, H), "KEYPOINT.NFEATS", C, "EPIPOLAR.SAMPLESIZE", K,
                         "DATASETS.IMAGE_SIZE", (image, image), "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                         # the fixtures start from configs/epipolar/keypoint_h36m_zresidual_fixed.yaml (:27-35) ...
                         "EPIPOLAR.ATTENTION", "avg", "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True,
                         "EPIPOLAR.MERGE", "late", "EPIPOLAR.SHARE_WEIGHTS", True] +
                        [str(v) for v in d["overrides"]])     # ... plus the case's own overrides
    mod = Epipolar(cfg=cfg).cuda().eval()
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd.")}
    assert sorted(mod.state_dict()) == sorted(sd), (sorted(mod.state_dict()), sorted(sd))   # same keys as the reference module
    mod.load_state_dict(sd)
    for k in d.files:
        if k.startswith("prior."):
            _, i, j = k.split(".")
            with torch.no_grad():
                mod.prior[(int(i), int(j))] = torch.nn.Parameter(torch.from_numpy(d[k]).cuda())
    cam = torch.from_numpy(d["cam"]).cuda()
    mod._cams.get = lambda *a, **k: cam                        # the algebra the reference computed for the fixture
    return mod


@pytest.mark.parametrize("name", MODES)
def test_mode_vs_reference(name):
    d = np.load(os.path.join(GOLDEN_DIR, "modes", name + ".npz"))
    mod = _module(d)
    assert "depth_given" in d.files or not mod._fused_mode(torch.zeros(1) if "rgb" in name else None, None)
    dev = lambda k: torch.from_numpy(d[k]).cuda()
    f1, f2 = dev("feat1").requires_grad_(True), dev("feat2").requires_grad_(True)
    kw = _call_kwargs(d, name, dev, depth_grad=True)
    fin, corr, depth, _ = mod(f1, f2, torch.from_numpy(d["P1"]), torch.from_numpy(d["P2"]), **kw)
    assert tuple(fin.shape) == d["finalout"].shape and tuple(depth.shape) == d["depth"].shape
    from epipolar_transformers_amd import ops

    # corr_pos may differ from the reference's only where the arg-max has a PROVEN tie in our own similarity (`depth`)
    locs = ops.sample_locs(mod.layer_spec(), torch.from_numpy(d["cam"]).cuda()).cpu().numpy()      # (K,N,H,W,2)
    got_corr, depth_np = corr.cpu().numpy(), depth.detach().cpu().numpy()
    ties = assert_corr_pos(locs, got_corr, d["corr_pos"], depth_np, True, tie=2e-6, max_frac=2e-2)   # (N,H,W) bool
    assert np.abs(depth_np - d["depth"]).max() <= 1e-5 * max(1.0, float(np.abs(d["depth"]).max()))
    scale = max(1.0, float(np.abs(d["finalout"]).max()))
    err = np.abs(fin.detach().cpu().numpy() - d["finalout"])
    is_max = "attention_max" in name or "depth_given_max" in name
    if is_max:
        # ATTENTION max GATHERS the arg-max sample: at a proven tie the other (equally good) sample's features come
        # out -- those pixels, and only those, are exempt; every other pixel must agree
        assert (err.max(1) <= 1e-4 * scale)[~ties].all()
    else:
        assert err.max() <= 1e-4 * scale
    (fin * dev("grad_out")).sum().backward()
    # gradient entries a tie pixel touches (ATTENTION max only): its own d(feat1) vector and d(feat2) at the taps of
    # the two tied samples
    N, _, H, W = d["feat1"].shape
    touched2 = np.zeros((N, H, W), dtype=bool)
    if is_max and ties.any():
        for pos in (got_corr, d["corr_pos"]):                                   # de-normalised (x, y), multiview.py:50-57
            for n, h, w in zip(*np.nonzero(ties)):
                gx = ((pos[n, h, w, 0] * 2.0 / (W - 1)) * W - 1) / 2           # tap space of grid_sample (align_corners=False)
                gy = ((pos[n, h, w, 1] * 2.0 / (H - 1)) * H - 1) / 2
                for yy in (int(np.floor(gy)) - 1, int(np.floor(gy)), int(np.floor(gy)) + 1, int(np.floor(gy)) + 2):
                    for xx in (int(np.floor(gx)) - 1, int(np.floor(gx)), int(np.floor(gx)) + 1, int(np.floor(gx)) + 2):
                        if 0 <= yy < H and 0 <= xx < W:
                            touched2[n, yy, xx] = True                          # (one pixel of slack around the 2x2 taps)
    for got, want, exempt in ((f1.grad, d["grad_feat1"], ties if is_max else None),
                              (f2.grad, d["grad_feat2"], touched2 if is_max else None)):
        got = np.zeros_like(want) if got is None else got.cpu().numpy()
        gerr = np.abs(got - want).max(1)                                        # (N,H,W)
        ok = gerr <= 1e-4 * max(float(np.abs(want).max()), 1e-6)
        assert ok.all() if exempt is None else ok[~exempt].all()
        if exempt is not None:
            assert exempt.mean() <= 0.1
    if "depth_given" in d.files:          # the supplied weights' own gradient (ATTENTION max: none flows, exactly zero)
        gd = kw["depth"].grad
        gd = np.zeros_like(d["grad_depth"]) if gd is None else gd.cpu().numpy()
        assert np.abs(gd - d["grad_depth"]).max() <= 1e-4 * max(float(np.abs(d["grad_depth"]).max()), 1e-6)
    _check_prior_grads(mod, d)


def _check_prior_grads(mod, d):
    """the prior tables' OWN gradient against the reference's autograd (epipolar.py:288-289, 300-301, 308-309): priorgrad.<i>.<j>
    of the fixture; a pair no sample of the batch uses gets no gradient there (zeros)"""
    keys = [k for k in d.files if k.startswith("priorgrad.")]
    if not keys:
        return
    scale = max(max(float(np.abs(d[k]).max()) for k in keys), 1e-6)
    assert scale > 1e-3, "the fixture was meant to carry a gradient for the prior tables"
    for k in keys:
        _, i, j = k.split(".")
        g = mod.prior[(int(i), int(j))].grad
        g = np.zeros_like(d[k]) if g is None else g.cpu().numpy()
        assert np.abs(g - d[k]).max() <= 1e-4 * scale, (k, float(np.abs(g - d[k]).max()), scale)


@pytest.mark.parametrize("name", MODES)
def test_torch_restatement_vs_reference(name):
    """`Epipolar._attend_general_chunk` -- the torch restatement of epipolar.py:131-247 / 272-321 that the "vs the restatement"
    tests of the general kernels lean on, and the route of shapes those kernels do not take -- pinned to the same outputs of
    the real reference as the HIP route above (EPIPOLAR_AMD.GENERAL_KERNEL False forces it)."""
    import warnings

    from epipolar_transformers_amd.epipolar import EpipolarSlowPathWarning

    d = np.load(os.path.join(GOLDEN_DIR, "modes", name + ".npz"))
    mod = _module(d)
    mod.cfg.defrost() if hasattr(mod.cfg, "defrost") else None
    mod.cfg.merge_from_list(["EPIPOLAR_AMD.GENERAL_KERNEL", False])
    dev = lambda k: torch.from_numpy(d[k]).cuda()
    f1, f2 = dev("feat1").requires_grad_(True), dev("feat2").requires_grad_(True)
    assert not mod._general_kernel_applies(f1, f2, None, None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", EpipolarSlowPathWarning)
        kw = _call_kwargs(d, name, dev, depth_grad=True)
        fin, corr, depth, _ = mod(f1, f2, torch.from_numpy(d["P1"]), torch.from_numpy(d["P2"]), **kw)
    assert np.abs(depth.detach().cpu().numpy() - d["depth"]).max() <= 1e-5 * max(1.0, float(np.abs(d["depth"]).max()))
    is_max = "attention_max" in name or "depth_given_max" in name
    scale = max(1.0, float(np.abs(d["finalout"]).max()))
    err = np.abs(fin.detach().cpu().numpy() - d["finalout"]).max(1)
    from epipolar_transformers_amd import ops
    locs = ops.sample_locs(mod.layer_spec(), torch.from_numpy(d["cam"]).cuda()).cpu().numpy()
    ties = assert_corr_pos(locs, corr.cpu().numpy(), d["corr_pos"], depth.detach().cpu().numpy(), True, tie=2e-6, max_frac=2e-2)
    assert (err <= 1e-4 * scale)[~ties].all() if is_max else err.max() <= 1e-4 * scale
    (fin * dev("grad_out")).sum().backward()
    if not is_max:        # (ATTENTION max: the tie bookkeeping of the gradients is test_mode_vs_reference's)
        for got, want in ((f1.grad, d["grad_feat1"]), (f2.grad, d["grad_feat2"])):
            got = np.zeros_like(want) if got is None else got.cpu().numpy()
            assert np.abs(got - want).max() <= 1e-4 * max(float(np.abs(want).max()), 1e-6)
    if "depth_given" in d.files:
        gd = kw["depth"].grad
        gd = np.zeros_like(d["grad_depth"]) if gd is None else gd.cpu().numpy()
        assert np.abs(gd - d["grad_depth"]).max() <= 1e-4 * max(float(np.abs(d["grad_depth"]).max()), 1e-6)
    _check_prior_grads(mod, d)


def test_param_yaml_runs_through_the_backbone():
    """`configs/epipolar/keypoint_h36m_param.yaml` semantics through epipolarposeR-50 (the one shipped epipolar YAML
    that used to raise NotImplementedError)."""
    from epipolar_transformers_amd import backbones, default_cfg, synthetic as syn

    size, hs = 64, 16
    cfg = default_cfg()
    cfg.merge_from_list(["BACKBONE.BODY", "epipolarposeR-50", "BACKBONE.PRETRAINED", False,
                         "KEYPOINT.HEATMAP_SIZE", (hs, hs), "KEYPOINT.NUM_PTS", 20, "KEYPOINT.SIGMA", 2.0,
                         "DATASETS.IMAGE_SIZE", (size, size), "EPIPOLAR.MERGE", "late", "EPIPOLAR.ATTENTION", "avg",
                         "EPIPOLAR.PARAMETERIZED", ("z", "theta", "phi", "g"), "EPIPOLAR.POOLING", True,
                         "EPIPOLAR.BOTTLENECK", 2, "EPIPOLAR.ZRESIDUAL", False, "EPIPOLAR.SAMPLESIZE", 16])
    net = backbones.build_backbone(cfg).cuda().eval()
    assert {"epipolar_sampler.theta.weight", "epipolar_sampler.phi.weight", "epipolar_sampler.g.weight",
            "epipolar_sampler.z.weight"} <= set(net.state_dict())
    assert tuple(net.epipolar_sampler.z.weight.shape) == (256, 128, 1, 1)
    P1, P2 = syn.make_pairs(1, 4, size, seed=2, jitter=(0.03, 2.0))
    img = torch.randn(4, 3, size, size, device="cuda")
    with torch.no_grad():
        src = net(img.roll(-1, 0))[0]
        feat, heat, locs, scos, corr, depth, sl, _ = net(img, [src, P2, None, P1, None, None, None])
    assert tuple(heat[0].shape) == (4, 20, hs, hs) and tuple(depth.shape) == (4, 8, hs, hs)      # K / 2 pooled samples
    assert torch.isfinite(heat[0]).all() and tuple(locs.shape) == (4, 20, 2)


def test_debug_mode_returns_the_reference_nine_tuple():
    """Epipolar(debug=True) (epipolar.py:264-265, the visualisers): finalout, corr_pos, depth, sample_locs (K,N,H,W,2,
    untransposed) + intersections, mask, valid_intersections, start, vec -- the geometry intermediates restated in torch
    ops must be consistent with the HIP kernel's sample locations: sample k = normalize(coord2pix(start + vec * step_k))."""
    from epipolar_transformers_amd import default_cfg, ops, synthetic as syn
    from epipolar_transformers_amd.epipolar import Epipolar

    H, C, K = 16, 8, 8
    cfg = default_cfg()
    cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (H, H), "KEYPOINT.NFEATS", C, "EPIPOLAR.SAMPLESIZE", K,
                         "DATASETS.IMAGE_SIZE", (4 * H, 4 * H), "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                         "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True])
    mod = Epipolar(debug=True, cfg=cfg).cuda().eval()
    plain = Epipolar(cfg=cfg).cuda().eval()
    plain.load_state_dict(mod.state_dict())
    P1, P2 = syn.make_pairs(1, 4, 4 * H, seed=5, jitter=(0.05, 3.0))
    f1, f2 = syn.make_features(4, C, H, H, seed=5)
    with torch.no_grad():
        out = mod(f1.cuda(), f2.cuda(), P1, P2)
        fin, corr, depth, none = plain(f1.cuda(), f2.cuda(), P1, P2)
    assert len(out) == 9 and none is None
    assert torch.equal(out[0], fin) and torch.equal(out[1], corr) and torch.equal(out[2], depth)
    locs, inter, mask, valid, start, vec = out[3:]
    N = 4
    assert tuple(locs.shape) == (K, N, H, H, 2) and tuple(inter.shape) == (N, H * H, 4, 2) and tuple(mask.shape) == (N, H * H, 4)
    assert tuple(valid.shape) == (N, H * H, 2, 2) and tuple(start.shape) == (N, H * H, 2) and tuple(vec.shape) == (1, N, H * H, 2)
    assert mask.dtype == torch.bool and ((mask.sum(-1) == 0) | (mask.sum(-1) >= 2)).all()
    spec = mod.layer_spec()
    steps = spec.steps.cuda().view(K, 1, 1, 1)
    pos = start.view(1, N, H * H, 2) + vec * steps                        # epipolar.py:409
    pix = (pos + 0.5 - 4 / 2.0) / 4                                       # coord2pix (multiview.py:163), resize factors 1
    norm = -1 + 2 * pix / (H - 1)                                         # normalize, USE_CORRECT_NORMALIZE (multiview.py:30-32)
    assert (norm.view(K, N, H, H, 2) - locs).abs().max().item() <= 2e-4


# ---- the parameterised / pooled / prior branches as ONE HIP kernel (et_epipolar_forward_general, forward only) ----------
GENERAL_KERNEL_MODES = MODES          # every fixture's branch has a forward on the general kernel


@pytest.mark.parametrize("name", GENERAL_KERNEL_MODES)
def test_general_kernel_vs_reference(name):
    """No gradient requested -> the module takes `et_epipolar_forward_general`; its outputs against the REAL reference's
    (theta / phi / g with BOTTLENECK 2 + POOLING; PRIOR added / multiplied; cosine similarity; ATTENTION max; FIND_CORR rgb)."""
    d = np.load(os.path.join(GOLDEN_DIR, "modes", name + ".npz"))
    mod = _module(d)
    dev = lambda k: torch.from_numpy(d[k]).cuda()
    f1, f2 = dev("feat1"), dev("feat2")
    kw = _call_kwargs(d, name, dev)
    with torch.no_grad():
        assert "depth" in kw or mod._general_kernel_applies(f1, f2, kw.get("ref1"), kw.get("ref2"))
    called = []
    from epipolar_transformers_amd import ops
    real = ops.forward_general_nhwc
    ops.forward_general_nhwc = lambda *a, **k: (called.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            fin, corr, depth, _ = mod(f1, f2, torch.from_numpy(d["P1"]), torch.from_numpy(d["P2"]), **kw)
    finally:
        ops.forward_general_nhwc = real
    assert called, "the HIP general kernel did not run"
    assert tuple(fin.shape) == d["finalout"].shape and tuple(depth.shape) == d["depth"].shape
    locs = ops.sample_locs(mod.layer_spec(), torch.from_numpy(d["cam"]).cuda()).cpu().numpy()
    depth_np = depth.cpu().numpy()
    ties = assert_corr_pos(locs, corr.cpu().numpy(), d["corr_pos"], depth_np, True, tie=2e-6, max_frac=2e-2)
    assert np.abs(depth_np - d["depth"]).max() <= 1e-5 * max(1.0, float(np.abs(d["depth"]).max()))
    err = np.abs(fin.cpu().numpy() - d["finalout"])
    tol = 1e-4 * max(1.0, float(np.abs(d["finalout"]).max()))
    if "attention_max" in name or "depth_given_max" in name:        # (a proven arg-max tie gathers the other, equally good sample: exempt, as above)
        assert (err.max(1) <= tol)[~ties].all()
    else:
        assert err.max() <= tol


@pytest.mark.parametrize("case", [
    dict(H=64, C=256, K=64, N=4, bottleneck=2, pooling=True, prior=False, softmax=True),     # keypoint_h36m_param.yaml's head
    dict(H=24, C=64, K=33, N=3, bottleneck=1, pooling=False, prior=True, softmax=True),      # ragged K, prior added
    dict(H=16, C=32, K=130, N=2, bottleneck=4, pooling=True, prior=True, priormul=True, softmax=True),   # K' = 65 > one wave
    dict(H=16, C=16, K=12, N=2, bottleneck=1, pooling=True, prior=False, softmax=False),     # soft-max off: sim / K'
    dict(H=24, C=32, K=20, N=3, bottleneck=2, pooling=True, prior=True, softmax=True, similarity="cos"),
    dict(H=24, C=32, K=20, N=3, bottleneck=1, pooling=False, prior=False, softmax=True, attention="max"),
])
def test_general_kernel_vs_torch_restatement(case):
    """The HIP kernel against the chunked torch restatement of the same branches (itself pinned to the reference fixtures
    by test_mode_vs_reference) at shapes the fixtures do not reach."""
    from epipolar_transformers_amd import default_cfg, synthetic as syn
    from epipolar_transformers_amd.epipolar import Epipolar

    H, C, K, N = case["H"], case["C"], case["K"], case["N"]
    par = ("z",) + (("theta", "phi", "g") if case["bottleneck"] > 1 else ("phi",))
    cfg = default_cfg()
    cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (H, H), "KEYPOINT.NFEATS", C, "EPIPOLAR.SAMPLESIZE", K,
                         "DATASETS.IMAGE_SIZE", (4 * H, 4 * H), "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                         "EPIPOLAR.ATTENTION", case.get("attention", "avg"), "EPIPOLAR.SIMILARITY", case.get("similarity", "dot"),
                         "EPIPOLAR.PARAMETERIZED", par, "EPIPOLAR.BOTTLENECK", case["bottleneck"],
                         "EPIPOLAR.ZRESIDUAL", case["bottleneck"] == 1, "EPIPOLAR.POOLING", case["pooling"],
                         "EPIPOLAR.PRIOR", case["prior"], "EPIPOLAR.PRIORMUL", bool(case.get("priormul")),
                         "EPIPOLAR.SOFTMAX_ENABLED", case["softmax"], "DATASETS.CAMERAS", (0, 1, 2, 3)])
    torch.manual_seed(7)
    mod = Epipolar(cfg=cfg).cuda().eval()
    with torch.no_grad():
        for nm in ("theta", "phi", "g"):
            if nm in par:
                getattr(mod, nm).weight.normal_(0, 0.3)
                getattr(mod, nm).bias.normal_(0, 0.3)
        if case["prior"]:
            mod.prior = {k: torch.nn.Parameter(torch.rand(K // 2 if case["pooling"] else K, H, H, device="cuda") * 0.5)
                         for k in mod.prior}
    P1, P2 = syn.make_pairs((N + 3) // 4, 4, 4 * H, seed=11, jitter=(0.05, 4.0))
    P1, P2 = P1[:N], P2[:N]
    f1, f2 = syn.make_features(N, C, H, H, seed=12)
    f1, f2 = f1.cuda(), f2.cuda()
    f2[:, :, 2, 3] = 0.0                                    # an all-zero source pixel: exact-zero taps
    f1[:, :, 5, 5] = 0.0                                    # an all-zero reference pixel (dot == 0 -> masked everywhere unless biased)
    cams = torch.arange(N) % 4, (torch.arange(N) + 1) % 4
    with torch.no_grad():
        assert mod._general_kernel_applies(f1, f2)
        out_h, attn_h, corr_h = mod._attend_general(f1, f2, P1, P2, cams[0], cams[1])
        out_t, attn_t, corr_t = mod._attend_general_chunk(f1, f2, P1, P2, cams[0], cams[1])
    assert attn_h.shape == attn_t.shape and out_h.shape == out_t.shape
    assert (attn_h - attn_t).abs().max().item() <= 1e-5 * max(1.0, attn_t.abs().max().item())
    from epipolar_transformers_amd import ops
    locs = ops.sample_locs(mod.layer_spec(), mod._cam(P1, P2, f1.device)).cpu().numpy()
    ties = assert_corr_pos(locs, corr_h.cpu().numpy(), corr_t.cpu().numpy(), attn_h.cpu().numpy(), True, tie=2e-6, max_frac=2e-2)
    ok = (out_h - out_t).abs().amax(1) <= 1e-4 * max(1.0, out_t.abs().max().item())          # (N,H,W)
    assert ok.all() if case.get("attention") != "max" else ok[~torch.from_numpy(ties).cuda()].all()


@pytest.mark.parametrize("name", MODES)
def test_general_kernel_routing(name):
    """Every option-mode fixture runs on the HIP kernels WITH a gradient requested (ops.GeneralAttend: forward and, since
    ABI 12, backward for cosine / ATTENTION max / both priors too): the slow-path warning is unreachable from them."""
    import warnings

    from epipolar_transformers_amd.epipolar import EpipolarSlowPathWarning

    d = np.load(os.path.join(GOLDEN_DIR, "modes", name + ".npz"))
    mod = _module(d)
    dev = lambda k: torch.from_numpy(d[k]).cuda()
    f1, f2 = dev("feat1").requires_grad_(True), dev("feat2").requires_grad_(True)
    kw = _call_kwargs(d, name, dev)
    assert "depth" in kw or mod._general_kernel_applies(f1, f2, kw.get("ref1"), kw.get("ref2"), kw["camera"], kw["other_camera"])
    with warnings.catch_warnings():
        warnings.simplefilter("error", EpipolarSlowPathWarning)
        fin, _, _, _ = mod(f1, f2, torch.from_numpy(d["P1"]), torch.from_numpy(d["P2"]), **kw)
        (fin * dev("grad_out")).sum().backward()
    assert torch.isfinite(f2.grad).all()


@pytest.mark.parametrize("case", [
    dict(H=24, C=32, K=20, N=3, bottleneck=2, pooling=True, softmax=True, similarity="cos"),
    dict(H=16, C=16, K=12, N=2, bottleneck=1, pooling=False, softmax=False, similarity="cos"),
    dict(H=24, C=32, K=20, N=3, bottleneck=1, pooling=False, softmax=True, attention="max"),
    dict(H=16, C=32, K=24, N=2, bottleneck=2, pooling=True, softmax=True, attention="max"),
    dict(H=24, C=64, K=33, N=4, bottleneck=1, pooling=False, softmax=True, prior=True),
    dict(H=16, C=32, K=130, N=4, bottleneck=4, pooling=True, softmax=True, prior=True, priormul=True),
    dict(H=16, C=16, K=12, N=3, bottleneck=1, pooling=False, softmax=False, prior=True),
    dict(H=16, C=16, K=12, N=3, bottleneck=1, pooling=False, softmax=False, prior=True, priormul=True),
    dict(H=20, C=32, K=16, N=4, bottleneck=2, pooling=False, softmax=True, prior=True, similarity="cos"),
    dict(H=20, C=16, K=16, N=4, bottleneck=1, pooling=False, softmax=True, prior=True, similarity="prior"),
    dict(H=16, C=16, K=16, N=4, bottleneck=1, pooling=True, softmax=True, prior=True, similarity="prior"),
], ids=lambda c: "-".join("%s%s" % (k, v) for k, v in c.items() if k not in ("H", "N", "C")))
def test_option_branch_gradients_vs_torch_restatement(case):
    """The HIP backward of the remaining option branches (rows a12 / N4: SIMILARITY cos (epipolar.py:290-293), ATTENTION max
    (:225-235: the gradient reaches the arg-max sample only), additive / multiplicative PRIOR incl. the prior tables' own
    gradient (:288-289, 300-301, 308-309), SIMILARITY prior) against autograd through the torch restatement of the same
    branches: both feature maps, the theta / phi / g convolutions, every prior table."""
    from epipolar_transformers_amd import default_cfg, synthetic as syn
    from epipolar_transformers_amd.epipolar import Epipolar

    H, C, K, N = case["H"], case["C"], case["K"], case["N"]
    par = ("z", "theta", "phi", "g")
    has_prior = bool(case.get("prior"))
    cfg = default_cfg()
    cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (H, H), "KEYPOINT.NFEATS", C, "EPIPOLAR.SAMPLESIZE", K,
                         "DATASETS.IMAGE_SIZE", (4 * H, 4 * H), "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                         "EPIPOLAR.ATTENTION", case.get("attention", "avg"), "EPIPOLAR.SIMILARITY", case.get("similarity", "dot"),
                         "EPIPOLAR.PARAMETERIZED", par, "EPIPOLAR.BOTTLENECK", case["bottleneck"],
                         "EPIPOLAR.ZRESIDUAL", case["bottleneck"] == 1, "EPIPOLAR.POOLING", case["pooling"],
                         "EPIPOLAR.PRIOR", has_prior, "EPIPOLAR.PRIORMUL", bool(case.get("priormul")),
                         "EPIPOLAR.SOFTMAX_ENABLED", case["softmax"], "DATASETS.CAMERAS", (0, 1, 2, 3)])
    torch.manual_seed(5)
    mod = Epipolar(cfg=cfg).cuda().eval()
    with torch.no_grad():
        for nm in ("theta", "phi", "g"):
            getattr(mod, nm).weight.normal_(0, 0.3)
            getattr(mod, nm).bias.normal_(0, 0.3)
    if has_prior:
        mod.prior = {k: torch.nn.Parameter(torch.rand(K // 2 if case["pooling"] else K, H, H, device="cuda") * 0.5 + 0.05)
                     for k in mod.prior}
    P1, P2 = syn.make_pairs(1, 4, 4 * H, seed=31, jitter=(0.05, 4.0))
    P1, P2 = P1[:N], P2[:N]
    f1, f2 = syn.make_features(N, C, H, H, seed=32)
    f2[:, :, 2, 3] = 0.0
    f1[:, :, 5, 5] = 0.0
    cams = (torch.arange(N) % 2, (torch.arange(N) % 2 + 1))          # pairs 0, 2 and 1, 3 share a table: their gradients add
    gout = torch.randn(N, C // case["bottleneck"], H, H, device="cuda")
    grads, outs = [], []
    for path in ("hip", "torch"):
        a, b = f1.cuda().requires_grad_(True), f2.cuda().requires_grad_(True)
        mod.zero_grad()
        for q in mod.prior.values():
            q.grad = None
        if path == "hip":
            assert mod._general_kernel_applies(a, b, None, None, cams[0], cams[1])
        out, attn, corr = (mod._attend_general if path == "hip" else mod._attend_general_chunk)(a, b, P1, P2, cams[0], cams[1])
        (out * gout).sum().backward()
        z = lambda t, like: torch.zeros_like(like) if t is None else t.clone()
        g = [z(a.grad, a), z(b.grad, b)] + [z(q.grad, q) for nm in ("theta", "phi", "g") for q in getattr(mod, nm).parameters()]
        g += [z(mod.prior[k].grad, mod.prior[k]) for k in sorted(mod.prior) if k in ((0, 1), (1, 2))]
        grads.append(g)
        outs.append((out.detach(), corr, attn.detach()))
    names = ["feat1", "feat2", "theta.w", "theta.b", "phi.w", "phi.b", "g.w", "g.b"] + (["prior(0,1)", "prior(1,2)"] if has_prior else [])
    is_max = case.get("attention") == "max"
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-4 * max(1.0, outs[1][0].abs().max().item()) or is_max
    flipped = 0.0
    if is_max:
        # ATTENTION max: where the two paths pick different samples it must be a PROVEN tie of the cosine similarities
        # (equal to 2e-6 in our own `attn`); such a pixel then sends its whole gradient to the other sample
        from epipolar_transformers_amd import ops
        locs = ops.sample_locs(mod.layer_spec(), mod._cam(P1, P2, torch.device("cuda"))).cpu().numpy()
        ties = assert_corr_pos(locs, outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), outs[0][2].cpu().numpy(), True, tie=2e-6, max_frac=2e-2)
        flipped = float(ties.mean())
    for nm, gh, gt in zip(names, *grads):
        tol = 2e-4 * max(gt.abs().max().item(), 1e-6)
        bad = (gh - gt).abs() > tol
        if is_max and nm in ("feat1", "feat2"):
            assert bad.float().mean().item() <= 5e-3 + 40 * flipped, (nm, bad.float().mean().item(), flipped)
        elif is_max:
            # (a weight gradient sums over ALL pixels: every flipped pixel moves it by that pixel's share)
            assert (gh - gt).abs().max().item() <= (2e-4 + 4 * flipped) * max(gt.abs().max().item(), 1e-6), (nm, flipped)
        else:
            assert not bad.any(), (nm, (gh - gt).abs().max().item(), gt.abs().max().item())
    if has_prior and not is_max and not (case.get("priormul") and not case["softmax"]):
        # the tables really receive a gradient (PRIORMUL only acts behind the soft-max, epipolar.py:308-309: without it
        # the tables are not used at all and their gradient is exactly zero on both paths)
        assert grads[0][-1].abs().max().item() > 0


@pytest.mark.parametrize("case", [
    dict(H=24, C=64, K=33, N=3, bottleneck=2, pooling=False, softmax=True),
    dict(H=16, C=32, K=130, N=2, bottleneck=4, pooling=True, softmax=True),
    dict(H=16, C=16, K=12, N=2, bottleneck=1, pooling=True, softmax=False),
    dict(H=16, C=16, K=8, N=2, bottleneck=1, pooling=False, softmax=True, other_grad=("other2",)),
])
def test_general_kernel_gradients_vs_torch_restatement(case):
    """ops.GeneralAttend (HIP forward + HIP backward, float atomics) against autograd through the torch restatement:
    gradients of both feature maps and of the theta / phi / g convolutions."""
    from epipolar_transformers_amd import default_cfg, synthetic as syn
    from epipolar_transformers_amd.epipolar import Epipolar

    H, C, K, N = case["H"], case["C"], case["K"], case["N"]
    par = ("z", "theta", "phi", "g")
    cfg = default_cfg()
    cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (H, H), "KEYPOINT.NFEATS", C, "EPIPOLAR.SAMPLESIZE", K,
                         "DATASETS.IMAGE_SIZE", (4 * H, 4 * H), "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                         "EPIPOLAR.ATTENTION", "avg", "EPIPOLAR.PARAMETERIZED", par, "EPIPOLAR.BOTTLENECK", case["bottleneck"],
                         "EPIPOLAR.ZRESIDUAL", case["bottleneck"] == 1, "EPIPOLAR.POOLING", case["pooling"],
                         "EPIPOLAR.SOFTMAX_ENABLED", case["softmax"], "EPIPOLAR.OTHER_GRAD", case.get("other_grad", ("other1", "other2"))])
    torch.manual_seed(3)
    mod = Epipolar(cfg=cfg).cuda().eval()
    with torch.no_grad():
        for nm in ("theta", "phi", "g"):
            getattr(mod, nm).weight.normal_(0, 0.3)
            getattr(mod, nm).bias.normal_(0, 0.3)
    P1, P2 = syn.make_pairs(1, 4, 4 * H, seed=21, jitter=(0.05, 4.0))
    P1, P2 = P1[:N], P2[:N]
    f1, f2 = syn.make_features(N, C, H, H, seed=22)
    gout = torch.randn(N, C // case["bottleneck"], H, H, device="cuda")
    grads = []
    for path in ("hip", "torch"):
        a, b = f1.cuda().requires_grad_(True), f2.cuda().requires_grad_(True)
        mod.zero_grad()
        assert mod._general_kernel_applies(a, b)
        out, attn, corr = (mod._attend_general if path == "hip" else mod._attend_general_chunk)(a, b, P1, P2)
        (out * gout).sum().backward()
        grads.append([a.grad.clone(), b.grad.clone()] + [q.grad.clone() for nm in ("theta", "phi", "g") for q in getattr(mod, nm).parameters()])
    names = ["feat1", "feat2", "theta.w", "theta.b", "phi.w", "phi.b", "g.w", "g.b"]
    for nm, gh, gt in zip(names, *grads):
        assert (gh - gt).abs().max().item() <= 2e-4 * max(gt.abs().max().item(), 1e-6), nm


def test_general_kernel_rejects_bad_arguments():
    """Error behaviour of the general entry points (status + message, never a silent wrong result)."""
    from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn

    H, K, N = 8, 9, 2
    dev = torch.device("cuda:0")
    P1, P2 = syn.make_pairs(1, 4, 4 * H, seed=1, jitter=(0.05, 2.0))
    cam = camera.pair_algebra(P1[:N], P2[:N]).to(dev)
    spec = ops.LayerSpec(H=H, W=H, K=K)
    q = torch.randn(N, H, H, 8, device=dev)
    with pytest.raises(_lib.EpipolarAmdError, match="even K"):
        ops.forward_general_nhwc(spec, q, q.clone(), q.clone(), cam, pooling=True)                 # POOLING needs an even K
    big = torch.randn(N, H, H, 516, device=dev)
    with pytest.raises(_lib.EpipolarAmdError, match="c_sim"):
        ops.forward_general_nhwc(spec, big, big.clone(), q.clone(), cam)                          # c_sim > 512
    with pytest.raises(ValueError, match="prior"):
        ops.forward_general_nhwc(spec, q, q.clone(), q.clone(), cam, prior=torch.zeros(N, K + 1, H, H, device=dev))
    with pytest.raises(_lib.EpipolarAmdError, match="no CPU fallback"):
        ops.forward_general_nhwc(spec, q.cpu(), q.cpu(), q.cpu(), cam)
    with pytest.raises(_lib.EpipolarAmdError, match="PRIOR_MUL"):
        ops.forward_general_nhwc(spec, q, q.clone(), q.clone(), cam, prior_mul=True)              # multiplies a prior that is not there
    out, attn, corr = ops.forward_general_nhwc(spec, q, q.clone(), q.clone(), cam)
    assert torch.isfinite(out).all() and (attn.sum(1) - 1).abs().max().item() < 1e-5
