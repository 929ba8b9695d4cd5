"""et_epipolar_forward_fused: the eval-mode layer as ONE data kernel -- the persistent sample + attention kernel with
x = feat_ref + bias + out . Wf^T (bn(z(out)) + out + feat, epipolar.py:250-253 + resnet.py:388, BN folded into z) as a third
GEMM on the tile's out rows -- against

  * the two-kernel path it replaces (et_epipolar_forward_tiled + et_residual_gemm): attention / corr_pos bit for bit,
    x to fp32-GEMM-level error;
  * float64 of the epilogue applied to `out` (the bound of tests/test_gpu_parity.py::test_residual_gemm_abi);
  * the C oracle's forward + the reference's own op sequence for the epilogue (conv1x1 + eval batch-norm + two adds);
  * with tiles forced onto the overflow list (row-set capacity 64; a source value beyond fp16's range): their `out` rows come
    from the one-block-per-tile kernel and their x rows from the follow-up kernel.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C = 256


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from epipolar_transformers_amd import _lib, camera, ops

    _lib.load()
    return _lib, camera, ops


def _inputs(n, h, seed, outlier=False):
    from epipolar_transformers_amd import synthetic as syn

    P1, P2 = syn.make_pairs((n + 3) // 4, 4, 4 * h, seed=seed, jitter=(0.05, 8.0))
    P1, P2 = P1[:n], P2[:n]
    f1, f2 = syn.make_features(n, C, h, h, seed=seed + 1)
    if outlier:
        f2[n - 1, 100, 5, 6] = 4.0e4          # beyond fp16 under the estimated scale: those tiles are redone in exact fp32
    g = torch.Generator().manual_seed(seed + 2)
    wf = torch.randn(C, C, generator=g) * 0.05 + torch.eye(C)
    bias = torch.randn(C, generator=g)
    return P1, P2, f1, f2, wf, bias


def _bound(out, wf, want):
    return 4e-6 * (out.double().abs() @ wf.double().abs().t()) + 3e-7 * (want.abs() + 1)


@pytest.mark.parametrize("n,h,k,variant,outlier", [(8, 64, 64, 0, False), (5, 48, 33, 0, False), (4, 16, 16, 0, False),
                                                   (4, 16, 16, 32768, False), (4, 64, 64, 0, True), (3, 33, 20, 0, False),
                                                   (3, 96, 64, 0, False), (2, 96, 16, 32768, False), (2, 80, 40, 0, True),
                                                   (4, 48, 33, 1048576, False)],
                         ids=["64x64-K64", "48x48-K33", "16x16-K16", "16x16-K16-overflow-tiles", "64x64-K64-fp16-guard", "33x33-K20",
                              "96x96-K64", "96x96-K16-overflow-tiles", "80x80-K40-fp16-guard", "48x48-K33-band-instance"])
def test_fused_layer_vs_two_kernels_and_float64(env, n, h, k, variant, outlier):
    _lib, camera, ops = env
    P1, P2, f1, f2, wf, bias = _inputs(n, h, seed=50 + h + k, outlier=outlier)
    spec = ops.LayerSpec(H=h, W=h, K=k, variant=variant)
    assert ops.fused_layer_applies(spec, C, n)
    ref, src = ops.to_nhwc(f1.cuda()), ops.to_nhwc(f2.cuda())
    cam = camera.pair_algebra(P1, P2).cuda()
    wf_d, bias_d = wf.cuda(), bias.cuda()
    packed = ops.residual_gemm_pack(wf_d)
    # the two-kernel path
    out2, attn2, corr2 = ops.forward_nhwc(spec, ref, src, cam)
    x2 = ops.residual_gemm(out2, packed, bias_d, ref)
    # one kernel
    ws = ops.tile_workspace(spec, n, C, ref.device)
    x1, attn1, corr1, out1 = ops.forward_fused_nhwc(spec, ref, src, cam, packed, bias_d, want_out=True, workspace=ws)
    torch.cuda.synchronize()
    ops.check_tile_errors(workspace=ws)
    assert torch.isfinite(x1).all()
    assert torch.equal(attn1, attn2) and torch.equal(corr1, corr2) and torch.equal(out1, out2)
    want = out2.double().reshape(-1, C) @ wf_d.double().t() + bias_d.double() + ref.double().reshape(-1, C)
    err = (x1.double().reshape(-1, C) - want).abs()
    bound = _bound(out2.reshape(-1, C), wf_d, want)
    assert (err <= bound).all(), (err / bound).max().item()
    assert (x1 - x2).abs().max().item() <= 2e-5 * max(1.0, x2.abs().max().item())
    # without `out` (the product path): same x, and the scratch is only touched for overflow tiles
    x3, attn3, corr3 = ops.forward_fused_nhwc(spec, ref, src, cam, packed, bias_d)
    assert torch.equal(x3, x1) and torch.equal(attn3, attn1) and torch.equal(corr3, corr1)
    if (variant & 32768) or outlier:
        base = (-ws.data_ptr()) % 256
        assert int(ws[base:base + 4].view(torch.int32).item()) > 0, "the case was meant to put tiles on the overflow list"


def test_fused_layer_vs_oracle_and_reference_epilogue(env, oracle_mod):
    """Forward of the C oracle, then the reference's own op sequence for bn(z(out)) + out + feat in eval mode."""
    _lib, camera, ops = env
    n, h, k = 4, 32, 24
    P1, P2, f1, f2, _, _ = _inputs(n, h, seed=91)
    g = torch.Generator().manual_seed(7)
    zw, zb = torch.randn(C, C, 1, 1, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    mean, var = 0.1 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
    cam = camera.pair_algebra(P1, P2)
    want = oracle_mod.forward(oracle_mod.LayerSpec(h, h, k), f1, f2, None, None, cam=cam.numpy())
    _, want_x = oracle_mod.epilogue(want["out"], f1.numpy(), zw.numpy(), zb.numpy(), gamma.numpy(), beta.numpy(), mean.numpy(),
                                    var.numpy(), training=False)
    s = gamma / torch.sqrt(var + 1e-5)
    wf = (zw.view(C, C) * s[:, None] + torch.eye(C)).cuda()
    bf = (zb * s + beta - mean * s).cuda()
    x, attn, corr = ops.forward_fused_nhwc(ops.LayerSpec(H=h, W=h, K=k), ops.to_nhwc(f1.cuda()), ops.to_nhwc(f2.cuda()), cam.cuda(),
                                           ops.residual_gemm_pack(wf), bf)
    x = x.permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(attn.cpu().numpy() - want["attn"]).max() <= 1e-5
    assert np.abs(x - want_x.numpy()).max() <= 1e-4


def test_fused_layer_is_what_the_module_runs_in_eval(env):
    """Epipolar.forward_fused (what PoseResNet._fuse calls) takes the one-kernel path for the 256-channel head and matches the
    two-kernel path it replaces (EPIPOLAR_AMD.FUSED_GEMM3 False)."""
    from epipolar_transformers_amd import default_cfg
    from epipolar_transformers_amd.epipolar import Epipolar

    _lib, camera, ops = env
    h, k = 32, 16
    P1, P2, f1, f2, _, _ = _inputs(4, h, seed=17)
    outs = []
    for fused in (True, False):
        cfg = default_cfg()
        cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (h, h), "KEYPOINT.NFEATS", C, "EPIPOLAR.SAMPLESIZE", k, "EPIPOLAR.ATTENTION", "avg",
                             "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True, "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                             "DATASETS.IMAGE_SIZE", (4 * h, 4 * h), "EPIPOLAR_AMD.FUSED_GEMM3", fused])
        torch.manual_seed(3)
        mod = Epipolar(cfg=cfg).cuda().eval()
        with torch.no_grad():
            mod.bn.weight.normal_(1, 0.1)
            mod.bn.bias.normal_(0, 0.1)
            mod.bn.running_mean.normal_(0, 0.1)
            mod.bn.running_var.uniform_(0.5, 1.5)
            calls = []
            keep = ops.forward_fused_nhwc
            ops.forward_fused_nhwc = lambda *a, **kw: (calls.append(1), keep(*a, **kw))[1]
            try:
                outs.append(mod.forward_fused(f1.cuda(), f2.cuda(), P1, P2))
            finally:
                ops.forward_fused_nhwc = keep
            assert bool(calls) == fused
    (xa, ca, aa, _), (xb, cb, ab, _) = outs
    assert torch.equal(ca, cb) and torch.equal(aa, ab)
    assert (xa - xb).abs().max().item() <= 2e-5 * max(1.0, xb.abs().max().item())


def test_fused_entry_refuses_shapes_of_the_other_kernels(env):
    _lib, camera, ops = env
    assert ops.fused_layer_applies(ops.LayerSpec(H=96, W=96, K=64), C, 2)          # (since the band-table instance: up to 96 x 96)
    assert not ops.fused_layer_applies(ops.LayerSpec(H=128, W=128, K=64), C, 2)
    assert not ops.fused_layer_applies(ops.LayerSpec(H=96, W=97, K=64), C, 2)
    assert not ops.fused_layer_applies(ops.LayerSpec(H=64, W=64, K=128), C, 2)
    assert not ops.fused_layer_applies(ops.LayerSpec(H=64, W=64, K=64, softmax_enabled=False), C, 2)
    assert not ops.fused_layer_applies(ops.LayerSpec(H=64, W=64, K=64), 128, 2)
    assert not ops.fused_layer_applies(ops.LayerSpec(H=64, W=64, K=64, variant=_lib.ET_VARIANT_TILE_CLASSIC), C, 2)
    spec = ops.LayerSpec(H=128, W=128, K=64)
    t = torch.zeros(1, 128, 128, C, device="cuda")
    with pytest.raises(_lib.EpipolarAmdError):
        ops.forward_fused_nhwc(spec, t, t, torch.zeros(1, 27, device="cuda"), ops.residual_gemm_pack(torch.eye(C, device="cuda")),
                               torch.zeros(C, device="cuda"))


# ---------------------------------------------------------------------------------------------------------------------
# The TRAINING-mode epilogue through the same GEMM kernel (round 5): et_z_batch_stats + et_residual_gemm with the batch
# statistics folded into the weight, in place of conv1x1 + batch_norm(training) + two adds as stock torch ops.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows", [64 * 5, 1587, 70000], ids=["5-blocks", "ragged-1587", "70000"])
def test_z_batch_stats_vs_float64(env, rows):
    """y, mean and biased variance against float64 -- with one channel that hardly varies around a large mean (0.2 +- 6e-3: the
    conditioning of the 8-channel fixtures; a sum-of-squares formula loses it) and one that is constant."""
    _lib, camera, ops = env
    g = torch.Generator(device="cuda").manual_seed(rows)
    out = torch.randn(rows, C, device="cuda", generator=g).relu_()
    wz = torch.randn(C, C, device="cuda", generator=g) * 0.05
    bz = torch.randn(C, device="cuda", generator=g) * 0.1
    wz[3] = 0
    wz[3, :8] = 1e-3            # channel 3: tiny spread ...
    bz[3] = 0.2                 # ... around 0.2
    wz[5] = 0                   # channel 5: constant
    y, mean, var = ops.z_batch_stats(out, ops.residual_gemm_pack(wz), bz)
    y2, mean2, var2 = ops.z_batch_stats(out, ops.residual_gemm_pack(wz), bz)
    assert torch.equal(y, y2) and torch.equal(mean, mean2) and torch.equal(var, var2)          # no atomics: bit-reproducible
    want = out.double() @ wz.double().t() + bz.double()
    assert (y.double() - want).abs().max().item() <= 4e-6 * max(1.0, want.abs().max().item())
    wm, wv = want.mean(0), want.var(0, unbiased=False)
    assert (mean.double() - wm).abs().max().item() <= 2e-6 * max(1.0, wm.abs().max().item())
    rel = ((var.double() - wv).abs() / wv.clamp_min(1e-30))
    rel[5] = 0
    assert rel.max().item() <= 1e-4, (rel.max().item(), int(rel.argmax()))
    assert abs(var[5].item()) <= 1e-12 and abs(mean[5].item() - bz[5].item()) <= 1e-7


@pytest.mark.parametrize("n,h,w,fused_x", [(3, 24, 24, False), (3, 24, 24, True), (2, 23, 23, True), (8, 64, 64, True)],
                         ids=["finalout-24x24", "x-24x24", "x-ragged-23x23", "x-64x64"])
def test_train_epilogue_kernels_vs_torch_ops(env, n, h, w, fused_x):
    """Epipolar in train mode (C = 256): outputs, running statistics and every gradient with the GEMM-kernel epilogue against
    the same module on stock torch ops (EPIPOLAR_AMD.FUSED_TRAIN_EPILOGUE False: conv1x1 + batch_norm + adds, which
    tests/test_gpu_parity.py::test_module_dropin_eval_and_train pins to the real reference)."""
    from epipolar_transformers_amd import default_cfg, synthetic as syn
    from epipolar_transformers_amd.epipolar import Epipolar

    _lib, camera, ops = env
    k = 16
    P1, P2 = syn.make_pairs((n + 3) // 4, 4, 4 * h, seed=5, jitter=(0.05, 4.0))
    P1, P2 = P1[:n], P2[:n]
    g0 = torch.Generator().manual_seed(11)
    f1, f2 = torch.randn(n, C, h, w, generator=g0).relu(), torch.randn(n, C, h, w, generator=g0).relu()
    gout = torch.randn(n, C, h, w, generator=g0).cuda()
    res = []
    for fused in (True, False):
        cfg = default_cfg()
        cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (h, w), "KEYPOINT.NFEATS", C, "EPIPOLAR.SAMPLESIZE", k, "EPIPOLAR.ATTENTION", "avg",
                             "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True, "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                             "DATASETS.IMAGE_SIZE", (4 * h, 4 * w), "EPIPOLAR_AMD.FUSED_TRAIN_EPILOGUE", fused])
        torch.manual_seed(3)
        mod = Epipolar(cfg=cfg).cuda().train()
        with torch.no_grad():
            mod.bn.weight.normal_(1, 0.1)
            mod.bn.bias.normal_(0, 0.1)
            mod.bn.running_mean.normal_(0, 0.1)
            mod.bn.running_var.uniform_(0.5, 1.5)
        a1, a2 = f1.cuda().requires_grad_(True), f2.cuda().requires_grad_(True)
        called = []
        keep = ops.z_batch_stats
        ops.z_batch_stats = lambda *a, **kw: (called.append(1), keep(*a, **kw))[1]
        try:
            y = mod.forward_fused(a1, a2, P1, P2)[0] if fused_x else mod(a1, a2, P1, P2)[0]
        finally:
            ops.z_batch_stats = keep
        assert bool(called) == fused
        (y * gout).sum().backward()
        res.append([y.detach(), mod.bn.running_mean.clone(), mod.bn.running_var.clone(), int(mod.bn.num_batches_tracked),
                    a1.grad.clone(), a2.grad.clone()] + [q.grad.clone() for q in (mod.z.weight, mod.z.bias, mod.bn.weight, mod.bn.bias)])
    names = ["y", "running_mean", "running_var", "num_batches_tracked", "d feat1", "d feat2", "d z.weight", "d z.bias", "d bn.weight", "d bn.bias"]
    for nm, a, b in zip(names, *res):
        if nm == "num_batches_tracked":
            assert a == b == 1
            continue
        scale = max(b.abs().max().item(), 1e-6)
        # (d z.bias is a sum that cancels to zero analytically -- batch norm removes the mean --: judged on the scale of d z.weight)
        tol = 2e-4 * (res[1][6].abs().max().item() if nm == "d z.bias" else scale)
        assert (a - b).abs().max().item() <= tol, (nm, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("rows,zres", [(64 * 5, True), (1587, True), (1587, False), (70000, True)], ids=["5-blocks", "ragged-1587", "ragged-no-zresidual", "70000"])
def test_z_backward_vs_float64_autograd(env, rows, zres):
    """et_z_backward (the batch norm's sums, then the GEMM kernel with dy formed on the fly) against float64 autograd through
    batch_norm(out Wz^T + bz, training) [+ out]: d out, the batch norm's input gradient, d gamma, d beta."""
    _lib, camera, ops = env
    g0 = torch.Generator(device="cuda").manual_seed(rows)
    out = torch.randn(rows, C, device="cuda", generator=g0).relu_()
    wz = torch.randn(C, C, device="cuda", generator=g0) * 0.05
    bz = torch.randn(C, device="cuda", generator=g0) * 0.1
    gamma = 1 + 0.1 * torch.randn(C, device="cuda", generator=g0)
    g = torch.randn(rows, C, device="cuda", generator=g0)
    y, mean, var = ops.z_batch_stats(out, ops.residual_gemm_pack(wz), bz)
    invstd = torch.rsqrt(var + 1e-5)
    dout, dy, dgamma, dbeta = ops.z_backward(g, y, mean, invstd, gamma, ops.residual_gemm_pack(wz.t().contiguous()), zres)
    dout2, dy2, dgamma2, dbeta2 = ops.z_backward(g, y, mean, invstd, gamma, ops.residual_gemm_pack(wz.t().contiguous()), zres)
    assert torch.equal(dout, dout2) and torch.equal(dy, dy2) and torch.equal(dgamma, dgamma2) and torch.equal(dbeta, dbeta2)
    o64 = out.double().requires_grad_(True)
    y64 = o64 @ wz.double().t() + bz.double()
    y64.retain_grad()
    ga = gamma.double().requires_grad_(True)
    be = torch.zeros(C, device="cuda", dtype=torch.float64, requires_grad=True)
    x = torch.nn.functional.batch_norm(y64, None, None, ga, be, True, 0.1, 1e-5)
    if zres:
        x = x + o64
    x.backward(g.double())
    for nm, got, want in (("d out", dout, o64.grad), ("d y", dy, y64.grad), ("d gamma", dgamma, ga.grad), ("d beta", dbeta, be.grad)):
        assert (got.double() - want).abs().max().item() <= 2e-5 * max(want.abs().max().item(), 1e-6), nm


@pytest.mark.parametrize("rows", [5, 16 * 7, 1587, 70000, 128 * 64 * 64], ids=["5", "112", "ragged-1587", "70000", "config2"])
def test_z_wgrad_vs_float64(env, rows):
    """et_z_wgrad (d Wz = dy^T out, d bz = sum dy: contracted over the rows on the matrix cores -- three bf16 terms per value, six
    cross terms per product, fp32 accumulation --, per-block partials summed in a fixed order) against float64 within the bound of an
    fp32 accumulation, and bit-reproducible."""
    _lib, camera, ops = env
    g0 = torch.Generator(device="cuda").manual_seed(rows)
    dy = torch.randn(rows, C, device="cuda", generator=g0)
    out = torch.randn(rows, C, device="cuda", generator=g0).relu_()
    dy[:, 7] *= 1e4            # (no fp16 anywhere -- bf16 keeps fp32's exponent: magnitudes do not matter)
    out[:, 9] *= 1e-6
    gw, gb = ops.z_wgrad(dy, out)
    gw2, gb2 = ops.z_wgrad(dy, out)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    want_w = dy.double().t() @ out.double()
    want_b = dy.double().sum(0)
    # fp32 accumulation over `rows` terms: a few ulp of the row-wise magnitude sum
    bound_w = 4e-7 * (dy.double().abs().t() @ out.double().abs()) + 1e-30
    assert ((gw.double() - want_w).abs() <= bound_w * max(1.0, rows ** 0.5 / 8)).all(), ((gw.double() - want_w).abs() / bound_w).max().item()
    bound_b = 4e-7 * dy.double().abs().sum(0) * max(1.0, rows ** 0.5 / 8) + 1e-30
    assert ((gb.double() - want_b).abs() <= bound_b).all()


def test_error_poll_is_asynchronous_and_survives_graph_capture(env, monkeypatch):
    """The product path (POISON_OUTPUTS off) polls the workspace's sticky error word without synchronising: every 16th call
    queues a 4-byte copy + an event.  Neither may happen while the stream is captured into a graph -- a copy / event record
    would become part of the graph and Event.query() invalidates a global-mode capture -- and the probe's state
    hangs on the workspace tensor, so a new workspace starts clean."""
    _lib, camera, ops = env
    monkeypatch.setattr(ops, "POISON_OUTPUTS", False)
    n, h, k = 4, 32, 16
    P1, P2, f1, f2, wf, bias = _inputs(n, h, seed=7)
    spec = ops.LayerSpec(H=h, W=h, K=k)
    ref, src = ops.to_nhwc(f1.cuda()), ops.to_nhwc(f2.cuda())
    cam = camera.pair_algebra(P1, P2).cuda()
    packed, bias_d = ops.residual_gemm_pack(wf.cuda()), bias.cuda()
    ws = ops.tile_workspace(spec, n, C, ref.device)
    x0 = ops.forward_fused_nhwc(spec, ref, src, cam, packed, bias_d, workspace=ws)[0]
    for _ in range(40):                      # > 2 polls: copies queued, events queried, nothing raised
        ops.forward_fused_nhwc(spec, ref, src, cam, packed, bias_d, workspace=ws)
    probe = getattr(ws, ops._PROBE_ATTR)
    assert probe.host is not None and probe.host.is_pinned()
    # a probe is pending or about to be due: capture now
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for _ in range(15):
            ops.forward_fused_nhwc(spec, ref, src, cam, packed, bias_d, workspace=ws)
        calls_before = getattr(ws, ops._PROBE_ATTR).calls
        with torch.cuda.graph(graph, stream=side):
            xg = ops.forward_fused_nhwc(spec, ref, src, cam, packed, bias_d, workspace=ws)[0]
        assert getattr(ws, ops._PROBE_ATTR).calls == calls_before, "the poll must not run during capture"
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(xg, x0)
    ops.check_tile_errors(workspace=ws)
    ws2 = ops.tile_workspace(spec, n, C, ref.device)
    assert getattr(ws2, ops._PROBE_ATTR, None) is None
