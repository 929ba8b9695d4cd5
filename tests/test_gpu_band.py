"""The persistent kernel's instance for maps above 64 x 64 (`epipolar_fwd_tile_ws_kernel<288, 8, *, true>`: 288-row arrays, a
slot table over the tile's 16-row band instead of the whole map, two columns per lane in S2; csrc/kernels_forward_tile_ws.inc):

  * forced onto small maps (ET_VARIANT_WS_BAND) it must return what the default instance returns: the same row sets, the same
    arithmetic in the same order -- bit for bit;
  * at 96 x 96 and on non-square maps (x- and y-major tiles) against the per-pixel kernels (plain fp32, no tiles), also with
    every tile forced onto the overflow list (64-row capacity, K = 16: the one-block-per-tile kernel at 384 rows takes them).
The BASELINE-shape tests against the oracle (tests/test_gpu_parity.py: configs[3] = 96 x 96, K = 64) run through this kernel too.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

C = 256


def _inputs(n, h, w, seed):
    from epipolar_transformers_amd import camera, synthetic as syn

    P1, P2 = syn.make_pairs((n + 3) // 4, 4, 4 * max(h, w), seed=seed, jitter=(0.05, 8.0))
    g = torch.Generator().manual_seed(seed)
    ref = torch.randn(n, h, w, C, generator=g).relu_().cuda()
    src = torch.randn(n, h, w, C, generator=g).relu_().cuda()
    return ref, src, camera.pair_algebra(P1[:n], P2[:n]).cuda()


def _overflow(ws):
    base = (-ws.data_ptr()) % 256
    return int(ws[base:base + 4].view(torch.int32).item())


@pytest.mark.parametrize("n,h,w,k", [(6, 64, 64, 64), (5, 48, 48, 33), (4, 16, 16, 16), (4, 33, 20, 20), (3, 20, 60, 31)],
                         ids=["64x64-K64", "48x48-K33", "16x16-K16", "33x20-K20", "20x60-K31"])
def test_band_instance_returns_what_the_default_instance_returns(n, h, w, k):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from epipolar_transformers_amd import _lib, ops

    ops.POISON_OUTPUTS = True
    ref, src, cam = _inputs(n, h, w, 300 + h + w)
    s0 = ops.LayerSpec(H=h, W=w, K=k)
    s1 = ops.LayerSpec(H=h, W=w, K=k, variant=_lib.ET_VARIANT_WS_BAND)
    ws0, ws1 = ops.tile_workspace(s0, n, C, ref.device), ops.tile_workspace(s1, n, C, ref.device)
    out0, attn0, corr0 = ops.forward_nhwc(s0, ref, src, cam, workspace=ws0)
    out1, attn1, corr1 = ops.forward_nhwc(s1, ref, src, cam, workspace=ws1)
    torch.cuda.synchronize()
    ops.check_tile_errors(workspace=ws1)
    assert _overflow(ws0) == _overflow(ws1) == 0
    # the same row sets, tile by tile
    assert torch.equal(ops.tile_stats(s0, n, C, ws0), ops.tile_stats(s1, n, C, ws1))
    # ... and the same arithmetic in the same order: bit for bit
    assert torch.equal(corr1, corr0) and torch.equal(attn1, attn0) and torch.equal(out1, out0)


@pytest.mark.parametrize("n,h,w,k,variant", [(3, 96, 96, 64, 0), (3, 80, 72, 40, 0), (2, 40, 96, 64, 0), (2, 96, 96, 16, 32768),
                                             (2, 72, 88, 16, 32768)],
                         ids=["96x96-K64", "80x72-K40", "40x96-K64", "96x96-K16-all-tiles-overflow", "72x88-K16-all-tiles-overflow"])
def test_band_instance_vs_per_pixel_kernels(n, h, w, k, variant):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from epipolar_transformers_amd import _lib, ops

    ops.POISON_OUTPUTS = True
    ref, src, cam = _inputs(n, h, w, 500 + h + w)
    spec = ops.LayerSpec(H=h, W=w, K=k, variant=variant)
    ws = ops.tile_workspace(spec, n, C, ref.device)
    out, attn, corr = ops.forward_nhwc(spec, ref, src, cam, workspace=ws)
    want_out, want_attn, want_corr = ops.forward_nhwc(ops.LayerSpec(H=h, W=w, K=k, variant=_lib.ET_VARIANT_NO_TILE), ref, src, cam)
    torch.cuda.synchronize()
    ops.check_tile_errors(workspace=ws)
    assert torch.isfinite(out).all() and torch.isfinite(attn).all()
    assert (attn - want_attn).abs().max().item() <= 1e-5
    assert (out - want_out).abs().max().item() <= 1e-4 * max(1.0, want_out.abs().max().item())
    assert (corr != want_corr).any(-1).float().mean().item() <= 1e-3          # (exact except at soft-max ties)
    tiles = n * ((h * w + 31) // 32)
    if variant:
        assert _overflow(ws) > tiles // 2, "the case was meant to put the tiles on the overflow list"
    else:
        assert _overflow(ws) <= max(1, tiles // 200)
        u = ops.tile_stats(spec, n, C, ws) & 0xffff
        assert int(u.max()) <= 288
