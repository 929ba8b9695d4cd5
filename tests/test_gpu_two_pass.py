"""The persistent forward for 64 < K <= 128: two passes of 64 samples per tile with an online soft-max (csrc/kernels_forward_tile_ws.inc,
KH = 2; BASELINE.json configs[4] = 128 x 128 maps, K = 128).  Against the C oracle on every camera rig, at K that are and are not
multiples of 64, on maps from 48 x 48 to 128 x 128; the paths that leave the kernel (a half's row set does not fit its band or
its 288-row array; a source value beyond fp16's range) and what couples the halves (a pixel whose samples are ALL masked, a maximum
in either half, the arg-max across the halves); repeated NaN-poisoned runs.  Reference: modeling/layers/epipolar.py:188-247, 272-321
at EPIPOLAR.SAMPLESIZE 128."""
import numpy as np
import pytest
import torch

from conftest import assert_corr_pos

pytestmark = pytest.mark.gpu

C = 256
TOL_ATTN, TOL_OUT = 1e-5, 1e-4
RIGS = ["ring", "epipole_inside", "epipole_border", "near_rectified_x", "near_rectified_y", "rectified_x", "identical", "h36m_room"]
CLASSIC = 65536


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from epipolar_transformers_amd import _lib, camera, ops

    _lib.load()
    return _lib, camera, ops


def _overflow(ws):
    base = (-ws.data_ptr()) % 256
    return int(ws[base:base + 4].view(torch.int32).item())


def _inputs(rig, n, h, seed):
    from epipolar_transformers_amd import synthetic as syn

    per = 4 if rig in ("ring", "h36m_room") else 2
    P1, P2 = syn.rig_pairs(rig, (n + per - 1) // per, 4 * h, seed=seed, jitter=None if rig == "epipole_border" else (0.05, 8.0))
    f1, f2 = syn.make_features(n, C, h, h, seed=seed + 1)
    return P1[:n], P2[:n], f1, f2


def _check(ops, oracle_mod, camera, spec, P1, P2, f1, f2, ws=None, rel_out=0.0):
    cam = camera.pair_algebra(P1, P2)
    with np.errstate(all="ignore"):
        want = oracle_mod.forward(oracle_mod.LayerSpec(spec.H, spec.W, spec.K), f1, f2, None, None, cam=cam.numpy())
    out, attn, corr = ops.forward_nhwc(spec, ops.to_nhwc(f1.cuda()), ops.to_nhwc(f2.cuda()), cam.cuda(), workspace=ws)
    torch.cuda.synchronize()
    attn_h, out_h, corr_h = attn.cpu().numpy(), out.permute(0, 3, 1, 2).cpu().numpy(), corr.cpu().numpy()
    assert np.isfinite(attn_h).all() and np.isfinite(out_h).all() and np.isfinite(corr_h).all()
    assert np.abs(attn_h - want["attn"]).max() <= TOL_ATTN
    assert (np.abs(out_h - want["out"]) - rel_out * np.abs(want["out"])).max() <= TOL_OUT
    if (corr_h != want["corr_pos"]).any():
        assert_corr_pos(want["sample_locs"], corr_h, want["corr_pos"], attn_h, True, 2e-6, 2e-2)
    return want, attn_h


@pytest.mark.parametrize("h,k", [(64, 128), (96, 100)], ids=["64x64-K128", "96x96-K100"])
@pytest.mark.parametrize("rig", RIGS)
def test_two_pass_forward_vs_oracle_on_rig(env, oracle_mod, rig, h, k):
    _lib, camera, ops = env
    P1, P2, f1, f2 = _inputs(rig, 2, h, seed=300 + h + len(rig))
    f1[0, :, 5, 7] = 0                               # an all-zero reference row: every sample masked, uniform attention 1 / K
    spec = ops.LayerSpec(H=h, W=h, K=k)
    ws = ops.tile_workspace(spec, 2, C, "cuda")
    want, attn = _check(ops, oracle_mod, camera, spec, P1, P2, f1, f2, ws)
    ops.check_tile_errors(workspace=ws)
    assert np.allclose(attn[0, :, 5, 7], 1.0 / k, atol=1e-7)
    # what did not fit a half's band / array went to the list kernels (whole tiles): a few per cent at most off the ring ...
    tiles = 2 * ((h * h + 31) // 32)
    assert _overflow(ws) <= (0 if rig in ("ring", "epipole_border", "h36m_room") else tiles // 2)
    # ... and the one-block-per-tile kernel gives the same tensors to rounding
    o2, a2, _ = ops.forward_nhwc(ops.LayerSpec(H=h, W=h, K=k, variant=CLASSIC), ops.to_nhwc(f1.cuda()), ops.to_nhwc(f2.cuda()),
                                 camera.pair_algebra(P1, P2).cuda())
    assert np.abs(a2.cpu().numpy() - attn).max() <= 3e-6


@pytest.mark.parametrize("h,k,n", [(48, 65, 3), (32, 128, 3), (64, 127, 2), (128, 128, 1), (80, 96, 2), (128, 70, 1)],
                         ids=["48x48-K65", "32x32-K128", "64x64-K127", "128x128-K128", "80x80-K96", "128x128-K70"])
def test_two_pass_forward_ragged_k(env, oracle_mod, h, k, n):
    """K = 65: the second half holds ONE sample; K = 127: one lane short; maps from 32 x 32 to 128 x 128."""
    _lib, camera, ops = env
    P1, P2, f1, f2 = _inputs("ring", n, h, seed=900 + h + k)
    _check(ops, oracle_mod, camera, ops.LayerSpec(H=h, W=h, K=k), P1, P2, f1, f2)


def test_two_pass_maximum_in_either_half_and_argmax_across_halves(env, oracle_mod):
    """Source maps built so that the best sample of some pixels lies in the first 64 samples and of others in the last 64 (one
    half of the source map is scaled up): alpha = exp(m0 - m) is 1 for the former and tiny for the latter; corr_pos must name the
    sample of the right half."""
    _lib, camera, ops = env
    h, k = 64, 128
    P1, P2, f1, f2 = _inputs("ring", 2, h, seed=77)
    f2[:, :, :, : h // 2] *= 3.0                      # strong features in the left half of the source map
    f2[1] = f2[1].flip(-1)                            # ... and in the right half for the second pair
    spec = ops.LayerSpec(H=h, W=h, K=k)
    want, attn = _check(ops, oracle_mod, camera, spec, P1, P2, f1, f2)
    best = want["attn"].argmax(1)
    assert (best < 64).mean() > 0.1 and (best >= 64).mean() > 0.1, "the case was meant to put maxima in both halves"


def test_two_pass_fp16_guard_and_overflow_tiles_go_to_the_list_once(env, oracle_mod):
    """A source value the per-pair scale pushes beyond fp16: the tiles whose FIRST half meets it, and the tiles whose SECOND half
    does, go to the exact-fp32 list kernel -- once each (the overflow list holds no tile twice) -- and every output is right."""
    _lib, camera, ops = env
    h, k = 64, 128
    P1, P2, f1, f2 = _inputs("ring", 2, h, seed=55)
    f2[0, 100, 20, 10] = 4.0e4
    f2[1, 7, 40, 50] = -6.0e4
    spec = ops.LayerSpec(H=h, W=h, K=k)
    ws = ops.tile_workspace(spec, 2, C, "cuda")
    # (outputs of magnitude 1e4 next to the outliers: fp32 rounding of the exact-fp32 redo itself is 4e-3 there, 8e-7 relative: the one-block-per-tile kernel gives the same bits)
    _check(ops, oracle_mod, camera, spec, P1, P2, f1, f2, ws, rel_out=2e-6)
    n_ovf = _overflow(ws)
    assert n_ovf > 0
    tiles = 2 * ((h * h + 31) // 32)
    base = (-ws.data_ptr()) % 256
    lst = ws[base + (64 + tiles * 32) * 4: base + (64 + tiles * 32 + n_ovf) * 4].view(torch.int32).cpu().numpy()
    assert len(set(lst.tolist())) == n_ovf, "a tile appears twice in the overflow list"


@pytest.mark.parametrize("h,k,n", [(128, 128, 6), (64, 100, 24)], ids=["128x128-K128-N6", "64x64-K100-N24"])
def test_two_pass_kernel_is_stable_over_repeated_runs(env, h, k, n):
    """Twenty NaN-poisoned runs against the per-pixel kernels, bit-identical to one another (the forward has no atomics)."""
    _lib, camera, ops = env
    P1, P2, f1, f2 = _inputs("ring", n, h, seed=11 + n)
    ref, src = ops.to_nhwc(f1.cuda()), ops.to_nhwc(f2.cuda())
    cam = camera.pair_algebra(P1, P2).cuda()
    assert ops.POISON_OUTPUTS
    o0, a0, c0 = ops.forward_nhwc(ops.LayerSpec(H=h, W=h, K=k, variant=_lib.ET_VARIANT_NO_TILE), ref, src, cam)
    tol_o = 1e-4 * max(1.0, o0.abs().max().item())
    spec = ops.LayerSpec(H=h, W=h, K=k)
    first = None
    for rep in range(20):
        o, a, c = ops.forward_nhwc(spec, ref, src, cam)
        ok = ((a - a0).abs().amax(1) <= 1e-5) & ((o - o0).abs().amax(-1) <= tol_o) & ~torch.isnan(c).any(-1)
        assert bool(ok.all()), "run %d: %d pixels differ from the per-pixel kernels" % (rep, int((~ok).sum()))
        if first is None:
            first = [t.clone() for t in (o, a, c)]
        else:
            assert all(torch.equal(t, u) for t, u in zip((o, a, c), first)), "run %d differs from run 0" % rep
    ops.check_tile_errors()
