"""CPU-side checks of the product's C ABI and host logic (no GPU, no compute
kernels launched): the library loads, exports every symbol the header declares,
the host-side camera algebra equals the reference's, and the per-sample set-up
code shared with the device kernels reproduces the reference's sample locations
bit for bit (through the CPU test hook et_debug_host_sample_setup)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_cases, load_golden

from epipolar_transformers_amd import _lib, build, camera, ops


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    text = open(os.path.join(ROOT, "include", "epipolar_amd.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char \*)\s*(et_\w+)\(", text, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.exported_symbols())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.et_abi_version() == _lib.ET_ABI_VERSION


def test_struct_layout_matches_header():
    text = open(os.path.join(ROOT, "include", "epipolar_amd.h")).read()
    body = text[text.index("typedef struct EtLayerDesc {"):text.index("} EtLayerDesc;")]
    names = []
    for line in body.splitlines()[1:]:
        line = line.split("/*")[0].strip()
        m = re.match(r"(int32_t|float)\s+([\w\s,]+);", line)
        if m:
            names += [n.strip() for n in m.group(2).split(",")]
    assert names == [f[0] for f in _lib.EtLayerDesc._fields_]
    assert ctypes.sizeof(_lib.EtLayerDesc) == 4 * len(names)


def test_validation_errors_are_reported(lib):
    d = ops.LayerSpec(H=8, W=8, K=8).desc(N=1, C=6)          # C not a multiple of 4
    rc = lib.et_epipolar_forward(ctypes.byref(d), *[ctypes.c_void_p(0)] * 12)
    assert rc != 0 and b"multiple of 4" in lib.et_last_error()
    d = ops.LayerSpec(H=8, W=8, K=8).desc(N=1, C=8)
    rc = lib.et_epipolar_forward(ctypes.byref(d), *[ctypes.c_void_p(0)] * 12)
    assert rc != 0 and b"NULL" in lib.et_last_error()


def test_empty_and_out_of_range_shapes_are_errors_not_silent(lib):
    """An empty batch makes the reference raise (`torch.stack` of nothing, epipolar.py:248); here every entry point
    reports it -- as it does K beyond 256 and more than 512 channels -- before touching the GPU."""
    null = ctypes.c_void_p(0)
    for kw, msg in ((dict(N=0, C=8), b"bad shape"), (dict(N=1, C=516), b"> 512"), (dict(N=1, C=8, K=300), b"K=300")):
        d = ops.LayerSpec(H=8, W=8, K=kw.pop("K", 8)).desc(**kw)
        assert lib.et_epipolar_forward(ctypes.byref(d), *[null] * 12) != 0 and msg in lib.et_last_error()
        assert lib.et_epipolar_backward(ctypes.byref(d), *[null] * 10, ctypes.c_size_t(0), null) != 0 and msg in lib.et_last_error()
        assert lib.et_sample_locs(ctypes.byref(d), *[null] * 6) != 0 and msg in lib.et_last_error()
        assert int(lib.et_epipolar_forward_workspace_bytes(ctypes.byref(d))) == 0          # no tile path either
    # the GEMM form of the residual fusion: only the 256-channel head, no NULL operands, aligned weight buffer
    assert lib.et_residual_gemm(ctypes.c_int64(64), 128, *[null] * 6) != 0 and b"256-channel" in lib.et_last_error()
    assert lib.et_residual_gemm(ctypes.c_int64(0), 256, *[null] * 6) != 0 and b"bad sizes" in lib.et_last_error()
    assert lib.et_residual_gemm(ctypes.c_int64(64), 256, *[null] * 6) != 0 and b"NULL" in lib.et_last_error()
    assert lib.et_residual_gemm_pack(null, null, null) != 0 and b"NULL" in lib.et_last_error()
    assert int(lib.et_residual_gemm_packed_bytes()) == 16 * 8 * 2 * 64 * 16 + 256


def test_cpu_tensors_are_rejected_not_silently_computed():
    spec = ops.LayerSpec(H=8, W=8, K=8)
    x = torch.zeros(1, 8, 8, 8)
    with pytest.raises(_lib.EpipolarAmdError):
        ops.forward_nhwc(spec, x, x, torch.zeros(1, 27))


@pytest.mark.parametrize("case", golden_cases())
def test_camera_algebra_equals_reference_loop(oracle_mod, case):
    d = load_golden(case)
    P1, P2 = torch.from_numpy(d["P1"]), torch.from_numpy(d["P2"])
    cam = camera.pair_algebra(P1, P2).numpy()
    a, b, c = oracle_mod.camera_algebra(P1, P2)      # per-matrix pinverse loop, as epipolar.py:336
    assert np.array_equal(cam[:, :12], a.reshape(-1, 12))
    assert np.array_equal(cam[:, 12:24], b.reshape(-1, 12))
    assert np.array_equal(cam[:, 24:], c, equal_nan=True)       # (rectified / identical rigs: the epipole is inf / nan, epipolar.py:348)
    # against the algebra frozen in the fixture (another CPU's LAPACK may differ in the last bits)
    assert np.allclose(cam, d["cam"], rtol=1e-4, atol=1e-7, equal_nan=True)


def _spec_from_golden(d, **kw):
    m = d["dims"]
    return ops.LayerSpec(H=m["H"], W=m["W"], K=m["K"], downsample=float(d["downsample"]),
                         correct_normalize=m["correct"], softmax_scale=float(d["softmax_scale"]),
                         softmax_enabled=m["softmax"], **kw)


@pytest.mark.parametrize("case", golden_cases())
def test_device_setup_code_reproduces_reference_sample_locs(lib, case):
    d = load_golden(case)
    spec = _spec_from_golden(d)
    m = d["dims"]
    cam = d["cam"]
    desc = spec.desc(m["N"], m["C"])
    K, W = m["K"], m["W"]
    taps = np.zeros((K, 4), np.int32)
    wts = np.zeros((K, 4), np.float32)
    locs = np.zeros((K, 2), np.float32)
    xs, ys, steps = spec.xs.numpy(), spec.ys.numpy(), spec.steps.numpy()
    fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(0)
    for n in range(m["N"]):
        camn = np.ascontiguousarray(cam[n])
        for ri, h in enumerate(d["rows"]):
            for w in rng.choice(W, size=min(W, 6), replace=False):
                rc = lib.et_debug_host_sample_setup(ctypes.byref(desc), fp(xs), fp(ys), fp(steps), fp(camn),
                                                    int(h), int(w), fp(taps), fp(wts), fp(locs))
                assert rc == 0, lib.et_last_error()
                want = d["sample_locs"][:, n, ri, w]                 # (K,2) from the real reference
                assert np.array_equal(locs, want), (n, h, w)
                # taps / weights against the textbook bilinear rule on those locations
                gx = (want[:, 0] + 1) * np.float32(W / 2) - np.float32(0.5)
                gy = (want[:, 1] + 1) * np.float32(m["H"] / 2) - np.float32(0.5)
                x0, y0 = np.floor(gx), np.floor(gy)
                fx, fy = gx - x0, gy - y0
                for k in range(K):
                    seen = {}
                    for r in range(4):
                        if taps[k, r] >= 0:
                            ty, tx = divmod(int(taps[k, r]), W)
                            assert (ty & 1, tx & 1) == (r >> 1, r & 1)            # parity routing
                            seen[(tx - int(x0[k]), ty - int(y0[k]))] = wts[k, r]
                        else:
                            assert wts[k, r] == 0
                    for (dx, dy), wv in seen.items():
                        assert dx in (0, 1) and dy in (0, 1)
                        wx = fx[k] if dx else np.float32(1) - fx[k]
                        wy = fy[k] if dy else np.float32(1) - fy[k]
                        assert wv == np.float32(wy * wx)
                    # every in-image tap of the 2x2 footprint is present
                    for dx in (0, 1):
                        for dy in (0, 1):
                            xx, yy = int(x0[k]) + dx, int(y0[k]) + dy
                            if 0 <= xx < W and 0 <= yy < m["H"] and abs(x0[k]) < 1e6:
                                assert (dx, dy) in seen


def test_layerspec_constants_match_reference_constructor():
    # epipolar.py:35-54 for the headline config: 4*x+1.5 grid, inclusive torch.range steps
    spec = ops.LayerSpec(H=64, W=64, K=64)
    assert spec.xs[0] == 1.5 and spec.xs[-1] == 253.5 and spec.ys[3] == 13.5
    assert spec.steps[0] == 0 and spec.steps[-1] == 1 and len(spec.steps) == 64
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = torch.range(0, 1, 1.0 / 63)
    assert torch.equal(ref, spec.steps)


def test_tile_path_eligibility_is_decided_on_the_host(lib):
    """et_epipolar_forward_workspace_bytes / et_epipolar_backward_tiled_workspace_bytes run on the host: they say
    which shapes take the MFMA tile kernels (C == 256, H*W <= 16384, one pixel's rows 4*min(K, max(W,H)) within the
    256-, 384- or 512-row tile array) and size the per-pair pixel-order scratch."""
    import ctypes

    from epipolar_transformers_amd import ops

    def sizes(H, W, K, C, N=3, variant=0):
        d = ops.LayerSpec(H=H, W=W, K=K, variant=variant).desc(N, C)
        return (int(lib.et_epipolar_forward_workspace_bytes(ctypes.byref(d))),
                int(lib.et_epipolar_backward_tiled_workspace_bytes(ctypes.byref(d))))

    def ws_bytes(tiles, pairs=3, hw=None):
        # pixel order (32 per tile) | overflow counter + sticky error word (64 words) | overflow list | statistics | scales |
        # segments
        # (header of 64 words first: [0] overflow count, [1] sticky error word; one float4 base line per tile last)
        # ... and the segments by pixel (tile_keys_kernel: the per-pixel half of the ordering), one float4 per pixel
        hw = (tiles // pairs) * 32 if hw is None else hw
        words = 64 + tiles * 32 + 2 * tiles + 4 * pairs + 4 + 4 * tiles * 32 + 4 * tiles + 4 * pairs * hw
        return words * 4 + 256

    fwd, bwd = sizes(64, 64, 64, 256)                       # configs[1]
    assert fwd == bwd == ws_bytes(3 * 128)
    assert sizes(64, 64, 64, 256, variant=_lib.ET_VARIANT_TILE_CLASSIC)[0] == ws_bytes(3 * 128)
    d_ns = ops.LayerSpec(H=64, W=64, K=64, softmax_enabled=False).desc(3, 256)
    assert int(lib.et_epipolar_forward_workspace_bytes(ctypes.byref(d_ns))) == ws_bytes(3 * 128)   # soft-max off: exact-fp32 tiles
    assert sizes(96, 96, 64, 256)[0] == ws_bytes(3 * 288)           # config 4: 384-row tiles
    assert sizes(10, 10, 16, 256)[0] == ws_bytes(3 * 4, hw=100)     # 100 pixels -> 4 padded tiles
    d = ops.LayerSpec(H=64, W=64, K=64).desc(3, 256)
    assert int(lib.et_epipolar_forward_workspace_stats_offset(ctypes.byref(d))) == (64 + 3 * 128 * 32 + 3 * 128) * 4
    # the sticky error word sits in the header: the same offset whatever the shape (a cached workspace is reused across shapes)
    for shape in ((64, 64, 64), (96, 96, 64), (16, 16, 16)):
        dd = ops.LayerSpec(H=shape[0], W=shape[1], K=shape[2]).desc(5, 256)
        assert int(lib.et_epipolar_forward_workspace_error_offset(ctypes.byref(dd))) == 4
    assert sizes(64, 64, 64, 128) == (0, 0)                 # other channel counts: per-pixel kernels
    f5, b5 = sizes(128, 128, 128, 256)
    assert f5 == b5 == ws_bytes(3 * 512)                    # config 5: 512-row tiles, K = 128 (two samples per lane)
    assert sizes(129, 128, 16, 256) == (0, 0)               # more than 16384 pixels per pair
    f, b = sizes(32, 32, 128, 256)
    assert f > 0 and b == f                                 # K > 64: both tile kernels (K <= 256)
    assert sizes(16, 16, 16, 256, variant=32768)[0] > 0 and sizes(64, 64, 64, 256, variant=32768) == (0, 0)


def test_library_is_built_without_slp_vectorisation():
    """The packed-fp32 instructions SLP vectorisation generates made the one-block-per-tile forward return wrong attention
    intermittently (scripts/dev/README.md): the flag must not get lost."""
    from epipolar_transformers_amd import build

    assert "-fno-slp-vectorize" in build.flags()
    assert "-ffp-contract=off" in build.flags()          # (one rounding per reference op: the parity tests rely on it)
    # ... and cannot be undone from the environment
    import os
    keep = os.environ.get("ET_EXTRA_HIPCC_FLAGS")
    try:
        for bad in ("-fslp-vectorize", "-mllvm -vectorize-slp=true", "-ffast-math", "-ffp-contract=fast"):
            os.environ["ET_EXTRA_HIPCC_FLAGS"] = bad
            with pytest.raises(RuntimeError):
                build.flags()
        os.environ["ET_EXTRA_HIPCC_FLAGS"] = "-DET_WS_PROFILE=2"
        assert "-DET_WS_PROFILE=2" in build.flags()
    finally:
        if keep is None:
            os.environ.pop("ET_EXTRA_HIPCC_FLAGS", None)
        else:
            os.environ["ET_EXTRA_HIPCC_FLAGS"] = keep


def test_outputs_are_poisoned_under_test():
    from epipolar_transformers_amd import ops

    assert ops.POISON_OUTPUTS, "tests/conftest.py must switch NaN-poisoning of the output buffers on"
    t = ops._empty((4, 3), device="cpu")
    assert bool(torch.isnan(t).all())
