import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # every output buffer of the HIP wrappers starts as NaN under test: a kernel that leaves part of its output unwritten
    # must not be able to hide behind the (correct) values a recycled allocation still holds from an earlier identical call
    from epipolar_transformers_amd import ops
    ops.POISON_OUTPUTS = True


def golden_cases():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if n not in ("mpjpe_scene", "triangulation", "model_r18") and not n.startswith("hourglass_")]   # (fixtures with their own tests)


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    H, W, C, K, N, image, correct, softmax, _ = [int(v) for v in d["meta"]]
    d["dims"] = dict(H=H, W=W, C=C, K=K, N=N, image=image, correct=bool(correct), softmax=bool(softmax))
    return d


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc

    orc.build()
    return orc


def denormalized_locations(locs, H, W, correct_normalize=True):
    """de_normalize (vision/multiview.py:39-57) of sample locations (..., 2) in float32, the way the kernels and the
    reference round it."""
    locs = np.asarray(locs, dtype=np.float32)
    one, two, half = np.float32(1), np.float32(2), np.float32(0.5)
    if correct_normalize:
        return np.stack([(locs[..., 0] + one) * np.float32(W - 1) / two, (locs[..., 1] + one) * np.float32(H - 1) / two], -1)
    return np.stack([(locs[..., 0] + one) * np.float32(W) / two - half, (locs[..., 1] + one) * np.float32(H) / two - half], -1)


def assert_corr_pos(locs, corr, want_corr, weights, correct_normalize=True, tie=2e-6, max_frac=2e-2):
    """corr_pos is the de-normalised location of the arg-max sample (epipolar.py:237-242): exact, except where the
    arg-max has a TIE that float rounding resolves differently.  Every pixel whose corr_pos differs from the expected one
    must be such a tie: the sample the reference picked carries (to `tie`) the same weight in OUR `weights` (the tensor
    the arg-max ran over: attention / similarity, (N,Ks,H,W)) as the sample we picked -- no blanket mismatch budget.
    `locs`: (K,N,H,W,2) sample locations (bit-equal to the reference's); only the first Ks are candidates (POOLING
    indexes the un-pooled list with the pooled index, epipolar.py:237-241).  Returns the boolean map of tie pixels."""
    corr, want_corr, weights = np.asarray(corr), np.asarray(want_corr), np.asarray(weights)
    neq = (corr != want_corr).any(-1)                                  # (N,H,W)
    if not neq.any():
        return neq
    K, N, H, W, _ = locs.shape
    den = denormalized_locations(locs, H, W, correct_normalize)[: weights.shape[1]]
    for n, h, w in zip(*np.nonzero(neq)):
        cand = den[:, n, h, w]                                         # (Ks,2)
        k_ref = np.nonzero((cand == want_corr[n, h, w]).all(-1))[0]
        k_our = np.nonzero((cand == corr[n, h, w]).all(-1))[0]
        assert len(k_ref) and len(k_our), "corr_pos is not one of the pixel's sample locations at %s" % ((n, h, w),)
        a = weights[n, :, h, w]
        assert abs(float(a[k_ref[0]]) - float(a[k_our[0]])) <= tie * max(1.0, abs(float(a[k_ref[0]]))), \
            "corr_pos differs at %s and it is not a tie: w[k_ref=%d]=%g w[k_ours=%d]=%g" % (
                (n, h, w), k_ref[0], a[k_ref[0]], k_our[0], a[k_our[0]])
    assert neq.mean() <= max_frac, "suspiciously many arg-max ties: %g" % neq.mean()
    return neq
