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


def golden_cases():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if n != "mpjpe_scene"]           # the MPJPE scene has its own tests


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
