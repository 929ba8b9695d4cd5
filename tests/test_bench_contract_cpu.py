"""bench.py's bookkeeping on CPU: the algorithmic bytes/flops of SURVEY.md section 8d, the CLI defaults the
driver relies on, and the committed PMC record the `traffic` field is read from."""
import importlib.util
import json
import os
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_work_matches_survey_numbers():
    b = _bench()
    assert b.algorithmic_bytes_per_pair(256, 64, 64, 64) == 13_664_352          # SURVEY.md 8d "Numbers", Config 2
    assert b.algorithmic_flops_per_pair(256, 64, 64, 64) == 805_306_368
    assert b.algorithmic_bytes_per_pair(256, 96, 96, 64) == 30_744_672          # Config 4
    assert b.algorithmic_bytes_per_pair(256, 128, 128, 128) == 58_851_424       # Config 5
    assert b.algorithmic_flops_per_pair(256, 128, 128, 128) == 6_442_450_944


def test_cli_defaults_are_the_headline_workload(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.frames, a.views, a.hw, a.channels, a.samples, a.partition) == (1, 32, 4, 64, 256, 64, "frames")
    assert a.steps >= 10 and a.warmup >= 1
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)


def test_committed_pmc_record_feeds_traffic_field():
    b = _bench()
    path = os.path.join(ROOT, "profiles", "fwd_pmc_latest.json")
    assert os.path.exists(path), "profiles/fwd_pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass) is missing"
    m = json.load(open(path))
    t = b.measured_hbm_traffic(m["C"], m["H"], m["W"], m["K"], m["pairs"])
    algo = b.algorithmic_bytes_per_pair(m["C"], m["H"], m["W"], m["K"]) * m["pairs"]
    # a plausible record: at least the algorithmic bytes, at most twice them (round 4: 1.64 x -- reads 1.8 x, L2 hit rate 73 %;
    # DESIGN.md section 4.1 "Measured" says what was tried about it)
    assert t is not None and 0.9 * algo < t < 2.0 * algo
    pf = os.path.join(ROOT, "profiles", "fwd_fused_pmc_latest.json")
    assert os.path.exists(pf), "profiles/fwd_fused_pmc_latest.json (the one-kernel layer's PMC pass) is missing"
    mf = json.load(open(pf))
    tf = b.measured_hbm_traffic(mf["C"], mf["H"], mf["W"], mf["K"], mf["pairs"], one_kernel=True)
    assert tf is not None and mf["which"] == "fused" and 0.9 * algo < tf < 2.2 * algo
    assert b.measured_hbm_traffic(m["C"], m["H"], m["W"], m["K"] + 1, m["pairs"]) is None
