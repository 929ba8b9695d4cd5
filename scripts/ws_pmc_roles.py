#!/usr/bin/env python
"""Dynamic instruction counts of the warp-specialised forward BY ROLE (development tool): the profiling build
(`python -m epipolar_transformers_amd.build --profile`) run once per experiment mask (scripts/ws_experiment.py explains the bits)
under `rocprofv3 --pmc ...`; the launches of the persistent kernel appear in the counter file in the order of MASKS below.

    cd /tmp && EPIPOLAR_AMD_LIB=$ROOT/epipolar_transformers_amd/lib/libepipolar_amd_prof.so \
        rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
        --output-format csv -d DIR -o pmc -- python scripts/ws_pmc_roles.py
    python scripts/ws_pmc_roles.py --summarise DIR
"""
import glob
import os
import sys

MASKS = (("everything", 0), ("no G1", 128), ("no G2", 64), ("vector waves only", 192), ("no SM", 8), ("S1 + S2 + copy only", 200),
         ("S1 + S2 only", 216), ("copy only", 232), ("skeleton", 248))
REPS = 2


def summarise(d):
    import csv
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        rows += [r for r in csv.DictReader(open(f)) if "fwd_tile_ws_kernel" in r.get("Kernel_Name", "")]
    by = {}
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)
    assert len(ids) == len(MASKS) * REPS, (len(ids), len(MASKS) * REPS)
    names = sorted({c for v in by.values() for c in v})
    print("%-24s" % "launch" + "".join("%22s" % n for n in names))
    for i, (name, _) in enumerate(MASKS):
        v = by[ids[i * REPS + REPS - 1]]
        print("%-24s" % name + "".join("%22.4g" % v.get(n, float("nan")) for n in names))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
        sys.exit(0)
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from epipolar_transformers_amd import camera, ops, synthetic as syn

    dev = torch.device("cuda:0")
    H, C, K = 64, 256, 64
    P1, P2 = syn.make_pairs(32, 4, H * 4, seed=1000, jitter=(0.05, 8.0))
    g = torch.Generator(device=dev).manual_seed(0)
    ref = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
    src = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
    cam = camera.pair_algebra(P1, P2).to(dev)
    spec = ops.LayerSpec(H=H, W=H, K=K)
    for name, bits in MASKS:
        os.environ["ET_WS_EXPERIMENT"] = str(bits)
        for _ in range(REPS):
            ops.forward_nhwc(spec, ref, src, cam)
        torch.cuda.synchronize()
