"""Build the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m epipolar_transformers_amd.build [--report]

The .so lands in epipolar_transformers_amd/lib/ (git-ignored, shipped to the
GPU box by gpurun).  -ffp-contract=off is REQUIRED: the geometry in
csrc/epipolar_geometry.h reproduces the reference's float32 roundings op by op
and spells its FMAs explicitly.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = [os.path.join(PKG, "csrc", "epipolar_kernels.hip")]
DEPS = SRC + [os.path.join(PKG, "csrc", f) for f in ("epipolar_geometry.h", "kernels_forward.inc",
                                                      "kernels_forward_tile.inc", "kernels_backward.inc", "kernels_backward_tile.inc",
                                                      "kernels_misc.inc")] + \
    [os.path.join(ROOT, "include", "epipolar_amd.h")]
LIB = os.path.join(PKG, "lib", "libepipolar_amd.so")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def flags():
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_library(force: bool = False, report: bool = False) -> str:
    if not (force or report or needs_build()):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc()] + flags() + (["-Rpass-analysis=kernel-resource-usage"] if report else []) + ["-o", LIB] + SRC
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        raise RuntimeError("hipcc failed (exit %d)" % proc.returncode)
    if report:
        print(resource_table(proc.stdout))
    return LIB


def resource_table(log: str) -> str:
    rows, cur = [], None
    for line in log.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" ")[0]] = val
    out = ["%-78s %5s %5s %7s %4s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS")]
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        out.append("%-78s %5s %5s %7s %4s %6s" % (name[:78], r.get("VGPRs"), r.get("TotalSGPRs"),
                                                 r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
    return "\n".join(out)


if __name__ == "__main__":
    build_library(force=True, report="--report" in sys.argv)
    print(LIB)
