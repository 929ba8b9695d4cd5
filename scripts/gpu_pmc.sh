#!/bin/bash
# PMC counter passes for the fused forward kernel (one rocprofv3 run per counter
# group; --pmc is never combined with sys/runtime tracing).  usage: gpu_pmc.sh TAG VARIANT [fwd|bwd]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
TAG=${1:-r01}; export PROF_VARIANT=${2:-0}; export PROF_KERNEL=${3:-fwd}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "VALUBusy MfmaUtil LDSBankConflict OccupancyPercent"
do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_${TAG}/g$i" -o pmc -- python "$ROOT/scripts/profile_kernel.py" > "$OUT/pmc_${TAG}_g$i.log" 2>&1
  echo "group $i ($grp): exit $?"
done
python - <<PY
import csv, glob, collections, os, re
out = "$OUT/pmc_${TAG}"
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(epipolar_\w+|tile_\w+_kernel)(<[^>]*>)?", row.get("Kernel_Name", ""))
        if not m:
            continue
        k = (m.group(0), row["Counter_Name"])          # per kernel (template arguments included)
        agg.setdefault(k, []).append(float(row["Counter_Value"]))
with open(out + "_summary.txt", "w") as fh:
    for (kn, cn), vals in agg.items():
        line = "%-62s %-32s n=%d mean=%.6g" % (kn, cn, len(vals), sum(vals) / len(vals))
        print(line); fh.write(line + "\n")
import json
want_bwd = os.environ.get("PROF_KERNEL", "fwd") == "bwd"                     # (the backward run launches one forward for the attention)
main = ([kn for kn, _ in agg if want_bwd and "bwd" in kn] + [kn for kn, _ in agg if "ws_kernel" in kn] +
        [kn for kn, _ in agg if "bwd_tile" in kn or "fwd_tile_kernel" in kn] +
        [kn for kn, _ in agg if kn.startswith("epipolar")] + [None])[0]          # the dominant kernel of the run
get = lambda name: next((sum(v) / len(v) for (kn, cn), v in agg.items() if cn == name and kn == main), None)
if get("FETCH_SIZE") is not None and get("WRITE_SIZE") is not None:
    json.dump({"C": 256, "H": int(os.environ.get("PROF_HW", 64)), "W": int(os.environ.get("PROF_HW", 64)),
               "K": int(os.environ.get("PROF_K", 64)), "pairs": 128, "kernel": main,
               "variant": int(os.environ.get("PROF_VARIANT", 0)), "which": os.environ.get("PROF_KERNEL", "fwd"), "FETCH_SIZE_KB": get("FETCH_SIZE"),
               "WRITE_SIZE_KB": get("WRITE_SIZE"), "TCC_HIT_sum": get("TCC_HIT_sum"), "TCC_MISS_sum": get("TCC_MISS_sum"),
               "VALUBusy": get("VALUBusy"), "SQ_INSTS_VALU": get("SQ_INSTS_VALU")},
              open(out + "_pmc.json", "w"), indent=1)
PY
