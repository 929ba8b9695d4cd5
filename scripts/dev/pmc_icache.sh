# development: instruction-cache counters of the forward kernels.  usage: pmc_icache.sh [fwd|fused]
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch\|INST_CACHE\|SQC_" | head -20
for which in fwd fused; do
  for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    PROF_KERNEL=$which timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/ic_$which/$(echo $c | cut -c1-6) -o pmc -- python $GRAFT_REPO_ROOT/scripts/profile_kernel.py > /tmp/ic.log 2>&1 || tail -3 /tmp/ic.log
  done
  python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/ic_$which/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ws_kernel" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$which", {k: round(sum(x)/len(x)) for k,x in agg.items()})
PY
done
