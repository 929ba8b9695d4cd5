# development: FETCH_SIZE / WRITE_SIZE / L2 misses of the forward for library variants.  usage: pmc_ab.sh "" _variant ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  i=0
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    EPIPOLAR_AMD_LIB=$GRAFT_REPO_ROOT/epipolar_transformers_amd/lib/libepipolar_amd$v.so timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pm$v/g$i -o pmc -- python $GRAFT_REPO_ROOT/scripts/profile_kernel.py > /tmp/pm.log 2>&1
  done
  python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/pm$v/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ws_kernel" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("variant [$v]", {k: round(sum(x)/len(x)) for k,x in agg.items()})
PY
done
