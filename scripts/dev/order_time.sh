# development: rocprofv3 average of tile_order_kernel and the forward kernels over scripts/fwd_ab.py
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_order -o trace -- python "$ROOT/scripts/fwd_ab.py" x > /tmp/order.log 2>&1)
python - <<P
import csv, glob
fs = glob.glob("/tmp/prof_order/**/*kernel_stats.csv", recursive=True)
if not fs:
    print(open("/tmp/order.log").read()[-2000:])
for r in csv.DictReader(open(fs[0])):
    if "order" in r["Name"] or "keys" in r["Name"] or "fwd_tile" in r["Name"]:
        print(r["Name"][:70], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "min %.1f us" % (float(r["MinNs"]) / 1e3))
P
