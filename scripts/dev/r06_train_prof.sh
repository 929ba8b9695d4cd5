#!/bin/bash
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-other-configs --no-cpu-baseline --no-end-to-end > /dev/null 2>&1 < /dev/null)
python - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/prof_train/trace_kernel_stats.csv')))
for r in rows[:24]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs'])/1e3)).rjust(9), 'us')
PY
