#!/bin/bash
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wgrad -o trace -- python $R/scripts/dev/wgrad_time.py > /dev/null 2>&1 < /dev/null)
python - <<PY
import csv
for r in csv.DictReader(open('$R/gpurun_out/prof_wgrad/trace_kernel_stats.csv')):
    if 'wgrad' in r['Name']:
        print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,2), 'us')
PY
