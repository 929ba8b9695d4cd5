#!/bin/bash
# duration of the ordering kernels inside 20 forward calls (Config 2; PROF_HW for other maps)
R=$PWD
(cd /tmp && export TMPDIR=/tmp && PROF_KERNEL=fwd PROF_REPS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv \
   -d $R/gpurun_out/prof_order -o trace -- python $R/scripts/profile_kernel.py > /dev/null 2>&1 < /dev/null)
python - <<PY
import csv
for r in csv.DictReader(open('$R/gpurun_out/prof_order/trace_kernel_stats.csv')):
    if 'tile_' in r['Name'] or 'epipolar' in r['Name']:
        print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,2), 'us', r['MinNs'])
PY
