#!/bin/bash
# rocprofv3 kernel stats of the one-kernel layer call on the hard rigs (which launch costs what)
R=$PWD
for rig in ${RIGS:-epipole_inside near_rectified_y h36m_room}; do
  (cd /tmp && export TMPDIR=/tmp && PROF_RIG=$rig PROF_KERNEL=fused PROF_REPS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv \
     -d $R/gpurun_out/prof_fwd_$rig -o trace -- python $R/scripts/profile_kernel.py > /dev/null 2>&1 < /dev/null)
  echo "== $rig"
  python - <<PY
import csv
for r in csv.DictReader(open('$R/gpurun_out/prof_fwd_$rig/trace_kernel_stats.csv')):
    if 'tile_' in r['Name'] or 'epipolar' in r['Name'] or 'residual' in r['Name']:
        print(r['Name'][:75].ljust(75), r['Calls'], round(float(r['AverageNs'])/1e3,2), 'us')
PY
done
