#!/bin/bash
# rocprofv3 kernel stats of the backward call on the hard rigs (which launch costs what)
R=$PWD
for rig in ${RIGS:-epipole_inside h36m_room}; do
  (cd /tmp && export TMPDIR=/tmp && PROF_RIG=$rig PROF_KERNEL=bwd PROF_REPS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv \
     -d $R/gpurun_out/prof_bwd_$rig -o trace -- python $R/scripts/profile_kernel.py > /dev/null 2>&1 < /dev/null)
  f=$(find $R/gpurun_out/prof_bwd_$rig -name "*kernel_stats.csv" | head -1)
  echo "== $rig $f"
  [ -n "$f" ] && head -8 "$f" | cut -c1-220
done
