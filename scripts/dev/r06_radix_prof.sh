#!/bin/bash
R=$PWD
L=$PWD/epipolar_transformers_amd/lib
for v in bitonic radix; do
  lib=$L/libepipolar_amd.so; [ $v = bitonic ] && lib=$L/libepipolar_amd_bitonic.so
  (cd /tmp && export TMPDIR=/tmp && EPIPOLAR_AMD_LIB=$lib PROF_KERNEL=fwd PROF_REPS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv \
     -d $R/gpurun_out/prof_order_$v -o trace -- python $R/scripts/profile_kernel.py > /dev/null 2>&1 < /dev/null)
  f=$(find $R/gpurun_out/prof_order_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  [ -n "$f" ] && head -5 "$f" | cut -c1-100,180-260
done
