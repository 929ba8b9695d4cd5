#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$ROOT/epipolar_transformers_amd/lib
for rep in 1 2; do
  for lib in new g1w2 g1w4; do
    f=$L/libepipolar_amd_$lib.so; [ $lib = new ] && f=$L/libepipolar_amd.so
    EPIPOLAR_AMD_LIB=$f AB_FUSED=1 timeout 200 python scripts/fwd_ab.py "fused [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c12_ab.txt"
    EPIPOLAR_AMD_LIB=$f timeout 200 python scripts/fwd_ab.py "sample+attention [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c12_ab.txt"
  done
done
