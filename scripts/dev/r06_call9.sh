#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$ROOT/epipolar_transformers_amd/lib
for lib in c6 new; do
  f=$L/libepipolar_amd_$lib.so; [ $lib = new ] && f=$L/libepipolar_amd.so
  EPIPOLAR_AMD_LIB=$f AB_HW=128 AB_K=128 AB_PAIRS=64 AB_VIEWS=8 timeout 300 python scripts/fwd_ab.py "config5 [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c9_ab.txt"
  EPIPOLAR_AMD_LIB=$f AB_HW=64 AB_K=128 AB_PAIRS=128 timeout 300 python scripts/fwd_ab.py "64x64 K=128 [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c9_ab.txt"
  EPIPOLAR_AMD_LIB=$f AB_HW=96 AB_K=100 AB_PAIRS=64 timeout 300 python scripts/fwd_ab.py "96x96 K=100 [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c9_ab.txt"
done
echo "== tests"; (time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) 2>&1 | tee "$OUT/r06_c9_tests.txt"
