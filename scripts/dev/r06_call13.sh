#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$ROOT/epipolar_transformers_amd/lib
for rep in 1 2; do
  for lib in c13 new; do
    f=$L/libepipolar_amd_$lib.so; [ $lib = new ] && f=$L/libepipolar_amd.so
    EPIPOLAR_AMD_LIB=$f AB_FUSED=1 timeout 200 python scripts/fwd_ab.py "fused [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c13_ab.txt"
    EPIPOLAR_AMD_LIB=$f timeout 200 python scripts/fwd_ab.py "sample+attention [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c13_ab.txt"
  done
done
for lib in c13 new; do
f=$L/libepipolar_amd_$lib.so; [ $lib = new ] && f=$L/libepipolar_amd.so
EPIPOLAR_AMD_LIB=$f AB_FUSED=1 AB_HW=96 timeout 200 python scripts/fwd_ab.py "fused 96 [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c13_ab.txt"
EPIPOLAR_AMD_LIB=$f AB_HW=128 AB_K=128 AB_PAIRS=64 AB_VIEWS=8 timeout 300 python scripts/fwd_ab.py "config5 [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c13_ab.txt"
done
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_rigs.py tests/test_gpu_band.py tests/test_gpu_two_pass.py -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/r06_c13_tests.txt"
