#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$ROOT/epipolar_transformers_amd/lib
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_rigs.py tests/test_gpu_band.py -m gpu -x -q 2>&1 | tail -3 | tee "$OUT/r06_c3_tests.txt"
for rep in 1 2; do
  for lib in r05 copyB new copyA_prio copyB_prio; do
    f=$L/libepipolar_amd_$lib.so; [ $lib = new ] && f=$L/libepipolar_amd.so
    EPIPOLAR_AMD_LIB=$f AB_FUSED=1 timeout 200 python scripts/fwd_ab.py "fused [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c3_ab.txt"
    EPIPOLAR_AMD_LIB=$f timeout 200 python scripts/fwd_ab.py "sample+attention [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c3_ab.txt"
  done
done
