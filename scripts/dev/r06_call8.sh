#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$ROOT/epipolar_transformers_amd/lib
for rep in 1 2; do
  for lib in c6 g3ring2 g3ring3 new g3ring6; do
    f=$L/libepipolar_amd_$lib.so; [ $lib = new ] && f=$L/libepipolar_amd.so
    EPIPOLAR_AMD_LIB=$f AB_FUSED=1 timeout 200 python scripts/fwd_ab.py "fused [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c8_ab.txt"
  done
done
for lib in c6 new; do
f=$L/libepipolar_amd_$lib.so; [ $lib = new ] && f=$L/libepipolar_amd.so
EPIPOLAR_AMD_LIB=$f AB_FUSED=1 AB_HW=96 timeout 200 python scripts/fwd_ab.py "fused 96 [$lib]" 2>&1 | grep "forward call\|Error\|error" | tee -a "$OUT/r06_c8_ab.txt"
done
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_rigs.py -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/r06_c8_tests.txt"
for i in 1 2 3 4 5; do timeout 300 python -m pytest "tests/test_gpu_rigs.py::test_backward_forms_vs_oracle_on_rig" -m gpu -q -k "epipole_inside and 96x96" 2>&1 | tail -2 | tee -a "$OUT/r06_c8_flaky.txt"; done
