#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$ROOT/epipolar_transformers_amd/lib
for rep in 1 2; do
 for rig in ring epipole_inside h36m_room near_rectified_y; do
  for lib in c5 new; do
    f=$L/libepipolar_amd_$lib.so; [ $lib = new ] && f=$L/libepipolar_amd.so
    EPIPOLAR_AMD_LIB=$f AB_RIG=$rig timeout 200 python scripts/bwd_ab.py "$lib" 2>&1 | grep "backward call\|Error\|error" | tee -a "$OUT/r06_c6_bwd_ab.txt"
  done
 done
done
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_rigs.py tests/test_gpu_parity.py -m gpu -x -q -k "backward or bwd" 2>&1 | tail -4 | tee "$OUT/r06_c6_tests.txt"
