#!/bin/bash
# round 3, GPU call 1: new tests on the round-2 kernels + the role experiment of the WS forward
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest (new + changed tests)"; timeout 900 python -m pytest tests/test_gpu_split_fp16.py tests/test_gpu_modes.py -m gpu -q --timeout 600 > "$OUT/r3c1_pytest_new.log" 2>&1; echo "exit $?"; tail -40 "$OUT/r3c1_pytest_new.log"
echo "== ws experiment"; EPIPOLAR_AMD_LIB="$ROOT/epipolar_transformers_amd/lib/libepipolar_amd_prof.so" timeout 300 python scripts/ws_experiment.py > "$OUT/r3c1_ws_experiment.txt" 2>&1; cat "$OUT/r3c1_ws_experiment.txt"
echo "== ws profile"; EPIPOLAR_AMD_LIB="$ROOT/epipolar_transformers_amd/lib/libepipolar_amd_prof.so" WS_PROFILE_LIGHT=1 timeout 300 python scripts/ws_profile.py > "$OUT/r3c1_ws_profile.txt" 2>&1; cat "$OUT/r3c1_ws_profile.txt"
echo "== full pytest"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x --deselect tests/test_gpu_split_fp16.py > "$OUT/r3c1_pytest_all.log" 2>&1; echo "exit $?"; tail -8 "$OUT/r3c1_pytest_all.log"
