#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== full gpu suite"; (time timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) 2>&1 | tee "$OUT/r06_final_gpu_tests.txt"
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a "$OUT/r06_final_gpu_tests.txt"
ONLY="shapestats" bash scripts/gpu_profiles.sh r06 2>&1 | grep "ws_kernel\|bwd_tile_kernel" | cut -c1-220
