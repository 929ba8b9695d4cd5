#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== full pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/r3c5_pytest.log" 2>&1; echo "exit $?"; tail -25 "$OUT/r3c5_pytest.log"
echo "== tile_check"; timeout 600 python scripts/tile_check.py 2>&1 | grep -v amdgpu.ids | tail -22 | tee "$OUT/r3c5_tile_check.txt"
