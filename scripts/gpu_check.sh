#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, bench, rocprof.
# Everything lands in gpurun_out/ which gpurun merges back.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
TAG=${1:-r01}
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showproductname 2>/dev/null | head -8 > "$OUT/gpu_info.txt"
nproc >> "$OUT/gpu_info.txt"
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke_$TAG.log" 2>&1; echo "smoke exit $?"
tail -3 "$OUT/smoke_$TAG.log"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu_$TAG.log" 2>&1; echo "pytest exit $?"
tail -25 "$OUT/pytest_gpu_$TAG.log"
echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 5 > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"; echo "bench exit $?"
cat "$OUT/bench_$TAG.json"; tail -5 "$OUT/bench_$TAG.err"
for v in 16384 256; do   # 16384: per-pixel default kernel (no MFMA tiles), 256: per-pixel baseline variant
  timeout 600 python bench.py --steps 10 --warmup 3 --variant $v --no-cpu-baseline > "$OUT/bench_${TAG}_variant$v.json" 2>> "$OUT/bench_$TAG.err"
  python - <<PY
import json
try:
    r=json.load(open("$OUT/bench_${TAG}_variant$v.json")); print("variant $v kernel_ms", r["roofline"]["kernel_ms"], "step ms", r["ms_per_step"])
except Exception as e: print("variant $v failed", e)
PY
done
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -o trace -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/rocprof_$TAG.log" 2>&1; echo "rocprof exit $?"
find "$OUT/prof_$TAG" -name "*kernel_stats*" | head -3
F=$(find "$OUT/prof_$TAG" -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -12 "$F"
# keep gpurun_out small: drop the raw per-dispatch trace if it is huge
find "$OUT/prof_$TAG" -name "*kernel_trace.csv" -size +20M -delete
