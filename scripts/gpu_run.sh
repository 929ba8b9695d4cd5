#!/bin/bash
# Runs on the GPU box (via gpurun).  usage: gpu_run.sh TAG step [step ...]
#   steps: check (scripts/tile_check.py) | smoke | pytest | bench | rocprof | pmc | pmcbwd
# Everything lands in gpurun_out/ which gpurun merges back.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
TAG=${1:-r02}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
for step in "$@"; do
  case $step in
    check)
      echo "== tile_check"; timeout 600 python scripts/tile_check.py > "$OUT/check_$TAG.log" 2>&1; echo "check exit $?"; tail -60 "$OUT/check_$TAG.log";;
    smoke)
      echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke_$TAG.log" 2>&1; echo "smoke exit $?"; tail -3 "$OUT/smoke_$TAG.log";;
    pytest)
      echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu_$TAG.log" 2>&1; echo "pytest exit $?"; tail -30 "$OUT/pytest_gpu_$TAG.log";;
    bench)
      echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 5 > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"; echo "bench exit $?"
      cat "$OUT/bench_$TAG.json"; tail -5 "$OUT/bench_$TAG.err";;
    rocprof)
      echo "== rocprof"
      (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -o trace -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/rocprof_$TAG.log" 2>&1; echo "rocprof exit $?")
      F=$(find "$OUT/prof_$TAG" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -8 "$F"
      find "$OUT/prof_$TAG" -name "*kernel_trace.csv" -size +20M -delete;;
    pmc) bash scripts/gpu_pmc.sh "$TAG" 0 fwd | tail -40;;
    pmcbwd) bash scripts/gpu_pmc.sh "${TAG}_bwd" 0 bwd | tail -40;;
    *) echo "unknown step $step";;
  esac
done
