#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== full pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/r3c6_pytest.log" 2>&1; echo "exit $?"; tail -15 "$OUT/r3c6_pytest.log" | cut -c1-300
for cfgname in "config4 --hw 96" "config5 --samples 128 --hw 128 --frames 8 --views 8" "config2"; do
  set -- $cfgname; name=$1; shift
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end "$@" > "$OUT/r3c6_bench_$name.json" 2> "$OUT/r3c6_bench_$name.err"
  python - <<PY
import json
try:
    r = json.load(open("$OUT/r3c6_bench_$name.json"))
    x = r["roofline"].get("exact_fp32", {})
    print("%-8s step %.3f ms  fwd %.3f ms (frac %.3f)  exact-fp32 fwd %.3f ms  bwd %.3f ms  rgemm %.3f" % ("$name", r["ms_per_step"], r["extra"]["fused_kernel_fwd_ms"], r["roofline"]["frac"], x.get("kernel_ms", -1), r["extra"]["fused_kernel_bwd_ms"], r["extra"].get("residual_gemm", {}).get("kernel_ms", -1)))
except Exception as e:
    print("$name failed", e); print(open("$OUT/r3c6_bench_$name.err").read()[-1500:])
PY
done
