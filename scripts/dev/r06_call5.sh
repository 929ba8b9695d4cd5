#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_modes.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/r06_c5_tests.txt"
echo "== bench"; (time timeout 1500 python bench.py > "$OUT/r06_c5_bench.json" 2> "$OUT/r06_c5_bench.err") 2>&1 | tail -4; tail -c 400 "$OUT/r06_c5_bench.err"
python - <<PY
import json
r = json.load(open("$OUT/r06_c5_bench.json"))
e = r["extra"]
print("step %.4f ms  value %.0f  roofline %.4f  s+a %.4f ms (%.3f)" % (r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["sample_attention_kernel"]["kernel_ms"], r["roofline"]["sample_attention_kernel"]["frac"]))
print("bwd", e["fused_kernel_bwd_stats_ms"], "deferred", e["fused_kernel_bwd_deferred_tiles"])
print("train", e["train_step"]["ms_per_step"])
for k in ("config4", "config5"):
    print(k, "fwd", e[k]["forward_stats_ms"], "bwd", e[k]["backward_stats_ms"], "deferred", e[k]["backward_deferred_tiles"], "layer", e[k]["layer_ms"])
for k, v in e["other_rigs"].items():
    print(k, "layer %.3f" % v["layer_ms"], "bwd", v["backward_stats_ms"], "deferred", v["backward_deferred_tiles"])
print("box", json.dumps(e["box"])[:1500])
print("cpu", json.dumps(r["cpu_baseline"])[:800])
print("e2e", e.get("end_to_end"))
PY
