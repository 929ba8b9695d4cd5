ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"; TAG=r03
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --samples 128 --hw 128 --frames 8 --views 8 > "$OUT/${TAG}_bench_config5.json" 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_config5" -o trace -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --samples 128 --hw 128 --frames 8 --views 8 > "$OUT/${TAG}_config5_rocprof.log" 2>&1)
f=$(find "$OUT/prof_${TAG}_config5" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_config5_kernel_stats.csv" && head -3 "$f" | cut -c1-200; rm -rf "$OUT/prof_${TAG}_config5"
python -c "
import json; r=json.load(open('$OUT/${TAG}_bench_config5.json')); print('config5 step %.3f fwd %.3f exact %.3f frac %.3f bwd %.3f value %.0f' % (r['ms_per_step'], r['extra']['fused_kernel_fwd_ms'], r['roofline']['exact_fp32']['kernel_ms'], r['roofline']['frac'], r['extra']['fused_kernel_bwd_ms'], r['value']))"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -2
