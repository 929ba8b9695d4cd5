#!/bin/bash
# Round profiles on the GPU box (via gpurun): every number DESIGN.md / README.md quote comes from a file this writes.
# usage: gpu_profiles.sh TAG      -> gpurun_out/<TAG>_*   (copy what is to be judged into profiles/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
TAG=${1:-r06}
export HSA_ENABLE_IPC_MODE_LEGACY=0
# ONLY="bench stats pmcbwd" gpu_profiles.sh TAG   runs those sections only (keys: bench two classic perpixel config4 config5
#   summary stats shapestats epilogue pmcfwd pmcfused roles e2e pmcbwd general micro)
want() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }
stats() {  # stats NAME -- bench args...   : rocprofv3 --kernel-trace --stats of a bench run, keep the kernel_stats csv
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_$name" -o trace -- \
      python "$ROOT/bench.py" "$@" > "$OUT/${TAG}_${name}_rocprof.log" 2>&1)
  local f=$(find "$OUT/prof_${TAG}_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${name}_kernel_stats.csv" && head -6 "$f"
  rm -rf "$OUT/prof_${TAG}_$name"
}
want bench && { echo "== bench (configs[1])"; timeout 900 python bench.py --steps 30 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; tail -c 600 "$OUT/${TAG}_bench.json"; echo; }
want two && { echo "== bench, the step as two kernels (sample+attention, then residual GEMM: rounds 1-3)"; timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-configs --two-kernels > "$OUT/${TAG}_bench_two_kernels.json" 2>> "$OUT/${TAG}_bench.err"; }
want classic && { echo "== bench classic (one-block-per-tile kernel, split-fp16 GEMMs)"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --variant 65536 > "$OUT/${TAG}_bench_variant65536_classic.json" 2>> "$OUT/${TAG}_bench.err"; }
want perpixel && { echo "== bench per-pixel"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --variant 16384 > "$OUT/${TAG}_bench_variant16384_perpixel.json" 2>> "$OUT/${TAG}_bench.err"; }
want config4 && { echo "== bench config4 head (96x96, K=64, 128 pairs)"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --hw 96 > "$OUT/${TAG}_bench_config4.json" 2>> "$OUT/${TAG}_bench.err"; }
want config5 && { echo "== bench config5 share (128x128, K=128, 8 views x 8 frames = 64 pairs)"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --samples 128 --hw 128 --frames 8 --views 8 > "$OUT/${TAG}_bench_config5.json" 2>> "$OUT/${TAG}_bench.err"; }
want summary && python - <<PY
import json
for n in ("bench", "bench_two_kernels", "bench_variant65536_classic", "bench_variant16384_perpixel", "bench_config4", "bench_config5"):
    try:
        r = json.load(open("$OUT/${TAG}_%s.json" % n))
        print("%-36s step %.3f ms  fwd %.3f ms  bwd %.3f ms  %.0f pair-views/s" % (n, r["ms_per_step"], r["extra"]["fused_kernel_fwd_ms"], r["extra"]["fused_kernel_bwd_ms"], r["value"]), r["extra"].get("end_to_end", ""))
    except Exception as e:
        print(n, "failed", e)
PY
want stats && {
echo "== rocprof kernel stats"
stats bench --steps 10 --warmup 3 --no-cpu-baseline
stats config4 --steps 5 --warmup 2 --no-cpu-baseline --hw 96
stats config5 --steps 5 --warmup 2 --no-cpu-baseline --samples 128 --hw 128 --frames 8 --views 8
}
# one rocprofv3 --kernel-trace --stats file PER (kernel family, shape): scripts/profile_kernel.py launches exactly one call size, so
# AverageNs in these files is the time of that shape (the stats of a whole bench run mix the call sizes of every config it times)
shape_stats() {  # shape_stats NAME  (PROF_* in the environment)
  local name=$1
  (cd /tmp && export TMPDIR=/tmp && PROF_REPS=60 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_$name" -o trace -- \
      python "$ROOT/scripts/profile_kernel.py" > "$OUT/${TAG}_${name}_rocprof.log" 2>&1)
  local f=$(find "$OUT/prof_${TAG}_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_shape_${name}_kernel_stats.csv" && head -4 "$f"
  rm -rf "$OUT/prof_${TAG}_$name"
}
want shapestats && {
echo "== rocprof kernel stats, one shape per file"
PROF_KERNEL=fused shape_stats config2_layer
PROF_KERNEL=fwd shape_stats config2_forward
PROF_KERNEL=bwd shape_stats config2_backward
PROF_KERNEL=fused PROF_HW=96 shape_stats config4_layer
PROF_KERNEL=fwd PROF_HW=96 shape_stats config4_forward
PROF_KERNEL=bwd PROF_HW=96 shape_stats config4_backward
PROF_KERNEL=fwd PROF_HW=128 PROF_K=128 PROF_PAIRS=64 PROF_VIEWS=8 shape_stats config5_forward
PROF_KERNEL=bwd PROF_HW=128 PROF_K=128 PROF_PAIRS=64 PROF_VIEWS=8 shape_stats config5_backward
}
want epilogue && {
echo "== residual epilogue kernel"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_epi" -o trace -- python - <<PY > "$OUT/${TAG}_epilogue_rocprof.log" 2>&1
import sys, torch
sys.path.insert(0, "$ROOT")
from epipolar_transformers_amd import ops
shape = (128, 64, 64, 256)
feat, out, y = (torch.randn(shape, device="cuda") for _ in range(3))
sc, sh = torch.randn(256, device="cuda"), torch.randn(256, device="cuda")
for _ in range(10):
    ops.residual_epilogue(feat, out, y, sc, sh, want_finalout=False, want_x=True)     # 4 tensors x 537 MB
    ops.residual_epilogue(feat, out, want_finalout=False, want_x=True)                 # 3 tensors
torch.cuda.synchronize()
PY
)
f=$(find "$OUT/prof_${TAG}_epi" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_epilogue_kernel_stats.csv" && head -4 "$f"; rm -rf "$OUT/prof_${TAG}_epi"
}
want pmcfwd && { echo "== PMC forward"; bash scripts/gpu_pmc.sh "${TAG}_fwd_tile" 0 fwd | tail -34; }
want pmcfused && { echo "== PMC one-kernel layer"; bash scripts/gpu_pmc.sh "${TAG}_fwd_fused" 0 fused | tail -34; }
want roles && { echo "== role experiments of the warp-specialised forward"; [ -f epipolar_transformers_amd/lib/libepipolar_amd_prof.so ] && EPIPOLAR_AMD_LIB=$ROOT/epipolar_transformers_amd/lib/libepipolar_amd_prof.so timeout 300 python scripts/ws_experiment.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/${TAG}_ws_role_experiment.txt"; }
want e2e && { echo "== end to end at the larger image sizes"; (timeout 400 python scripts/e2e_shapes.py --image 384 --frames 32 --views 4; timeout 400 python scripts/e2e_shapes.py --image 512 --frames 8 --views 8 --samples 128; timeout 400 python scripts/e2e_shapes.py --image 384 --frames 8 --views 4 --body epipolarposeR-152) 2>&1 | grep -v amdgpu.ids | tee "$OUT/${TAG}_e2e_shapes.txt"; }
want pmcbwd && { echo "== PMC backward"; bash scripts/gpu_pmc.sh "${TAG}_bwd_tile" 0 bwd | tail -34; }
want general && { echo "== parameterised + pooled head through the general kernel"; timeout 300 python scripts/general_mode_time.py 2>/dev/null | tail -1 | tee "$OUT/${TAG}_general_mode_time.txt"; }
want micro && {
echo "== microbenchmarks"
for m in mfma_valu_overlap mfma_valu_samewave load_patterns mfma_lds_rates; do
  [ -x scripts/micro/$m ] && timeout 120 scripts/micro/$m > "$OUT/${TAG}_micro_$m.txt" 2>&1 && tail -3 "$OUT/${TAG}_micro_$m.txt"
done
}
