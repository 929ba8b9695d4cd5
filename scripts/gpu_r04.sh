#!/bin/bash
# Round-4 GPU session helper (via gpurun).  usage: gpu_r04.sh TAG step [step ...]
#   steps: check | wsprof | pytest | pytestnew | bench | convprobe | rocprof | pmc | pmcbwd
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
TAG=${1:-r04}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
for step in "$@"; do
  case $step in
    check)
      echo "== tile_check"; timeout 600 python scripts/tile_check.py > "$OUT/check_$TAG.log" 2>&1; echo "check exit $?"; grep -v amdgpu.ids "$OUT/check_$TAG.log" | tail -70;;
    wsprof)
      echo "== ws_profile"; EPIPOLAR_AMD_LIB=$ROOT/epipolar_transformers_amd/lib/libepipolar_amd_prof.so timeout 300 python scripts/ws_profile.py > "$OUT/wsprof_$TAG.txt" 2>&1; echo "wsprof exit $?"; grep -v amdgpu.ids "$OUT/wsprof_$TAG.txt" | tail -12;;
    pytest)
      echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > "$OUT/pytest_gpu_$TAG.log" 2>&1; echo "pytest exit $?"; tail -40 "$OUT/pytest_gpu_$TAG.log";;
    pytestnew)
      echo "== pytest gpu (new tests)"; timeout 1500 python -m pytest tests/test_gpu_split_fp16.py tests/test_gpu_rccl.py -m gpu -q --timeout 900 > "$OUT/pytest_new_$TAG.log" 2>&1; echo "pytest exit $?"; tail -60 "$OUT/pytest_new_$TAG.log";;
    bench)
      echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 5 > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"; echo "bench exit $?"
      cat "$OUT/bench_$TAG.json"; tail -5 "$OUT/bench_$TAG.err";;
    convprobe)
      echo "== conv probe"; timeout 900 python scripts/conv_probe.py --image 384 --batch ${CONV_BATCH:-32} --settings ${CONV_SETTINGS:-cl0,cl1,nchw0,nchw1} > "$OUT/convprobe_$TAG.txt" 2>&1; echo "convprobe exit $?"; grep -v amdgpu.ids "$OUT/convprobe_$TAG.txt" | tail -50;;
    benchmodes)
      echo "== bench: host algebra overlapped / serial / graph"
      for m in "" "--serial-host" "--graph"; do
        timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-configs $m > "$OUT/benchmode_$TAG.json" 2>> "$OUT/bench_$TAG.err"
        python -c "import json;r=json.load(open('$OUT/benchmode_$TAG.json'));print('mode [$m]: ms_per_step %.4f  fwd call %.4f  residual gemm %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['extra']['residual_gemm']['kernel_ms']))"
      done;;
    hostprobe)
      echo "== host algebra"; timeout 300 python scripts/host_probe.py > "$OUT/hostprobe_$TAG.txt" 2>&1; grep -v amdgpu.ids "$OUT/hostprobe_$TAG.txt" | tail -30;;
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
