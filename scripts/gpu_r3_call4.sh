#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest split + parity (-x)"; timeout 900 python -m pytest tests/test_gpu_split_fp16.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x > "$OUT/r3c4_pytest.log" 2>&1; echo "exit $?"; tail -30 "$OUT/r3c4_pytest.log"
echo "== timing"; timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee "$OUT/r3c4_timing.txt"
import sys, torch
sys.path.insert(0, ".")
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn
dev = torch.device("cuda:0")
H, C, K = 64, 256, 64
P1, P2 = syn.make_pairs(32, 4, H * 4, seed=1000, jitter=(0.05, 8.0))
g = torch.Generator(device=dev).manual_seed(0)
ref = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
src = torch.randn(128, H, H, C, device=dev, generator=g).relu_()
cam = camera.pair_algebra(P1, P2).to(dev)
def events_ms(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(t) / len(t), t[0]
for name, v in (("ws2 default", 0), ("ws v1", _lib.ET_VARIANT_WS_V2), ("classic fp32", _lib.ET_VARIANT_TILE_CLASSIC)):
    spec = ops.LayerSpec(H=H, W=H, K=K, variant=v)
    m, lo = events_ms(lambda: ops.forward_nhwc(spec, ref, src, cam))
    print("%-16s forward call %.3f ms (min %.3f)" % (name, m, lo), flush=True)
ops.check_tile_errors()
o2 = ops.forward_nhwc(ops.LayerSpec(H=H, W=H, K=K), ref, src, cam)
o1 = ops.forward_nhwc(ops.LayerSpec(H=H, W=H, K=K, variant=_lib.ET_VARIANT_NO_TILE), ref, src, cam)
print("ws2 vs per-pixel: out %.2e attn %.2e" % ((o2[0]-o1[0]).abs().max().item(), (o2[1]-o1[1]).abs().max().item()))
PY
echo "== ws profile (ws2, full stamps)"; EPIPOLAR_AMD_LIB="$ROOT/epipolar_transformers_amd/lib/libepipolar_amd_prof.so" timeout 300 python scripts/ws_profile.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/r3c4_ws2_profile.txt"
echo "== rocprof kernel times"; (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_r3c4" -o trace -- python "$ROOT/scripts/profile_kernel.py" > "$OUT/r3c4_rocprof.log" 2>&1); F=$(find "$OUT/prof_r3c4" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -8 "$F" | cut -c1-200; find "$OUT/prof_r3c4" -name "*kernel_trace.csv" -delete
