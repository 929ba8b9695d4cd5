#!/bin/bash
# round 6, call 1: the vector-wave G3 -- correctness of the fused layer, A/B against the round-5 library, role counters of round 5's kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$ROOT/epipolar_transformers_amd/lib
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_rigs.py tests/test_gpu_band.py -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/r06_c1_tests.txt"
echo "== A/B"
for rep in 1 2; do
  for lib in r05 new; do
    f=$L/libepipolar_amd.so; [ $lib = r05 ] && f=$L/libepipolar_amd_r05.so
    EPIPOLAR_AMD_LIB=$f AB_FUSED=1 timeout 200 python scripts/fwd_ab.py "fused [$lib]" 2>&1 | grep "forward call" | tee -a "$OUT/r06_c1_ab.txt"
    EPIPOLAR_AMD_LIB=$f timeout 200 python scripts/fwd_ab.py "sample+attention [$lib]" 2>&1 | grep "forward call" | tee -a "$OUT/r06_c1_ab.txt"
  done
done
EPIPOLAR_AMD_LIB=$L/libepipolar_amd_r05.so AB_FUSED=1 AB_HW=96 timeout 200 python scripts/fwd_ab.py "fused 96 [r05]" 2>&1 | grep "forward call" | tee -a "$OUT/r06_c1_ab.txt"
AB_FUSED=1 AB_HW=96 timeout 200 python scripts/fwd_ab.py "fused 96 [new]" 2>&1 | grep "forward call" | tee -a "$OUT/r06_c1_ab.txt"
echo "== role counters (round-5 kernel, profiling build)"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  EPIPOLAR_AMD_LIB=$L/libepipolar_amd_prof.so timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_roles_g$i" -o pmc -- python "$ROOT/scripts/ws_pmc_roles.py" > "$OUT/r06_c1_roles_g$i.log" 2>&1
  python "$ROOT/scripts/ws_pmc_roles.py" --summarise "$OUT/pmc_roles_g$i" | tee -a "$OUT/r06_c1_roles.txt"
  rm -rf "$OUT/pmc_roles_g$i"
done
