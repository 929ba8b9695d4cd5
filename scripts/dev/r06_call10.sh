#!/bin/bash
# the round's profile collection, part 1: role counters + phase cycles of the CURRENT kernel, shape-clean kernel stats, PMC passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$ROOT/epipolar_transformers_amd/lib
echo "== phase cycles"; EPIPOLAR_AMD_LIB=$L/libepipolar_amd_prof.so WS_PROFILE_LIGHT=1 python scripts/ws_profile.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/r06_ws_phase_cycles.txt"
echo "== role timing"; EPIPOLAR_AMD_LIB=$L/libepipolar_amd_prof.so timeout 300 python scripts/ws_experiment.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/r06_ws_role_experiment.txt"
echo "== role counters"
(cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  EPIPOLAR_AMD_LIB=$L/libepipolar_amd_prof.so timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_roles_g$i" -o pmc -- python "$ROOT/scripts/ws_pmc_roles.py" > "$OUT/r06_roles_g$i.log" 2>&1
  python "$ROOT/scripts/ws_pmc_roles.py" --summarise "$OUT/pmc_roles_g$i" | tee -a "$OUT/r06_ws_role_counters.txt"
  rm -rf "$OUT/pmc_roles_g$i"
done)
ONLY="shapestats pmcfwd pmcfused pmcbwd" bash scripts/gpu_profiles.sh r06 2>&1 | tail -120
