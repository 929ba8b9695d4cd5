EPIPOLAR_AMD_LIB=epipolar_transformers_amd/lib/libepipolar_amd_prof1.so python scripts/dev/slp_repro.py "with SLP vectorisation   " 2>&1 | grep -v amdgpu.ids
EPIPOLAR_AMD_LIB=epipolar_transformers_amd/lib/libepipolar_amd_prof.so python scripts/dev/slp_repro.py "-fno-slp-vectorize       " 2>&1 | grep -v amdgpu.ids
EPIPOLAR_AMD_LIB=epipolar_transformers_amd/lib/libepipolar_amd_prof1.so python scripts/dev/slp_repro.py "with SLP vectorisation   " 2>&1 | grep -v amdgpu.ids
EPIPOLAR_AMD_LIB=epipolar_transformers_amd/lib/libepipolar_amd_prof.so python scripts/dev/slp_repro.py "-fno-slp-vectorize       " 2>&1 | grep -v amdgpu.ids
