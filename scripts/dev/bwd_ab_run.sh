L=epipolar_transformers_amd/lib
python scripts/bwd_ab.py base
for v in "$@"; do EPIPOLAR_AMD_LIB=$PWD/$L/libepipolar_amd_$v.so python scripts/bwd_ab.py $v; done
python scripts/bwd_ab.py base-again
