#!/bin/bash
# the radix sort of the tile ordering against the bitonic network: same permutation (bit for bit), and the step
L=$PWD/epipolar_transformers_amd/lib
EPIPOLAR_AMD_LIB=$L/libepipolar_amd_bitonic.so python scripts/dev/order_perm_dump.py gpurun_out/perm_bitonic.npz 2>&1 | tail -1
python scripts/dev/order_perm_dump.py gpurun_out/perm_radix.npz 2>&1 | tail -1
python scripts/dev/order_perm_dump.py --compare gpurun_out/perm_bitonic.npz gpurun_out/perm_radix.npz
python -m pytest tests/test_gpu_order.py -q -m gpu 2>&1 | tail -2
for pass in 1 2; do
EPIPOLAR_AMD_LIB=$L/libepipolar_amd_bitonic.so python scripts/fwd_ab.py bitonic 2>&1 | grep -v Warn | tail -2
python scripts/fwd_ab.py radix 2>&1 | grep -v Warn | tail -2
done
