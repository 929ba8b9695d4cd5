#!/bin/bash
# the forward's left-over list kernel: launch sizing (2 | 3 blocks per CU) and how a short list is shared (8 blocks per tile or none |
# 2 / 4 / 8 by the list's length): one-kernel layer call per rig, one box
L=$PWD/epipolar_transformers_amd/lib
for pass in 1 2; do
for rig in ring epipole_inside near_rectified_y h36m_room epipole_border; do
  EPIPOLAR_AMD_LIB=$L/libepipolar_amd_listold.so AB_FUSED=1 AB_RIG=$rig python scripts/fwd_ab.py "2/cu, 8-or-1" 2>&1 | grep "forward call" | cut -c1-150
  EPIPOLAR_AMD_LIB=$L/libepipolar_amd_list2adapt.so AB_FUSED=1 AB_RIG=$rig python scripts/fwd_ab.py "2/cu, adaptive" 2>&1 | grep "forward call" | cut -c1-150
  AB_FUSED=1 AB_RIG=$rig python scripts/fwd_ab.py "3/cu, adaptive" 2>&1 | grep "forward call" | cut -c1-150
done
done
