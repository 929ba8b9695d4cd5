#!/bin/bash
# A/B of the backward's over-capacity policy: old (search + split in place, one-array 256-row second launch) | listonly (old policy,
# merged 288-column second launch) | new (beyond 256 over-capacity tiles: deferred at once; merged 288-column second launch)
L=$PWD/epipolar_transformers_amd/lib
for pass in 1 2; do
for rig in ring epipole_inside h36m_room near_rectified_y epipole_border; do
  EPIPOLAR_AMD_LIB=$L/libepipolar_amd_oldbwd.so AB_RIG=$rig python scripts/bwd_ab.py old 2>&1 | grep "backward call"
  EPIPOLAR_AMD_LIB=$L/libepipolar_amd_listonly.so AB_RIG=$rig python scripts/bwd_ab.py listonly 2>&1 | grep "backward call"
  AB_RIG=$rig python scripts/bwd_ab.py new 2>&1 | grep "backward call"
done
done
