#!/bin/bash
# A/B of two builds of libgpak_hip.so inside ONE gpurun box (step times differ by ~2 % between boxes, by 0.1 % inside
# one): bash tools/ab_libs.sh tools/bin/libA.so tools/bin/libB.so [N ...]
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    cp $L gp_ss_ak_amd/libgpak_hip.so
    echo "== $L"
    python tools/time_sizes.py "$@" 2>&1 | grep "^N="
  done
done
