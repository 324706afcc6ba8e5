#!/bin/bash
# A/B of builds of libgpak_hip.so inside one box: trailing-update kernel alone (tools/time_gemm.py), then the whole step
# (tools/time_sizes.py): bash tools/ab_gemm.sh libA.so libB.so ...
for rep in 1 2; do for L in "$@"; do cp $L gp_ss_ak_amd/libgpak_hip.so; echo "== $L"; GEMM_SIZES=8192,32768 python tools/time_gemm.py random 2>&1 | grep "^Np"; python tools/time_sizes.py 8192 32768 2>&1 | grep "^N="; done; done
