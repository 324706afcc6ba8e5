#!/bin/bash
# diagnostic build: potrf.hip with cycle-counter stamps at the phase boundaries of gpak_potrf128_f64
# (tools/time_potrf_phases.py reads them); `make -C gp_ss_ak_amd/csrc` afterwards restores the product build
set -e
cd "$(dirname "$0")/../gp_ss_ak_amd/csrc"
make -s -j6
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGPAK_POTRF_TIMING $GPAK_EXTRA -c potrf.hip -o potrf.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libgpak_hip.so *.o -lpthread
rm -f potrf.o
