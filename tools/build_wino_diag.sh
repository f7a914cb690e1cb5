#!/bin/bash
# diagnostic builds of conv_wino.hip with stamps and parts of the main loop removed (WINO_DIAG bits: 1 no halo reads,
# 2 no weight reads, 4 no DMA inside the loop): lib/libwinodiag<bits>.so.  Outputs are wrong by construction; only cycle shares matter.
set -e
cd "$(dirname "$0")/../hyperpri_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I."
/opt/rocm/bin/hipcc $F -x hip -c api.cpp -o /tmp/wd_api.o
for d in 0 1 2 4 7; do
  /opt/rocm/bin/hipcc $F -DHPRI_STAMPS -DWINO_DIAG=$d -x hip -c conv_wino.hip -o /tmp/wd_$d.o &
done
wait
for d in 0 1 2 4 7; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libwinodiag$d.so /tmp/wd_api.o /tmp/wd_$d.o; done
echo built
