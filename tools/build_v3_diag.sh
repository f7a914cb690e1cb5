#!/bin/bash
# diagnostic builds of conv_bf16v3.hip with stamps and parts of the main loop removed (V3_DIAG bits: 1 no weight DMA inside the loop,
# 2 no halo DMA inside the loop, 4 fragments read once per stage, 8 no waits / barriers inside the loop): lib/libv3diag<bits>.so.
# Outputs are wrong by construction for bits != 0; only cycles matter.
set -e
cd "$(dirname "$0")/../hyperpri_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I. -DHPRI_DIAG_KERNELS"
/opt/rocm/bin/hipcc $F -x hip -c api.cpp -o /tmp/v3d_api.o &
/opt/rocm/bin/hipcc $F -x hip -c conv_fwd.hip -o /tmp/v3d_fwd.o &
for d in ${DIAGS:-0 1 2 3 4 8 15}; do
  /opt/rocm/bin/hipcc $F -DHPRI_STAMPS -DHPRI_DIAG_KERNELS -DV3_DIAG=$d -x hip -c conv_bf16v3.hip -o /tmp/v3d_$d.o &
done
wait
for d in ${DIAGS:-0 1 2 3 4 8 15}; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libv3diag$d.so /tmp/v3d_api.o /tmp/v3d_fwd.o /tmp/v3d_$d.o; done
echo built
