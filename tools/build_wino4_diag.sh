#!/bin/bash
# diagnostic builds of conv_wino4.hip with stamps and parts of the main loop removed (WINO_DIAG bits: 4 no weight DMA inside the
# loop, 8 no halo DMA inside the loop): lib/libwino4diag<bits>.so.  Outputs are wrong by construction; only cycles matter.
set -e
cd "$(dirname "$0")/../hyperpri_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I."
/opt/rocm/bin/hipcc $F -x hip -c api.cpp -o /tmp/w4d_api.o
/opt/rocm/bin/hipcc $F -x hip -c conv_wino.hip -o /tmp/w4d_wino.o
for d in ${DIAGS:-0 4 8 12}; do
  /opt/rocm/bin/hipcc $F -DHPRI_STAMPS -DWINO_DIAG=$d -x hip -c conv_wino4.hip -o /tmp/w4d_$d.o &
done
wait
for d in ${DIAGS:-0 4 8 12}; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libwino4diag$d.so /tmp/w4d_api.o /tmp/w4d_$d.o /tmp/w4d_wino.o; done
echo built
