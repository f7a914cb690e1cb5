#!/bin/bash
# Diagnostic build of the bf16-plane convolution with in-kernel s_memtime stamps (never the shipped library).
set -e
cd "$(dirname "$0")/../hyperpri_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I."
for s in api.cpp conv_fwd.hip pack.hip; do /opt/rocm/bin/hipcc $F -x hip -c $s -o /tmp/st_${s%.*}.o & done
/opt/rocm/bin/hipcc $F -DHPRI_STAMPS -x hip -c conv_wino.hip -o /tmp/st_wino.o &
/opt/rocm/bin/hipcc $F -DHPRI_STAMPS -x hip -c conv_wino4.hip -o /tmp/st_wino4.o &
/opt/rocm/bin/hipcc $F -DHPRI_STAMPS -x hip -c conv_bf16v2.hip -o /tmp/st_v2.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libv2stamps.so /tmp/st_api.o /tmp/st_conv_fwd.o /tmp/st_pack.o /tmp/st_v2.o /tmp/st_wino.o /tmp/st_wino4.o
echo built ../lib/libv2stamps.so
