#!/bin/bash
# build_variant.sh <name> <conv_fwd source> [extra hipcc flags...] -> experiments/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
OUT=experiments/build_$NAME; mkdir -p $OUT
C=hyperpri_amd/csrc
WG=${WGRAD_SRC:-$C/conv_wgrad.hip}
for f in api.cpp pack.hip bn.hip elementwise.hip step.hip ingest.hip; do
  [ -f $OUT/${f%.*}.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip -c $C/$f -I $C -o $OUT/${f%.*}.o &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip -c $SRC -I $C "$@" -o $OUT/conv_fwd.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip -c $WG -I $C "$@" -o $OUT/conv_wgrad.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o experiments/lib_$NAME.so $OUT/*.o
echo built experiments/lib_$NAME.so
