#!/bin/bash
# whole-library variants that differ in conv_bf16v3.hip's compile-time switches: lib/var_<name>.so for tools/ab_v3_variants.sh
# usage: tools/build_v3_variants.sh "name1:-DFLAG1 name2:-DFLAG2,-DFLAG3"
set -e
cd "$(dirname "$0")/../hyperpri_amd"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Icsrc"
OBJS=$(ls lib/*.o | grep -v conv_bf16v3.o | grep -v _diag.o)
for spec in $1; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  [ "$flags" = "$name" ] && flags=""
  /opt/rocm/bin/hipcc $F $flags -x hip -c csrc/conv_bf16v3.hip -o /tmp/v3var_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/var_$name.so $OBJS /tmp/v3var_$name.o
done
echo built
