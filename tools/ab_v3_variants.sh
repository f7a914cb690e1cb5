#!/bin/bash
# Same-box A/B of conv_bf16v3 build variants: each lib/var_<name>.so (built by tools/build_v3_variants.sh) is copied over
# libhyperpri_hip.so in the box's scratch copy and tools/v3_bench.py runs on a few layers.  usage: tools/ab_v3_variants.sh "<names>"
R=${GRAFT_REPO_ROOT:-/root/repo}; L=$R/hyperpri_amd/lib
cp $L/libhyperpri_hip.so /tmp/lib_keep.so
for v in $1; do
  cp $L/var_$v.so $L/libhyperpri_hip.so
  echo "== variant $v"
  timeout -k 10 200 python $R/tools/v3_bench.py 2>/dev/null | cut -c1-200 || exit 1
done
cp /tmp/lib_keep.so $L/libhyperpri_hip.so
