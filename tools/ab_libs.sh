#!/bin/bash
# Same-box A/B of whole-library variants (hyperpri_amd/lib/var_<name>.so, built in the container): each variant is copied over
# libhyperpri_hip.so in the box's scratch copy of the repo and the fp32 bench runs; interleaved rounds.
# usage: tools/ab_libs.sh "<names>" [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; L=$R/hyperpri_amd/lib
cp $L/libhyperpri_hip.so /tmp/lib_keep.so
# two untimed runs first: the first processes on a fresh box measure low
for w in 1 2; do HPRI_PRECISION=${PREC:-fp32} timeout -k 10 120 python $R/bench.py --steps 8 --warmup 2 --bf16-steps 0 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-training-shaped --no-configs > /dev/null 2>&1; done
for r in $(seq 1 ${2:-2}); do
  for v in $1; do
    cp $L/var_$v.so $L/libhyperpri_hip.so
    HPRI_PRECISION=${PREC:-fp32} timeout -k 10 120 python $R/bench.py --steps 8 --warmup 2 --bf16-steps 0 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-training-shaped --no-configs $3 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', b['value'], b['ms_per_step'], b.get('loss'))" || exit 1
  done
done
cp /tmp/lib_keep.so $L/libhyperpri_hip.so
