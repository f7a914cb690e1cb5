#!/bin/bash
# Same-box A/B of environment switches: the bench's main loop in the given precision, interleaved rounds.
# usage: PREC=bf16 tools/ab_env.sh "NAME=a NAME=b,OTHER=c" [rounds] [extra bench args]   (a comma joins several variables of one variant)
# Two untimed runs first: the first processes on a fresh box measure up to 30 % low (bf16: 119, 133, 155, 153, then 169 cubes/s).
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in 1 2; do HPRI_PRECISION=${PREC:-fp32} timeout -k 10 120 python $R/bench.py --steps 8 --warmup 2 --bf16-steps 0 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-training-shaped --no-configs > /dev/null 2>&1; done
for r in $(seq 1 ${2:-2}); do
  for v in $1; do
    env ${v//,/ } HPRI_PRECISION=${PREC:-fp32} timeout -k 10 120 python $R/bench.py --steps 8 --warmup 2 --bf16-steps 0 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-training-shaped --no-configs $3 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', b['value'], b['ms_per_step'], b.get('loss'))" || exit 1
  done
done
