#!/bin/bash
# Same-box A/B of two source trees (e.g. a worktree of an earlier commit under the repo, built in the container): the bench's
# main loop in the given precision, interleaved rounds.   usage: PREC=bf16 tools/ab_trees.sh "<dir1> <dir2>" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 ${2:-2}); do
  for v in $1; do
    HPRI_PRECISION=${PREC:-fp32} timeout -k 10 120 python $R/$v/bench.py --steps 8 --warmup 2 --bf16-steps 0 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-training-shaped --no-configs 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', b['value'], b['ms_per_step'], b.get('loss'))" || exit 1
  done
done
