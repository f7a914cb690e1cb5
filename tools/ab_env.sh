#!/bin/bash
# A/B of one environment switch on the bench step, interleaved (A B A B) in fresh processes: value + ms per step of the fp32 and the bf16 mode.
# usage: tools/ab_env.sh VAR=value [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}; KV=$1; TAG=${2:-ab_env}; OUT=$R/gpurun_out/$TAG.txt; : > $OUT
A="--steps 20 --warmup 5 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-roofline --no-configs --bf16-steps 20"
for i in 1 2; do
  for arm in base $KV; do
    if [ "$arm" = base ]; then E=""; else E="$KV"; fi
    env $E timeout -k 10 200 python3 $R/bench.py $A 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$arm', 'fp32', d['value'], d['ms_per_step'], 'bf16', (d.get('bf16_mode') or {}).get('value'), 'bf16x3', (d.get('bf16x3_mode') or {}).get('value'))" >> $OUT || exit 2
  done
done
cat $OUT
