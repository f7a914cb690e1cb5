#!/bin/bash
# rocprofv3 kernel tables of the bf16 step: two streams (default) and one stream.   usage: tools/prof_bf16_ab.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/${1:-r03x}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HPRI_PRECISION=bf16
A="--steps 5 --warmup 2 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-roofline --no-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/two -- python3 $R/bench.py $A > $OUT/two.json 2> $OUT/two.err || exit 2
export HPRI_SIDE_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one -- python3 $R/bench.py $A > $OUT/one.json 2> $OUT/one.err || exit 3
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
