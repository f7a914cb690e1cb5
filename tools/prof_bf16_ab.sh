R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03x; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HPRI_PRECISION=bf16
A="--steps 5 --warmup 2 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-roofline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/two -- python3 $R/bench.py $A > $OUT/two.json 2> $OUT/two.err || exit 2
export HPRI_SIDE_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one_on -- python3 $R/bench.py $A > $OUT/one_on.json 2> $OUT/one_on.err || exit 3
export HPRI_FUSE_BN_REDUCE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one_off -- python3 $R/bench.py $A > $OUT/one_off.json 2> $OUT/one_off.err || exit 4
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
