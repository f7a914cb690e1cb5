#!/bin/bash
# Round measurement on the GPU box: bench line, rocprofv3 kernel stats of the same command, and two PMC passes
# (FETCH_SIZE, WRITE_SIZE -- separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Outputs under gpurun_out/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT   # NB: gpurun merges into the local gpurun_out/ -- remove the local copy of the tag dir before a re-run
cd /tmp && export TMPDIR=/tmp
cp $R/hyperpri_amd/lib/libhyperpri_hip.so.stamp $OUT/lib_stamp.txt    # which build the counters belong to (bench.py replays them only for that build)
timeout -k 10 500 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
tail -1 $OUT/bench.json | cut -c1-600
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-configs > $OUT/bench_prof.json 2> $OUT/prof.err || exit 2
# the same with every kernel on ONE stream (HPRI_SIDE_STREAM=0): per-kernel durations without a concurrent weight gradient, as the
# HIP-event pass of bench.py's roofline leg measures them
HPRI_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- python3 $R/bench.py --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-configs > /dev/null 2> $OUT/prof1.err || exit 2
# counter passes on ONE stream: a dispatch's counters are chip-wide, a concurrent kernel on the second stream would be counted in
export HPRI_SIDE_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-configs --no-roofline > /dev/null 2> $OUT/pmc_fetch.err || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-configs --no-roofline > /dev/null 2> $OUT/pmc_write.err || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-configs --no-roofline > /dev/null 2> $OUT/pmc_sq.err || exit 5
# keep only the per-kernel summaries small enough to merge back
find $OUT -name '*kernel_trace.csv' -size +20M -delete
ls -la $OUT $OUT/*/* | head -40
