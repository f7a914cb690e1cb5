#!/bin/bash
# Per-precision-mode kernel stats + SQ PMC pass:  bash tools/measure_mode.sh bf16x3 tag
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-bf16x3}
TAG=${2:-mode_$MODE}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HPRI_PRECISION=$MODE
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-roofline --no-configs > $OUT/bench.json 2> $OUT/prof.err || exit 2
# counter passes on ONE stream (counters are chip-wide per dispatch window)
export HPRI_SIDE_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-roofline --no-configs > /dev/null 2> $OUT/pmc_sq.err || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py --steps 2 --warmup 1 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-roofline --no-configs > /dev/null 2> $OUT/pmc_lds.err || echo "lds pmc pass failed"
# HBM traffic of the mode's kernels: FETCH_SIZE and WRITE_SIZE in separate passes (tools/pmc_traffic.py <dir> <tag> parses them)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-roofline --no-configs > /dev/null 2> $OUT/pmc_fetch.err || echo "fetch pmc pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --bf16-steps 0 --no-cpu-baseline --no-optimizer-leg --no-training-shaped --no-roofline --no-configs > /dev/null 2> $OUT/pmc_write.err || echo "write pmc pass failed"
find $OUT -name '*kernel_trace.csv' -size +20M -delete
tail -1 $OUT/bench.json | cut -c1-300
