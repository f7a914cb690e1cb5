#!/bin/bash
# rocprofv3 kernel table of config C3 (SpectralUNET-1650 @608x700, batch 1) in one precision mode.  usage: tools/prof_c3.sh [mode] [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}; MODE=${1:-bf16}; OUT=$R/gpurun_out/${2:-c3_$MODE}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HPRI_PRECISION=$MODE
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/run_config.py spectral 1 3 > $OUT/run.log 2> $OUT/prof.err || exit 2
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
tail -1 $OUT/run.log
