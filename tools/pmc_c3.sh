#!/bin/bash
# HBM traffic of config C3's kernels in one precision mode: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, one stream.
# usage: tools/pmc_c3.sh [mode] [tag]     (summarise with tools/pmc_c3_summary.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; MODE=${1:-bf16}; OUT=$R/gpurun_out/${2:-c3_pmc_$MODE}; rm -rf $OUT; mkdir -p $OUT
cp $R/hyperpri_amd/lib/libhyperpri_hip.so.stamp $OUT/lib_stamp.txt
cd /tmp && export TMPDIR=/tmp
export HPRI_PRECISION=$MODE HPRI_SIDE_STREAM=0
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/run_config.py spectral 1 1 > $OUT/fetch.log 2> $OUT/fetch.err || exit 2
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/run_config.py spectral 1 1 > $OUT/write.log 2> $OUT/write.err || exit 3
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
ls $OUT
