#!/bin/bash
# PMC passes (SQ, LDS, HBM fetch/write) over an arbitrary python command:  bash tools/pmc_cmd.sh TAG tools/run_config.py spectral 1 1
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cp $R/hyperpri_amd/lib/libhyperpri_hip.so.stamp $OUT/lib_stamp.txt    # which build the counters belong to
cd /tmp && export TMPDIR=/tmp
ARGS=()
for a in "$@"; do if [ -e "$R/$a" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi; done
timeout -k 10 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 "${ARGS[@]}" > $OUT/sq.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_lds -- python3 "${ARGS[@]}" > $OUT/lds.log 2>&1 || exit 2
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 "${ARGS[@]}" > $OUT/fetch.log 2>&1 || exit 3
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 "${ARGS[@]}" > $OUT/write.log 2>&1 || exit 4
find $OUT -name '*kernel_trace.csv' -size +20M -delete
echo done
