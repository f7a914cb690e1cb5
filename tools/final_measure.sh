#!/bin/bash
# The round's closing measurement on one GPU box, in the driver's order: full GPU test suite, smoke, bench.py (fresh process), then the
# profiled / counter passes of the same command (tools/measure.sh), the 238->64 layer's counter passes, the bf16-mode kernel tables,
# determinism and Dice-parity runs.  usage: tools/final_measure.sh <tag>      (outputs under gpurun_out/<tag>*)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04}; G=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $G/${TAG}_gputests_final.log 2>&1; echo "gpu tests rc $?"; tail -2 $G/${TAG}_gputests_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $G/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -1 $G/${TAG}_smoke.log
bash tools/measure.sh $TAG > $G/${TAG}_measure.log 2>&1; echo "measure rc $?"
bash tools/pmc_cmd.sh ${TAG}fc tools/first_conv.py 10 bf16_out_only > $G/${TAG}_fc_pmc.log 2>&1; echo "first-conv pmc rc $?"
bash tools/prof_bf16_ab.sh ${TAG}bf16 > $G/${TAG}_bf16_prof.log 2>&1; echo "bf16 profile rc $?"
timeout -k 10 300 python tools/determinism_check.py fp32 6 > $G/${TAG}_determinism.txt 2>&1; timeout -k 10 300 python tools/determinism_check.py bf16 6 >> $G/${TAG}_determinism.txt 2>&1; echo "determinism rc $?"; tail -2 $G/${TAG}_determinism.txt
timeout -k 10 900 python tools/dice_parity.py 14 20 ${TAG} fp32 > $G/${TAG}_dice_fp32.log 2>&1; echo "dice fp32 rc $?"; tail -2 $G/${TAG}_dice_fp32.log
timeout -k 10 900 python tools/dice_parity.py 14 20 ${TAG}_bf16 bf16 > $G/${TAG}_dice_bf16.log 2>&1; echo "dice bf16 rc $?"; tail -2 $G/${TAG}_dice_bf16.log
