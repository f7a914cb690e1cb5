#!/bin/bash
# The round's closing measurement on one GPU box, in the driver's order.  Three parts, one gpurun call each (a call is limited to 20 min):
#   tools/final_measure.sh <tag> 1   full GPU test suite, smoke, bench.py + its profiled / counter passes (tools/measure.sh), the 238->64
#                                    layer's counter passes;  then on the build host:  python tools/pmc_traffic.py gpurun_out/<tag> <tag>
#                                    and  python tools/pmc_traffic.py gpurun_out/<tag>fc <tag>_first_conv  (they write profiles/<tag>_*pmc*.json
#                                    with the library stamp, which bench.py replays only for that build)
#   tools/final_measure.sh <tag> 2   C3 counter passes, bf16-mode kernel tables, determinism, Dice parity (fp32, bf16, f16), bench.py once more (now
#                                    with the replayed counter figures: the line to commit as profiles/<tag>_bench.json)
#   tools/final_measure.sh <tag> 3   stock DDP on one rank, predict path with / without the fused first layer, the force-sync bench line, C3 kernel table
# Outputs under gpurun_out/<tag>*.  gpurun MERGES into the local gpurun_out/: remove gpurun_out/<tag> <tag>fc <tag>bf16 before a re-run, or
# pick the newest file of a directory (ls -t) when copying a summary into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04}; PART=${2:-1}; G=$R/gpurun_out
cd $R
if [ "$PART" = 1 ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -q > $G/${TAG}_gputests_final.log 2>&1; echo "gpu tests rc $?"; tail -1 $G/${TAG}_gputests_final.log
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $G/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -1 $G/${TAG}_smoke.log
  bash tools/measure.sh $TAG > $G/${TAG}_measure.log 2>&1; echo "measure rc $?"
  bash tools/pmc_cmd.sh ${TAG}fc tools/first_conv.py 10 bf16_out_only > $G/${TAG}_fc_pmc.log 2>&1; echo "first-conv pmc rc $?"
elif [ "$PART" = 2 ]; then
  # (the C3 counter passes first, summarised here as well: bench.py below replays profiles/<tag>_c3_bf16_pmc_traffic.json for its C3 line;
  #  run the same pmc_traffic.py command on the build host afterwards for the committed copy)
  bash tools/pmc_c3.sh bf16 ${TAG}_c3_bf16 > /dev/null 2>&1; echo "c3 pmc rc $?"
  python tools/pmc_traffic.py gpurun_out/${TAG}_c3_bf16 ${TAG}_c3_bf16 > $G/${TAG}_c3_pmc_summary.log 2>&1; echo "c3 pmc summary rc $?"
  bash tools/prof_bf16_ab.sh ${TAG}bf16 > $G/${TAG}_bf16_prof.log 2>&1; echo "bf16 profile rc $?"
  timeout -k 10 300 python tools/determinism_check.py fp32 6 > $G/${TAG}_determinism.txt 2>&1
  timeout -k 10 300 python tools/determinism_check.py bf16 6 >> $G/${TAG}_determinism.txt 2>&1; echo "determinism rc $?"; grep -v amdgpu $G/${TAG}_determinism.txt | tail -2
  timeout -k 10 400 python tools/dice_parity.py 14 20 ${TAG} fp32 > $G/${TAG}_dice_fp32.log 2>&1; echo "dice fp32 rc $?"; tail -1 $G/${TAG}_dice_fp32.log | cut -c1-300
  timeout -k 10 400 python tools/dice_parity.py 14 20 ${TAG}_bf16 bf16 > $G/${TAG}_dice_bf16.log 2>&1; echo "dice bf16 rc $?"; tail -1 $G/${TAG}_dice_bf16.log | cut -c1-300
  timeout -k 10 300 python tools/dice_parity.py 14 20 ${TAG}_f16 f16 > $G/${TAG}_dice_f16.log 2>&1; echo "dice f16 rc $?"; tail -1 $G/${TAG}_dice_f16.log | cut -c1-300
  timeout -k 10 500 python bench.py > $G/${TAG}_bench_final.json 2> $G/${TAG}_bench_final.err; echo "bench rc $?"
else
  timeout -k 10 300 python tools/ddp_stock_bench.py > $G/${TAG}_ddp_stock_one_rank.json 2> $G/${TAG}_ddp_stock.err; echo "stock ddp rc $?"
  timeout -k 10 200 python tools/predict_bench.py 4 > $G/${TAG}_predict_ingest.json 2> $G/${TAG}_predict.err; echo "predict rc $?"
  for k in bf16 f16; do timeout -k 10 120 python tools/ingest_conv_bench.py $k 2>/dev/null; done > $G/${TAG}_ingest_conv.jsonl; echo "ingest conv rc $?"
  timeout -k 10 300 python bench.py --force-sync --stock-ddp --no-configs --no-cpu-baseline > $G/${TAG}_force_sync_stock.json 2> $G/${TAG}_force_sync_stock.err; echo "force-sync bench rc $?"
  bash tools/prof_c3.sh bf16 ${TAG}_c3_bf16_final > /dev/null 2>&1; echo "c3 profile rc $?"
  bash tools/pmc_cmd.sh ${TAG}ig tools/ingest_conv_bench.py bf16 > $G/${TAG}_ig_pmc.log 2>&1; echo "ingest conv pmc rc $?"    # then: python tools/pmc_traffic.py gpurun_out/${TAG}ig ${TAG}_ingest_conv
fi
