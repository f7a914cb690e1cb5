set -o pipefail
python -m pytest tests/test_gpu_f16.py tests/test_gpu_ddp_stock.py tests/test_gpu_ddp_sink.py -q -m gpu -x 2>&1 | tail -15
echo "tests rc $?"
timeout -k 10 300 python bench.py --force-sync --stock-ddp --no-configs --no-cpu-baseline > gpurun_out/r05_force_sync_stock.json 2> gpurun_out/r05_force_sync_stock.err; echo "bench rc $?"
timeout -k 10 400 python tools/ddp_stock_bench.py > gpurun_out/r05_ddp_stock_one_rank.json 2> gpurun_out/r05_ddp_stock.err; echo "ddp bench rc $?"
python - <<'PY'
import json
b=json.loads(open("gpurun_out/r05_force_sync_stock.json").read().strip().splitlines()[-1])
print(b["value"], b["stock_ddp"])
r=json.load(open("gpurun_out/r05_ddp_stock_one_rank.json"))
for m,v in r["modes"].items(): print(m, v["median_ms"], v["stock_ddp_over_plain"], v["stock_ddp_bucket_view_over_plain"])
PY
