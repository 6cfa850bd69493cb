# one-rank RCCL dry run of the N > 1 path: CU reserve A/B and bf16 buckets (profiles/r03_rccl_dryrun.json)
common="--no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-entrypoint-figure --no-gemm-timer --steps 20 --warmup 5"
export MEMHIP_BENCH_FORCE_DIST=1
echo "[" > gpurun_out/r03_rccl_dryrun.json
first=1
for i in 1 2; do
for cfg in "--reserve-cus 0" "--reserve-cus 8" "--reserve-cus 16" "--reserve-cus 32" "--reserve-cus 0 --bucket-dtype bf16" "--reserve-cus 16 --bucket-dtype bf16"; do
  MASTER_PORT=$((29600 + RANDOM % 200)) python bench.py $common $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['rccl']; r.pop('note',None)
print(json.dumps({'cfg': '$cfg', 'ms_per_step': d['ms_per_step'], 'p50': d['ms_per_step_p50'], 'rccl': r}))" > /tmp/line.json
  if [ $first = 0 ]; then echo "," >> gpurun_out/r03_rccl_dryrun.json; fi; first=0
  cat /tmp/line.json >> gpurun_out/r03_rccl_dryrun.json
  python -c "
import json; d=json.load(open('/tmp/line.json')); print(d['cfg'], d['ms_per_step'], d['rccl']['ms_per_step_p50_with_exchange'], d['rccl']['ms_per_step_p50_without_exchange'], d['rccl']['allreduce_exposed_ms'])"
done; done
echo "]" >> gpurun_out/r03_rccl_dryrun.json
