python -m pytest tests -m gpu -q -x 2>&1 | tail -4
common="--no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-entrypoint-figure --no-gemm-timer --steps 20 --warmup 5"
for i in 1 2 3; do
for cfg in "" "--no-dp-skip"; do
python bench.py $common $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg[$cfg]', d['ms_per_step'], d['ms_per_step_p50'], d['config']['last_loss'])"
done; done
