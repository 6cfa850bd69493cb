common="--no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-entrypoint-figure --steps 24 --warmup 5"
for i in 1 2; do
for st in 0 4000 -2000 -4000; do
python bench.py $common --opt gemm_stagger=$st 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); f=d['roofline']['gemm_family']['per_epilogue']
print('stagger $st', d['ms_per_step_p50_uninstrumented'], {k:(v['avg_us']) for k,v in f.items()})"
done; done
