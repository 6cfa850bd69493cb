common="--no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-entrypoint-figure --no-gemm-timer --steps 60 --warmup 5"
for cfg in "" "--no-dp-skip" "" "--no-dp-skip"; do
MEMHIP_BENCH_STEP_TIMES=1 python bench.py $common $cfg 2>&1 | grep "per-step ms" | tail -1 | python -c "
import sys,re,statistics as st
v=[float(x) for x in re.findall(r'[0-9.]+', sys.stdin.read().split(':',1)[1])]
v2=sorted(v); print('cfg[$cfg] mean %.2f p50 %.2f p10 %.2f p90 %.2f max %.2f  n>p50+1: %d' % (sum(v)/len(v), v2[len(v)//2], v2[len(v)//10], v2[9*len(v)//10], v2[-1], sum(x>v2[len(v)//2]+1 for x in v)))"
done
