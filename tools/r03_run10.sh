#!/bin/bash
# A/B: last block's MLP branch on the rows that reach the head (default) vs on every row (--no-tail-rows), interleaved, one box
for i in 1 2 3; do
for cfg in "" "--no-tail-rows"; do
  MEMHIP_BENCH_STEP_TIMES=1 python bench.py --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-entrypoint-figure --no-gemm-timer --steps 60 --warmup 5 $cfg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if 'per-step ms' in l and 'host' not in l:
        v=json.loads(l.split('ms:')[1]); s=sorted(v); print('cfg[$cfg] mean %.2f p50 %.2f max %.2f'%(sum(v)/len(v), s[len(s)//2], s[-1]))
"
done; done
