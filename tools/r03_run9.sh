#!/bin/bash
# relaxed waits behind the GEMM epilogue: correctness + A/B against the -DP8_RELAX=0 build
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -q -x 2>&1 | tail -3
for i in 1 2; do
  echo "== relax"; timeout 300 python tools/epi_probe.py fc1 fc2 proj qkv 2>&1 | grep -v stagger
  echo "== norelax"; MEMHIP_LIB=mem_amd/exp/norelax.so timeout 300 python tools/epi_probe.py fc1 fc2 proj qkv 2>&1 | grep -v stagger
done
for i in 1 2; do
  for lib in "" mem_amd/exp/norelax.so; do
    MEMHIP_LIB=$lib MEMHIP_BENCH_STEP_TIMES=1 python bench.py --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-entrypoint-figure --no-gemm-timer --steps 40 --warmup 5 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if 'per-step ms' in l and 'host' not in l:
        v=json.loads(l.split('ms:')[1]); s=sorted(v); print('lib[$lib] mean %.2f p50 %.2f max %.2f'%(sum(v)/len(v), s[len(s)//2], s[-1]))
"
  done
done
