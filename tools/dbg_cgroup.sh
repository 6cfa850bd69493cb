#!/bin/bash
# is the host thread throttled by the container's CPU quota?
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; python - <<'PY'
import os, torch
print("affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
PY
st() { grep -E "nr_throttled|throttled_usec|nr_periods|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
for cfg in "" "--no-dp-skip" "" "--no-dp-skip"; do
  st
  MEMHIP_BENCH_STEP_TIMES=1 python bench.py --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure --no-gemm-timer --steps 60 --warmup 5 $cfg 2>&1 | python -c "
import sys,re,json
for l in sys.stdin:
    if 'per-step ms' in l and 'host' not in l:
        v=json.loads(l.split('ms:')[1]); s=sorted(v); print('cfg[$cfg] mean %.2f p50 %.2f max %.2f'%(sum(v)/len(v), s[len(s)//2], s[-1]))
"
  st
done
