#!/bin/bash
# round 4, run 4: full-line (P/Q) epilogue of gemm_p8 against gemm_p8d (independent path, old layout: bit-equality) and against
# the previous commit's library (time); tests; interleaved step A/B
mkdir -p gpurun_out
python tools/p8d_check.py > gpurun_out/r04_pq_check.log 2>&1; echo "rc=$?" >> gpurun_out/r04_pq_check.log
grep -c true gpurun_out/r04_pq_check.log; tail -3 gpurun_out/r04_pq_check.log
MEMHIP_LIB=mem_amd/exp/r04base.so python tools/p8d_check.py > gpurun_out/r04_pq_check_base.log 2>&1
python - <<'PY'
import json
new=[json.loads(l) for l in open('gpurun_out/r04_pq_check.log') if l.startswith('{')]
old=[json.loads(l) for l in open('gpurun_out/r04_pq_check_base.log') if l.startswith('{')]
for a,b in zip(new,old):
    print(f"M={a['M']} N={a['N']} K={a['K']} {a['epi']:10s} equal={all(a['equal'])} old p8 {b['p8_us']:7.1f} us  new p8 {a['p8_us']:7.1f} us  ({a['p8_us']/b['p8_us']-1:+.1%})  [p8d {a['p8d_us']}]")
PY
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r04_run4_tests.log 2>&1; tail -3 gpurun_out/r04_run4_tests.log
bash tools/ab_lib.sh mem_amd/exp/r04base.so "" 3 > gpurun_out/r04_run4_ab.log 2>&1; cat gpurun_out/r04_run4_ab.log
