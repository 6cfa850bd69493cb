#!/bin/bash
# round 5, call 30: the fused 14 x 14 attention backward with 2 / 3 times as many (shorter) workgroups, inside the two-stream step
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2; do
  for V in default a16m2 a16m3; do
    if [ $V = default ]; then L=""; else L="mem_amd/exp/$V.so"; fi
    MEMHIP_LIB=$L python bench.py $F > gpurun_out/r05_a16m_$rep.json 2> gpurun_out/r05_a16m_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_a16m_$rep.json").read().strip().splitlines()[-1])
print("$V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done
for V in "" mem_amd/exp/a16m2.so mem_amd/exp/a16m3.so; do MEMHIP_LIB=$V python tools/attn16_time.py 2>&1 | tail -1; done
