#!/bin/bash
# round 5, call 17: nontemporal loads of read-once operands inside the step, interleaved: auxnt (GELU' second operand), adamnt1 / adamnt3
# (AdamW gradient + moment loads / + moment stores), lnbnt (LayerNorm-backward activation and gradient rows)
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2 3; do
  for V in default auxnt adamnt1 adamnt3 lnbnt; do
    if [ $V = default ]; then L=""; else L="mem_amd/exp/$V.so"; fi
    MEMHIP_LIB=$L python bench.py $F > gpurun_out/r05_nt_${V}_$rep.json 2> gpurun_out/r05_nt_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_nt_${V}_$rep.json").read().strip().splitlines()[-1])
print("$V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done 2>&1 | tee gpurun_out/r05_nt_loads_ab.txt
