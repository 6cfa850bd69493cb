#!/bin/bash
# round 5, call 21: group plan settles for a round that is >= 80 % full; engine.wgrad_group 1 vs 2 on ViT-B (bench) and ViT-L (bench_vitl)
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2 3; do
  for V in 1 2; do
    python bench.py $F --wgrad-group $V > gpurun_out/r05_g80_${V}_$rep.json 2> gpurun_out/r05_g80_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_g80_${V}_$rep.json").read().strip().splitlines()[-1])
print("ViT-B wgrad_group $V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done 2>&1 | tee gpurun_out/r05_group80_ab.txt
for rep in 1 2; do for V in 1 2; do echo "ViT-L wgrad_group $V rep $rep: $(MEMHIP_WGRAD_GROUP=$V python tools/bench_vitl.py 64 4 2>&1 | tail -1 | cut -c1-160)"; done; done 2>&1 | tee -a gpurun_out/r05_group80_ab.txt
