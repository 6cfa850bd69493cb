#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-raster-figure --no-config5-figure --no-config4-figure --no-tokenizer-figure > gpurun_out/r05_entry.json 2> gpurun_out/r05_entry.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_entry.json").read().strip().splitlines()[-1])
e=d["entrypoint"]; print({k:v for k,v in e.items() if k not in ("workload",)})
PY
