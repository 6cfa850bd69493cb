#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_tokenizer_gpu.py -q -x -s 2>&1 | grep -E "passed|failed|planted" | tail -4
python tools/tok_cert_probe.py 2>&1 | grep -v "capacity 32\|capacity 64" | tee gpurun_out/r05_tok_cert_probe.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-raster-figure --no-config5-figure --no-config4-figure > gpurun_out/r05_tokbench.json 2> gpurun_out/r05_tokbench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_tokbench.json").read().strip().splitlines()[-1])
w=d["with_tokenizer"]; f=w["fp16x2_mode"]; print("with_tok", w["value"], w["ms_per_step"], f["tokenizer_ms_per_step"], f["raw_fp16x2_tokenizer_ms"], f["certification"])
e=d["entrypoint"]; print("entry", e["value"], e["ms_per_step"], e["stages_alone_ms"], e["tokenizer_certification"]["flagged_samples_per_batch"])
PY
