#!/bin/bash
# round 5, call 2: certified tokenizer tests, data-parallel tests with the hook on the side stream, one-rank RCCL dry run with
# the round-4 per-layer join (--hook-join) beside the new hook placement (interleaved), tokenizer figure of the bench
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_tokenizer_gpu.py tests/test_ddp_gpu.py tests/test_train_gpu.py -q -x -s 2>&1 | tail -40 > gpurun_out/r05_run2_tests.txt; tail -15 gpurun_out/r05_run2_tests.txt
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2; do
  for V in join side; do
    if [ $V = join ]; then X="--hook-join"; else X=""; fi
    MEMHIP_BENCH_FORCE_DIST=1 python bench.py $F $X > gpurun_out/r05_dry_${V}_$rep.json 2> gpurun_out/r05_dry_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_dry_${V}_$rep.json").read().strip().splitlines()[-1])
r=d["rccl"]; print("$V $rep", d["ms_per_step"], {k:r[k] for k in r if k.startswith("ms_") or k.endswith("_ms") or k=="hook_stream"})
PY
  done
done 2>&1 | tee gpurun_out/r05_dry_ab.txt
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-raster-figure --no-config5-figure --no-config4-figure > gpurun_out/r05_run2_bench.json 2> gpurun_out/r05_run2_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_run2_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("with_tokenizer"))[:3000]); print(json.dumps(d.get("entrypoint"))[:600])
PY
