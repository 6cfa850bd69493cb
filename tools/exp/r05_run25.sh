#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python - <<'PY'
import os
os.environ["MEMHIP_LIB"]="mem_amd/exp/tnnt.so"
import sys; sys.path.insert(0, ".")
import torch
from mem_amd import ops
M=50432
A=torch.randn(M,768,device="cuda").bfloat16(); B=torch.randn(M,3072,device="cuda").bfloat16(); o=torch.zeros(768,3072,device="cuda")
ws=torch.empty(ops.gemm_tn_workspace(M,768,3072),dtype=torch.uint8,device="cuda")
ops.gemm_tn(A,B,M,768,3072,o,accumulate=False,workspace=ws)
ref=A.float().t()@B.float()
print("nt variant rel err", ((o-ref).norm()/ref.norm()).item())
PY
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2 3; do
  for V in default tnnt; do
    if [ $V = default ]; then L=""; else L="mem_amd/exp/$V.so"; fi
    MEMHIP_LIB=$L python bench.py $F > gpurun_out/r05_tnnt_${V}_$rep.json 2> gpurun_out/r05_tnnt_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_tnnt_${V}_$rep.json").read().strip().splitlines()[-1])
print("$V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done
