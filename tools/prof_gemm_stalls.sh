#!/bin/bash
# usage (GPU box, repo root): tools/prof_gemm_stalls.sh <tag> M N K   -- wave-cycle breakdown of the NT GEMM on one shape
# SQ counters in one pass (8 SQ slots): WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) +
# ACTIVE_INST_ANY ~= WAVE_CYCLES (quad-cycles); MFMA busy in cycles.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/$tag -- python tools/one_gemm.py "$@" > gpurun_out/$tag.log 2>&1
f=$(ls gpurun_out/$tag/*/*counter_collection.csv | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm" not in k: continue
    k = k[k.find("gemm"):][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    w = c["SQ_WAVE_CYCLES"] or 1
    print(k, {x: round(v / w, 3) for x, v in c.items() if x.startswith("SQ_WAIT") or x.startswith("SQ_ACTIVE")},
          "mfma_busy/wave_cycles(quad)", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * w), 3))
PY
rm -rf gpurun_out/$tag
