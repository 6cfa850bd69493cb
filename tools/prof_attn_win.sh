#!/bin/bash
# usage (GPU box, repo root): tools/prof_attn_win.sh <tag> <mode>  -- wave-cycle breakdown of the window-attention forward
tag=$1; mode=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  rm -rf gpurun_out/$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/$tag -- python tools/attn_win_one.py $mode 3 > gpurun_out/$tag.log 2>&1
  f=$(ls gpurun_out/$tag/*/*counter_collection.csv | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "attn" not in k: continue
    k = k[k.find("attn"):][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    w = c["SQ_WAVE_CYCLES"] or 1
    print(k, {x: round(v / w, 3) for x, v in c.items() if x != "SQ_WAVE_CYCLES"}, "wave_cycles", w)
PY
done
rm -rf gpurun_out/$tag
