#!/bin/bash
# usage (on the GPU box, from the repo root): tools/attn_pmc.sh <tag> <attn16 0|1>
# SQ counters of the attention kernels (two passes of 8 SQ counters, counters only next to --kernel-trace).
tag=$1; mode=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
p=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  p=$((p+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/${tag}_p$p -- python3 tools/attn_run.py $mode 3 > gpurun_out/${tag}_p$p.log 2>&1
  f=$(ls gpurun_out/${tag}_p$p/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(attn\w*kernel)", row.get("Kernel_Name", ""))
    if not m: continue
    a = agg[m.group(1)][row["Counter_Name"]]
    a[0] += 1; a[1] += float(row["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k, {c: round(v[1] / v[0]) for c, v in d.items()})
PY
  rm -rf gpurun_out/${tag}_p$p
done
