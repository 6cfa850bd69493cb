#!/bin/bash
# tools/attn_win_prof.sh <modes>: per-kernel durations of the window attention at the config-#5 size (rocprofv3 --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WIN_MODES=${1:-1,9} WIN_TIME_ONLY=1
rm -rf gpurun_out/awp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/awp -- python tools/attn_win_check.py all time > gpurun_out/awp.log 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/awp/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n)[:52]
    if "attn" in n: print(f"{n:54s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
rm -rf gpurun_out/awp
