#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_any.sh <tag> <script.py> [args]   -- kernel-time summary of any script
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python "$@" > gpurun_out/$tag.log 2>&1
tail -2 gpurun_out/$tag.log | cut -c1-400
f=$(ls gpurun_out/$tag/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
