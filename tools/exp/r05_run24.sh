#!/bin/bash
# round 5, call 24: per-layer kernel times of the raw fp16x2 tokenizer forward (256 x 224^2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_tok -- python tools/tok_layers_prof.py > gpurun_out/r05_tok.log 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/r05_tok/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last repetition: find kernels after the 3rd occurrence of the first conv kernel name
names = [r["Kernel_Name"] for r in rows]
conv = [i for i, n in enumerate(names) if "conv_gemm_f16x2" in n or "nchw_to_padded" in n or "argmax" in n]
per = len(conv) // 4
last = conv[-per:]
tot = 0
for i in last:
    r = rows[i]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    print(f'{re.sub(r"[(].*", "", r["Kernel_Name"])[:60]:60s} grid {r.get("Grid_Size_X", r.get("Grid_Size", "?")):>8s} {d:9.1f} us')
print("sum", round(tot, 1), "us")
PY
rm -rf gpurun_out/r05_tok
