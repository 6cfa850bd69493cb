# per-launch durations of the raw fp16x2 tokenizer forward (256 x 224^2): which layer runs at what rate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tokl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tokl -- python tools/tok_layers_prof.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tokl/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
conv = [(r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or "") for r in rows]
# the last forward: find the last nchw_to_padded kernel
idx = max(i for i, c in enumerate(conv) if "nchw_to_padded" in c[0])
tot = 0.0
for name, us, g, w in conv[idx:]:
    print(f"{name:42s} {us:9.1f} us  grid {g} wg {w}"); tot += us
print("total", round(tot / 1e3, 2), "ms")
PY
rm -rf gpurun_out/tokl
