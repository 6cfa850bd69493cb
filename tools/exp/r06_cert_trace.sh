# kernel timeline of ONE certified fp16x2 tokenizer call (bench tokenizer, 256 uniform-random images): what the certification costs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/cert_one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mem_amd.vae_model import DiscreteVAE, HipTokenizer
torch.manual_seed(20251)
vae = DiscreteVAE(input_H=224, input_W=224, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3, hidden_dim=384, channels=3).cuda().eval()
img = torch.rand(256, 3, 224, 224, device="cuda")
tok = HipTokenizer(vae, max_batch=256, precision="fp16x2", audit_every=0)
for _ in range(3): tok.get_codebook_indices(img)
torch.cuda.synchronize()
print(tok.certification_stats())
PY
rm -rf gpurun_out/certt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/certt -- python /tmp/cert_one.py > gpurun_out/certt.log 2>&1
tail -1 gpurun_out/certt.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/certt/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "nchw_to_padded_nhwc4_f16x2" in n)
t0 = int(rows[last]["Start_Timestamp"])
tot = {}
for r in rows[last:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {n:46s} {d:8.1f} us  grid {r.get('Grid_Size_X') or r.get('Grid_Size')}")
print("span", (int(rows[-1]["End_Timestamp"]) - t0) / 1e3, "us")
PY
rm -rf gpurun_out/certt
