"""Combine the two PMC passes of tools/prof_pmc.sh into profiles/gemm_traffic.json:
python tools/pmc_combine.py gpurun_out/<tag>_FETCH_SIZE.json gpurun_out/<tag>_WRITE_SIZE.json profiles/gemm_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read
(MI355X_MICROARCH.md, HBM section), so it is doubled."""
import json, sys
f, w, out = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3]
res = {}
for k in sorted(set(f) | set(w)):
    fb = 2.0 * f.get(k, {}).get("avg", 0.0) * 1024
    wb = w.get(k, {}).get("avg", 0.0) * 1024
    res[k] = {"launches": f.get(k, w.get(k))["launches"], "fetch_bytes": round(fb), "write_bytes": round(wb),
              "hbm_bytes_per_launch": round(fb + wb)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
