"""Sum one rocprofv3 PMC counter per kernel family: python tools/pmc_summary.py counter_collection.csv COUNTER
Prints JSON {family: {"launches": n, "avg": per-launch counter value}}; a third argument "split" keeps template
arguments (gemm_p8_kernel<3,256> ...) as separate families."""
import csv, json, re, sys
from collections import defaultdict
path, counter = sys.argv[1], sys.argv[2]
SPLIT = len(sys.argv) > 3 and sys.argv[3] == "split"
agg = defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != counter:
            continue
        name = row.get("Kernel_Name", "")
        m = re.search(r"(gemm\w*kernel|attn\w*kernel|ln_\w+|branch_bwd\w*|adamw\w*|raster\w*|event_norm\w*|ce_kernel|colsum\w*)(<[^>]*>)?", name)
        fam = m.group(1) if m else "other"
        if m and m.group(2) and SPLIT:                      # keep the template arguments (epilogue, tile height) apart
            fam += m.group(2).replace(" ", "")
        a = agg[fam]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
print(json.dumps({k: {"launches": v[0], "avg": v[1] / v[0]} for k, v in sorted(agg.items())}, indent=1))
