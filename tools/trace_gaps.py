"""GPU idle time of a bench run from a rocprofv3 kernel trace: union of the kernel intervals (all streams) against the
wall span, and the largest gaps with the kernels on either side.   python tools/trace_gaps.py <kernel_trace.csv> [skip_frac]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = iv[0][0] + (iv[-1][1] - iv[0][0]) * skip          # steady state only
iv = [x for x in iv if x[0] >= t_lo]
busy, gaps = 0, []
cur_s, cur_e, last_name = iv[0][0], iv[0][1], iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name[:60], n[:60]))
        cur_s, cur_e, last_name = s, e, n
    elif e > cur_e:
        cur_e, last_name = e, n
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print("span %.2f ms  busy %.2f ms  idle %.2f ms (%.1f%%)  kernels %d  gaps %d" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, 100 * (span - busy) / span, len(iv), len(gaps)))
import collections
by = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    by[(a, b)][0] += 1; by[(a, b)][1] += g
print("idle by (kernel before -> kernel after), top 25:")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %8.1f us total  %5d gaps  avg %6.1f us   %s -> %s" % (v[1] / 1e3, v[0], v[1] / v[0] / 1e3, k[0], k[1]))
