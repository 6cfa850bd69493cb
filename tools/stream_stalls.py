"""Where the launch stream of the two-stream step waits: per hardware queue of a rocprofv3 kernel trace, the gaps between consecutive
kernels of the SAME queue inside the steady-state window, attributed to the kernel that follows the gap (a gap of the main queue that is
not launch latency is a wait for an event of the weight-gradient stream).   python tools/stream_stalls.py <kernel_trace.csv> [skip_frac]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
qk = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
print("columns:", list(rows[0].keys()))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qk, "0")) for r in rows)
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip
ev = [e for e in ev if e[0] >= t_lo]
steps = sum(1 for e in ev if "adamw_kernel" in e[2]) or 1
byq = collections.defaultdict(list)
for e in ev: byq[e[3]].append(e)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e[1] - e[0] for e in lst)
    gaps = collections.defaultdict(lambda: [0, 0.0])
    tot = 0.0
    for a, b in zip(lst, lst[1:]):
        g = b[0] - a[1]
        if g > 0:
            tot += g
            if g > 6000:
                k = gaps[(a[2][:48], b[2][:48])]; k[0] += 1; k[1] += g
    print(f"queue {q}: {len(lst)} kernels, busy {busy/1e6/steps:.2f} ms/step, gaps {tot/1e6/steps:.2f} ms/step over {steps} steps; gaps > 6 us by (before -> after):")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {v[1]/1e3/steps:8.1f} us/step  {v[0]/steps:6.1f} per step  avg {v[1]/v[0]/1e3:7.1f} us   {k[0]} -> {k[1]}")
