"""Per-step kernel table of a bench run from a rocprofv3 kernel TRACE, restricted to the steady-state steps: a step ends
with its adamw_kernel launch, so the window runs from the end of the (skip+1)-th adamw_kernel to the end of the last
one -- set-up work (buffer allocation fills, warm-up, the bench's secondary figures) is not attributed to the steps,
unlike a --stats summary divided by the step count.
    python tools/step_kernels.py <kernel_trace.csv> [steps to skip = 3] [rows = 40]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ends = [e for s, e, n in iv if "adamw_kernel" in n]
assert len(ends) > skip + 1, "not enough adamw_kernel launches in the trace"
t0, t1, steps = ends[skip], ends[-1], len(ends) - 1 - skip
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]
by = collections.defaultdict(lambda: [0, 0])
tot = 0
for s, e, n in iv:
    if s >= t0 and e <= t1:
        k = short(n); by[k][0] += 1; by[k][1] += e - s; tot += e - s
print("window: %d steps, %.3f ms wall per step, %.3f ms of kernel time per step" % (steps, (t1 - t0) / 1e6 / steps, tot / 1e6 / steps))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-70s  launches/step %7.2f  avg %8.1f us  per-step %7.3f ms  %5.1f%%" % (k, v[0] / steps, v[1] / v[0] / 1e3, v[1] / 1e6 / steps, 100.0 * v[1] / tot))
fills = sum(v[1] for k, v in by.items() if "FillFunctor" in k)
print("at::native FillFunctor kernels inside the window: %d launches, %.3f ms per step" % (sum(v[0] for k, v in by.items() if "FillFunctor" in k), fills / 1e6 / steps))
