"""Debug: which threads of a bench.py run burn CPU (the GPU box's container has a 16-CPU CFS quota on a 256-core host)."""
import os, sys, runpy, collections
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-tokenizer-figure", "--no-raster-figure", "--no-config4-figure",
            "--no-entrypoint-figure", "--no-gemm-timer", "--steps", "60", "--warmup", "5"] + sys.argv[1:]
os.environ["MEMHIP_BENCH_STEP_TIMES"] = "1"
try:
    runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
finally:
    tck = os.sysconf("SC_CLK_TCK")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for t in os.listdir("/proc/self/task"):
        try:
            s = open(f"/proc/self/task/{t}/stat").read()
            comm = s[s.index("(") + 1:s.rindex(")")]
            f = s[s.rindex(")") + 2:].split()
            agg[comm][0] += 1
            agg[comm][1] += (int(f[11]) + int(f[12])) / tck
        except Exception:
            pass
    print("threads by name: (count, cpu s)", {k: (v[0], round(v[1], 1)) for k, v in agg.items()}, file=sys.stderr)
