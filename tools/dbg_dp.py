"""Debug: where does the host stall in skip mode?  Wraps every ops.* call and HostStager.put with a host timer and prints the
calls that took more than 5 ms, per step."""
import os, sys, time, torch, numpy as np, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mem_amd import ops, utils as U, vit_engine as VE
slow = []
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = (time.perf_counter() - t0) * 1e3
        if dt > 5: slow.append((name, round(dt, 1)))
        return r
    setattr(mod, name, g)
for n in dir(ops):
    if callable(getattr(ops, n)) and not n.startswith("_") and n[0].islower() and n not in ("check", "declare", "ptr", "stream_ptr"):
        try: wrap(ops, n)
        except Exception: pass
op = U.HostStager.put
def put(self, a):
    t0 = time.perf_counter(); r = op(self, a); dt = (time.perf_counter() - t0) * 1e3
    if dt > 5: slow.append(("stager.put[%d]" % self.host[0].numel(), round(dt, 1)))
    return r
U.HostStager.put = put
pl = VE.ViTEngine._dp_plan
def plan(self, dp, B):
    t0 = time.perf_counter(); r = pl(self, dp, B); dt = (time.perf_counter() - t0) * 1e3
    if dt > 5: slow.append(("dp_plan", round(dt, 1)))
    return r
VE.ViTEngine._dp_plan = plan
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-tokenizer-figure", "--no-raster-figure", "--no-config4-figure", "--no-entrypoint-figure", "--no-gemm-timer", "--steps", "60", "--warmup", "5"]
os.environ["MEMHIP_BENCH_STEP_TIMES"] = "1"
import runpy
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
finally:
    c = collections.Counter(n for n, _ in slow)
    print("slow host calls (>5 ms):", dict(c), file=sys.stderr)
    print("examples:", slow[:60], file=sys.stderr)
