"""-m "not gpu": `python bench.py --gpus N` from a bare shell starts its own rank processes (one per GPU, through
torchrun as a child, before anything touches the GPU), relays their output and prints rank 0's JSON line last; the
driver's own torchrun form keeps working.  Checked here with --rendezvous-only (gloo; no GPU in this container).
Reference launch forms: /root/reference/train-pipeline.sbatch:126, mem/run_mem_pretraining.py:302-309."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e["OMP_NUM_THREADS"] = "1"
    return e


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    out = json.loads(lines[-1])                       # the JSON line is the LAST stdout line
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0
    assert len(lines) == 1, lines                     # everything else went to stderr


def test_driver_torchrun_form():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0


def test_world_size_mismatch_is_loud():
    e = _env(); e.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rendezvous-only"], env=e,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_scale_ab_script_argument_forms():
    """tools/scale_ab.sh (the A/B set for the first N > 1 run: --reserve-cus 0|16 x --bucket-dtype fp32|bf16): every
    bench.py command line it builds parses, starts its ranks and returns a JSON line (SCALE_AB_DRY=1: --rendezvous-only)."""
    e = _env(); e["SCALE_AB_DRY"] = "1"
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_ab.sh"), "2", "3", "1"], env=e, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    rows = [l for l in r.stdout.splitlines() if l.startswith("rep ")]
    assert len(rows) == 8 and all("n_gpus 2" in l for l in rows), r.stdout
    for cus in ("0", "16"):
        for dt in ("fp32", "bf16"):
            assert sum(f"reserve_cus={cus} bucket={dt}:" in l for l in rows) == 2
    lines = open(os.path.join(ROOT, "gpurun_out", "scale_ab_N2.jsonl")).read().splitlines()
    assert len(lines) == 8 and all(json.loads(l)["n_gpus"] == 2 for l in lines)
