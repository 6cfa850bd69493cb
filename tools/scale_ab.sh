#!/bin/bash
# tools/scale_ab.sh N [steps] [warmup] [extra bench.py args...]: the A/B set to run right after the first `bench.py --gpus N`
# on a multi-GPU node (the 8-GPU runs are the driver's: nothing here was ever executed on more than one device).  Four
# interleaved configurations of the SAME bench, each a fresh set of rank processes (bench.py starts its own torchrun child):
#     --reserve-cus 0|16   CUs the persistent GEMM / attention grids leave to RCCL's channel kernels while buckets are in flight
#     --bucket-dtype fp32|bf16   wire format of the gradient buckets (fp32 = the reference's DDP exchange)
# and per run one line: the configuration, samples/s, ms per step, the exposed part of the gradient exchange
# (rccl.allreduce_exposed_ms) and the devices RCCL ran on (rccl.ranks: N distinct pci_bus_ids).  The full JSON lines go to
# gpurun_out/scale_ab_N<N>.jsonl.  SCALE_AB_DRY=1 (tests/test_bench_launch.py, no GPU): --rendezvous-only is added to every
# run, so only the launch forms are exercised.
set -u
N=${1:?usage: tools/scale_ab.sh N [steps] [warmup] [bench args]}; STEPS=${2:-20}; WARM=${3:-5}
shift; [ $# -gt 0 ] && shift; [ $# -gt 0 ] && shift
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/scale_ab_N${N}.jsonl; : > "$OUT"
EXTRA=(--no-cpu-baseline --no-gemm-timer "$@")
[ "${SCALE_AB_DRY:-0}" = 1 ] && EXTRA+=(--rendezvous-only)
rc=0
for rep in 1 2; do
  for cfg in "0 fp32" "16 fp32" "0 bf16" "16 bf16"; do
    set -- $cfg
    line=$(python bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --reserve-cus "$1" --bucket-dtype "$2" "${EXTRA[@]}" 2>>gpurun_out/scale_ab_N${N}.err | tail -1)
    [ -z "$line" ] && { echo "rep $rep reserve_cus=$1 bucket=$2: no JSON line (see gpurun_out/scale_ab_N${N}.err)"; rc=1; continue; }
    echo "$line" >> "$OUT"
    python - "$rep" "$1" "$2" "$line" <<'PY'
import json, sys
rep, cus, dt, line = sys.argv[1:5]
d = json.loads(line)
r = d.get("rccl") or {}
ids = sorted({str(x.get("pci_bus_id")) for x in r.get("ranks", [])})
print(f"rep {rep} reserve_cus={cus} bucket={dt}: n_gpus {d.get('n_gpus')} value {d.get('value')} ms_per_step {d.get('ms_per_step')} "
      f"exposed_ms {r.get('allreduce_exposed_ms')} exchange_minus_no_hook_ms {r.get('exchange_minus_no_hook_ms')} "
      f"devices {len(ids)} {ids}")
PY
  done
done
exit $rc
