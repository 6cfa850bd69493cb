"""Combine the SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE passes of tools/prof_mfma.sh into per-kernel-family
MFMA-pipe utilisation (rocprofv3's derived MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) x SIMDs);
pmc_summary.py SUMS the 8 per-XCD GRBM_GUI_ACTIVE instances, so max ~= sum / 8).
python tools/mfma_util.py gpurun_out/<tag> > profiles/mfma_util.json"""
import json, sys
tag = sys.argv[1]
busy = json.load(open(f"{tag}_SQ_VALU_MFMA_BUSY_CYCLES.json"))
gui = json.load(open(f"{tag}_GRBM_GUI_ACTIVE.json"))
XCDS, SIMDS = 8, 1024
out, tb, tg = {}, 0.0, 0.0
for k, v in busy.items():
    if k not in gui or not gui[k]["avg"]:
        continue
    cyc = gui[k]["avg"] / XCDS                      # kernel duration in shader cycles
    n = v["launches"]
    tb += n * v["avg"]
    tg += n * cyc
    if v["avg"] > 0:
        out[k] = {"launches": n, "mfma_busy_cycles_per_launch": round(v["avg"]), "gpu_cycles_per_launch": round(cyc),
                  "mfma_util": round(v["avg"] / (cyc * SIMDS), 4)}
out["_whole_step"] = {"mfma_util": round(tb / (tg * SIMDS), 4),
                      "note": "all kernels of the profiled steps: sum of MFMA-busy cycles / (sum of kernel cycles x 1024 SIMDs); "
                              "cycles are actual shader clocks, so this is utilisation of the pipe at the clock the chip held"}
print(json.dumps(out, indent=1))
