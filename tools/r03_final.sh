#!/bin/bash
# Round-3 evidence set (run on the GPU box from the repo root; outputs under gpurun_out/, copied to profiles/ afterwards):
#   r03fin_bench.json            python bench.py (defaults: 100 steps after 20 warm-up, every secondary figure, cpu_baseline)
#   r03fin_seq_kernel_stats.csv  rocprofv3 --kernel-trace --stats, weight gradients on the launch stream (per-kernel accounting)
#   r03fin_two_kernel_stats.csv  the default two-stream step (durations stretched by the overlap)
#   r03fin_mfma_*.json           SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes -> mfma_util
#   r03fin_FETCH/WRITE_SIZE.json HBM traffic per kernel (separate passes)
python bench.py > gpurun_out/r03fin_bench.json 2> gpurun_out/r03fin_bench.err
tail -c 400 gpurun_out/r03fin_bench.json; echo
tools/prof.sh r03fin_seq --no-side-stream --steps 10 --warmup 3
tools/prof.sh r03fin_two --steps 10 --warmup 3
python tools/trace_gaps.py $(ls gpurun_out/r03fin_two/*/*kernel_trace.csv | head -1) 0.8 > gpurun_out/r03fin_gaps.txt 2>&1
rm -rf gpurun_out/r03fin_seq gpurun_out/r03fin_two
tools/prof_mfma.sh r03fin_mfma
python tools/mfma_util.py gpurun_out/r03fin_mfma > gpurun_out/r03fin_mfma_util.json
tools/prof_pmc.sh r03fin_pmc
python tools/pmc_combine.py gpurun_out/r03fin_pmc_FETCH_SIZE.json gpurun_out/r03fin_pmc_WRITE_SIZE.json gpurun_out/r03fin_traffic.json > /dev/null
rm -rf gpurun_out/r03fin_pmc_FETCH_SIZE gpurun_out/r03fin_pmc_WRITE_SIZE
ls -la gpurun_out | grep r03fin
