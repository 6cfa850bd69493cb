#!/bin/bash
# Round-4 evidence set (run on the GPU box from the repo root; outputs under gpurun_out/, copied to profiles/ afterwards):
#   r04fin_tests.txt             python -m pytest tests -m gpu
#   r04fin_bench.json            python bench.py (defaults: every secondary figure, cpu_baseline, config5 ViT-L line)
#   r04fin_seq_kernel_stats.csv  rocprofv3 --kernel-trace --stats, weight gradients on the launch stream (per-kernel accounting)
#   r04fin_two_kernel_stats.csv  the default two-stream step (durations stretched by the overlap)
#   r04fin_mfma_*.json           SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes -> mfma_util
#   r04fin_FETCH/WRITE_SIZE.json HBM traffic per kernel (separate passes)
#   r04fin_vitl_kernel_stats.csv ViT-L/16 480x640 (config #5), B = 16
#   r04fin_attn16.txt            14x14 attention kernels alone, B = 256
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r04fin_tests.txt; cat gpurun_out/r04fin_tests.txt
python bench.py > gpurun_out/r04fin_bench.json 2> gpurun_out/r04fin_bench.err
tail -c 400 gpurun_out/r04fin_bench.json; echo
tools/prof.sh r04fin_seq --no-side-stream --steps 10 --warmup 3
tools/prof.sh r04fin_two --steps 10 --warmup 3
python tools/trace_gaps.py $(ls gpurun_out/r04fin_two/*/*kernel_trace.csv | head -1) 0.8 > gpurun_out/r04fin_gaps.txt 2>&1
rm -rf gpurun_out/r04fin_seq gpurun_out/r04fin_two
tools/prof_mfma.sh r04fin_mfma
python tools/mfma_util.py gpurun_out/r04fin_mfma > gpurun_out/r04fin_mfma_util.json
tools/prof_pmc.sh r04fin_pmc
python tools/pmc_combine.py gpurun_out/r04fin_pmc_FETCH_SIZE.json gpurun_out/r04fin_pmc_WRITE_SIZE.json gpurun_out/r04fin_traffic.json > /dev/null
rm -rf gpurun_out/r04fin_pmc_FETCH_SIZE gpurun_out/r04fin_pmc_WRITE_SIZE
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04fin_vitl -- python tools/bench_vitl.py 16 5 > gpurun_out/r04fin_vitl.log 2>&1
f=$(ls gpurun_out/r04fin_vitl/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r04fin_vitl_kernel_stats.csv; rm -rf gpurun_out/r04fin_vitl
python tools/attn16_time.py 2>&1 | tail -2 > gpurun_out/r04fin_attn16.txt
ls -la gpurun_out | grep r04fin
