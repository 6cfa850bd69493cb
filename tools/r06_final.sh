#!/bin/bash
# Round-6 evidence set: ONE call on ONE box after the last kernel commit (run from the repo root on the GPU box; outputs under
# gpurun_out/r06fin_*, copied to profiles/r06_final_* by tools/r06_collect.py):
#   r06fin_tests.txt               python -m pytest tests -m gpu
#   r06fin_bench.json              python bench.py (defaults: every secondary figure, cpu_baseline, config5 ViT-L line)
#   r06fin_seq_kernel_stats.csv    rocprofv3 --kernel-trace --stats, weight gradients on the launch stream (per-kernel accounting)
#   r06fin_seq_step_kernels.txt    per-step table from the TRACE of the same run (tools/step_kernels.py)
#   r06fin_two_kernel_stats.csv    the default two-stream step (durations stretched by the overlap) + r06fin_gaps.txt
#   r06fin_mfma_util.json          SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes
#   r06fin_traffic.json            HBM FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
#   r06fin_vitl_kernel_stats.csv   ViT-L/16 480x640 (config #5), B = 64;  r06fin_vitl_mfma_util.json, r06fin_vitl_traffic.json
#   r06fin_raster_kernel_stats.csv config #4 rasterizer (64 x 1 M events);  r06fin_raster_traffic.json
#   r06fin_attn16.txt, r06fin_attn_win.txt   attention kernels alone (attn_win: 0 = stream kernels, 1 = slot layout, 9 = 1 + dS workspace)
#   r06fin_conv_waves.txt          fp16x2 tokenizer convolutions: 4 / 8 waves per workgroup (128 x 128 tiles) / 8 waves + the phase-interleaved 256 x 128 tile
#   r06fin_clock.json              in-kernel shader clock of the GEMM main loops (stamp build)
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r06fin_tests.txt; cat gpurun_out/r06fin_tests.txt
python bench.py > gpurun_out/r06fin_bench.json 2> gpurun_out/r06fin_bench.err
tail -c 400 gpurun_out/r06fin_bench.json; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BARGS="--no-cpu-baseline --no-gemm-timer --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure"
# ---- ViT-B step: sequential trace (stats + per-step table from the same trace), two-stream trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06fin_seq -- python bench.py $BARGS --no-side-stream --steps 10 --warmup 3 > gpurun_out/r06fin_seq.log 2>&1
cp $(ls gpurun_out/r06fin_seq/*/*kernel_stats.csv | head -1) gpurun_out/r06fin_seq_kernel_stats.csv
python tools/step_kernels.py $(ls gpurun_out/r06fin_seq/*/*kernel_trace.csv | head -1) 3 50 > gpurun_out/r06fin_seq_step_kernels.txt
rm -rf gpurun_out/r06fin_seq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06fin_two -- python bench.py $BARGS --steps 10 --warmup 3 > gpurun_out/r06fin_two.log 2>&1
cp $(ls gpurun_out/r06fin_two/*/*kernel_stats.csv | head -1) gpurun_out/r06fin_two_kernel_stats.csv
python tools/trace_gaps.py $(ls gpurun_out/r06fin_two/*/*kernel_trace.csv | head -1) 0.8 > gpurun_out/r06fin_gaps.txt 2>&1
rm -rf gpurun_out/r06fin_two
# ---- ViT-B step: counters
tools/prof_mfma.sh r06fin_mfma
python tools/mfma_util.py gpurun_out/r06fin_mfma > gpurun_out/r06fin_mfma_util.json
tools/prof_pmc.sh r06fin_pmc
python tools/pmc_combine.py gpurun_out/r06fin_pmc_FETCH_SIZE.json gpurun_out/r06fin_pmc_WRITE_SIZE.json gpurun_out/r06fin_traffic.json > /dev/null
rm -rf gpurun_out/r06fin_pmc_FETCH_SIZE gpurun_out/r06fin_pmc_WRITE_SIZE
# ---- ViT-L (config #5), B = 64: trace + counters (none existed before round 5)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06fin_vitl -- python tools/bench_vitl.py 64 4 > gpurun_out/r06fin_vitl.log 2>&1
cp $(ls gpurun_out/r06fin_vitl/*/*kernel_stats.csv | head -1) gpurun_out/r06fin_vitl_kernel_stats.csv; rm -rf gpurun_out/r06fin_vitl
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r06fin_vitl_$c -- python tools/bench_vitl.py 64 1 > gpurun_out/r06fin_vitl_$c.log 2>&1
  python tools/pmc_summary.py $(ls gpurun_out/r06fin_vitl_$c/*/*counter_collection.csv | head -1) $c split > gpurun_out/r06fin_vitl_$c.json
  rm -rf gpurun_out/r06fin_vitl_$c
done
cp gpurun_out/r06fin_vitl_SQ_VALU_MFMA_BUSY_CYCLES.json gpurun_out/r06fin_vitlm_SQ_VALU_MFMA_BUSY_CYCLES.json
cp gpurun_out/r06fin_vitl_GRBM_GUI_ACTIVE.json gpurun_out/r06fin_vitlm_GRBM_GUI_ACTIVE.json
python tools/mfma_util.py gpurun_out/r06fin_vitlm > gpurun_out/r06fin_vitl_mfma_util.json
python tools/pmc_combine.py gpurun_out/r06fin_vitl_FETCH_SIZE.json gpurun_out/r06fin_vitl_WRITE_SIZE.json gpurun_out/r06fin_vitl_traffic.json > /dev/null
# ---- config #4 rasterizer: trace + traffic
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06fin_raster -- python tools/raster_bench.py > gpurun_out/r06fin_raster.log 2>&1
cp $(ls gpurun_out/r06fin_raster/*/*kernel_stats.csv | head -1) gpurun_out/r06fin_raster_kernel_stats.csv; rm -rf gpurun_out/r06fin_raster
tools/prof_pmc_raster.sh r06fin_raster > /dev/null 2>&1
# ---- attention kernels alone, GEMM clock
python tools/attn16_time.py 2>&1 | tail -2 > gpurun_out/r06fin_attn16.txt
WIN_TIME_ONLY=1 WIN_MODES=0,1,9 python tools/attn_win_check.py all time 2>&1 | grep "^mode" > gpurun_out/r06fin_attn_win.txt
WAVES=4,8,16 python tools/exp/r06_conv_waves.py 2>&1 | grep -v amdgpu > gpurun_out/r06fin_conv_waves.txt
MEMHIP_CLOCK_OUT=gpurun_out/r06fin_clock.json MEMHIP_LIB=variants/stamp.so python tools/clock_probe.py > gpurun_out/r06fin_clock.log 2>&1
ls -la gpurun_out | grep r06fin
