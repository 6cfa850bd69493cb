cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_mae_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_mae.py | tee gpurun_out/r03_mae_bench.json; python tools/bench_mae.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_mae_prof -- python tools/bench_mae.py > /dev/null 2>&1
cp $(ls gpurun_out/r03_mae_prof/*/*kernel_stats.csv | head -1) gpurun_out/r03_mae_kernel_stats.csv; rm -rf gpurun_out/r03_mae_prof
python tools/prof_summary.py gpurun_out/r03_mae_kernel_stats.csv 13 24
