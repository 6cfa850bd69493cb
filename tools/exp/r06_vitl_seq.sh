cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_vitl.py 64 4 > gpurun_out/r06_vitl_two.json 2>&1
MEMHIP_NO_SIDE=1 python tools/bench_vitl.py 64 4 > gpurun_out/r06_vitl_seq.json 2>&1
export MEMHIP_NO_SIDE=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_vitl_seq -- python tools/bench_vitl.py 64 3 > gpurun_out/r06_vitl_seq.log 2>&1
cp $(ls gpurun_out/r06_vitl_seq/*/*kernel_stats.csv | head -1) gpurun_out/r06_vitl_seq_kernel_stats.csv; rm -rf gpurun_out/r06_vitl_seq
tail -1 gpurun_out/r06_vitl_two.json; tail -1 gpurun_out/r06_vitl_seq.json
