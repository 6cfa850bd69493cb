#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tools/prof.sh r04fin_seq2 --no-side-stream --steps 10 --warmup 3
t=$(ls gpurun_out/r04fin_seq2/*/*kernel_trace.csv | head -1)
python tools/step_kernels.py $t 3 60 > gpurun_out/r04fin_seq_step_kernels.txt
rm -rf gpurun_out/r04fin_seq2; head -50 gpurun_out/r04fin_seq_step_kernels.txt; tail -1 gpurun_out/r04fin_seq_step_kernels.txt
