#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tools/prof.sh r04fin_seq2 --no-side-stream --steps 10 --warmup 3
t=$(ls gpurun_out/r04fin_seq2/*/*kernel_trace.csv | head -1)
python tools/step_kernels.py $t 3 60 > gpurun_out/r04fin_seq_step_kernels.txt
rm -rf gpurun_out/r04fin_seq2; head -3 gpurun_out/r04fin_seq_step_kernels.txt; grep -i "fill\|zero" gpurun_out/r04fin_seq_step_kernels.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_p50'])"; done
