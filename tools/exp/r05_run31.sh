#!/bin/bash
# round 5, call 31: where the launch stream waits for the weight-gradient stream (tools/stream_stalls.py on a two-stream trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
BARGS="--no-cpu-baseline --no-gemm-timer --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_st -- python bench.py $BARGS --steps 12 --warmup 4 > gpurun_out/r05_st.log 2>&1
python tools/stream_stalls.py $(ls gpurun_out/r05_st/*/*kernel_trace.csv | head -1) 0.6 > gpurun_out/r05_stream_stalls.txt 2>&1
rm -rf gpurun_out/r05_st
cat gpurun_out/r05_stream_stalls.txt | cut -c1-200
