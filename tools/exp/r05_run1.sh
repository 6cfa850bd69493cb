#!/bin/bash
# round 5, call 1: (a) baseline bench of the shipped library on this box, (b) 32x32x16-MFMA main loops (timing builds, wrong
# results) beside the shipped 16x16x32 loops: TFLOP/s per shape (interleaved) and in-kernel clock
cd /root/repo; mkdir -p gpurun_out
python bench.py --steps 30 --warmup 8 > gpurun_out/r05_run1_bench.json 2> gpurun_out/r05_run1_bench.err
tail -c 300 gpurun_out/r05_run1_bench.json; echo
for rep in 1 2; do
  for V in ship mfma32; do
    echo "== $V (rep $rep)"
    if [ $V = ship ]; then python tools/bench_gemm.py; else MEMHIP_LIB=mem_amd/exp/$V.so python tools/bench_gemm.py; fi
  done
done 2>&1 | tee gpurun_out/r05_mfma32_tflops.txt
for V in stamp stamp32 stamp stamp32; do echo "== $V"; MEMHIP_CLOCK_OUT=gpurun_out/r05_clock_$V.json MEMHIP_LIB=mem_amd/exp/$V.so python tools/clock_probe.py 2>&1 | grep "^gemm"; done 2>&1 | tee gpurun_out/r05_mfma32_clock.txt
