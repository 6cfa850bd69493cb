#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -m gpu -x -q > gpurun_out/r04_run6_tests.log 2>&1; tail -5 gpurun_out/r04_run6_tests.log
