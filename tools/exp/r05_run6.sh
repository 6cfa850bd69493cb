#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "attn_win" 2>&1 | tail -15
python -m pytest tests/test_model_gpu.py -q -x -k "config5" 2>&1 | tail -30
