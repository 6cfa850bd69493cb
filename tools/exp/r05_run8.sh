#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_tokenizer_gpu.py -q -x -s 2>&1 | grep -E "passed|failed|deviation|planted|worst" | tail -12
python tools/tok_cert_probe.py 2>&1 | tee gpurun_out/r05_tok_cert_probe.txt
