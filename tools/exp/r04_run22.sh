#!/bin/bash
cd /root/repo
python -m pytest tests/test_train_gpu.py -q -x -m gpu 2>&1 | tail -3
bash tools/ab_bench.sh "" "--opt-overlap" 3
