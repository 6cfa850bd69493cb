#!/bin/bash
for v in stamp stampnodma; do echo "== $v"; MEMHIP_LIB=mem_amd/exp/$v.so python tools/clock_probe.py 2>&1 | grep "gemm_p8 "; done
