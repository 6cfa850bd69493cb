#!/bin/bash
cd /root/repo
for V in stamp stampnodma stampnoread; do echo "== $V"; MEMHIP_LIB=mem_amd/exp/$V.so python tools/clock_probe.py 2>&1 | grep "^gemm_p8"; done
