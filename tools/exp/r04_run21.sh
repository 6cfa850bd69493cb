#!/bin/bash
cd /root/repo
for V in stamp stampnd4 stamp stampnd4; do echo "== $V"; MEMHIP_LIB=mem_amd/exp/$V.so python tools/clock_probe.py 2>&1 | grep "^gemm_p8"; done
