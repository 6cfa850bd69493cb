#!/bin/bash
# pass 2 of the long-stream rasterizer with one component removed (wrong results, timing only; variants built with
# tools/build_variant_fast.sh accN raster.hip -DACC_EXP=N: 1 = no LDS atomics, 2 = no output stores, 3 = no key / header loads)
for rep in 1 2; do
  echo "shipped"; VALS=0 python tools/exp/r06_raster_pipe_ab.py 2>&1 | grep uniform | tail -1
  for e in 1 2 3; do echo "ACC_EXP=$e"; MEMHIP_LIB=variants/acc$e.so VALS=0 python tools/exp/r06_raster_pipe_ab.py 2>&1 | grep uniform | tail -1; done
done
