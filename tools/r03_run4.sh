# rasterizer: A/B nontemporal loads, rocprofv3 kernel stats at both sizes, event_norm PMC traffic
for i in 1 2; do
echo "== shipped"; python tools/raster_small_bench.py | tail -1; B=32 python tools/raster_bench.py | grep binned
echo "== nt"; MEMHIP_LIB=mem_amd/exp/raster_nt.so python tools/raster_small_bench.py | tail -1; MEMHIP_LIB=mem_amd/exp/raster_nt.so B=32 python tools/raster_bench.py | grep binned
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_raster_small -- python tools/raster_small_bench.py > /dev/null 2>&1
cp $(ls gpurun_out/r03_raster_small/*/*kernel_stats.csv | head -1) gpurun_out/r03_raster_256x30k_kernel_stats.csv
B=32 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_raster_1m -- python tools/raster_bench.py > /dev/null 2>&1
cp $(ls gpurun_out/r03_raster_1m/*/*kernel_stats.csv | head -1) gpurun_out/r03_raster_32x1m_kernel_stats.csv
rm -rf gpurun_out/r03_raster_small gpurun_out/r03_raster_1m
head -5 gpurun_out/r03_raster_256x30k_kernel_stats.csv | cut -c1-200
head -6 gpurun_out/r03_raster_32x1m_kernel_stats.csv | cut -c1-200
# event_norm HBM traffic at the bench size (256 x 224^2 u8 -> 2-chan f32)
cat > /tmp/en_bench.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mem_amd import transforms as T
img = torch.randint(0, 4, (256, 3, 224, 224), dtype=torch.uint8, device="cuda")
for _ in range(5): T.event_norm(img, T.EV_RM_TS | T.EV_HOTPIX | T.EV_NORMALIZE, 10.0, 0.5, 2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): T.event_norm(img, T.EV_RM_TS | T.EV_HOTPIX | T.EV_NORMALIZE, 10.0, 0.5, 2)
e1.record(); torch.cuda.synchronize()
print("event_norm 256x224^2 u8 -> f32x2: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
PY
python /tmp/en_bench.py
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r03_en_$c -- python /tmp/en_bench.py > /dev/null 2>&1
  python tools/pmc_summary.py $(ls gpurun_out/r03_en_$c/*/*counter_collection.csv | head -1) $c > gpurun_out/r03_en_$c.json
  rm -rf gpurun_out/r03_en_$c
done
python tools/pmc_combine.py gpurun_out/r03_en_FETCH_SIZE.json gpurun_out/r03_en_WRITE_SIZE.json gpurun_out/r03_event_norm_traffic.json | grep -A5 event_norm
