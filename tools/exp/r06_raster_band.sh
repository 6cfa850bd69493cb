# pixels per band of the two-pass rasterizer (pass-2 workgroup LDS = 4 B per pixel): 32764 (one 128 KB workgroup per CU) / 16380 / 8188
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for bp in 32764 16380 8188; do
  echo -n "rep $rep band_px $bp: "; RASTER_BINNED_ONLY=1 RASTER_BAND_PX=$bp python tools/raster_bench.py 2>/dev/null | grep uniform | sed 's/ *B=64.*1000000://'
done; done
for bp in 32764 16380; do
  RASTER_BINNED_ONLY=1 RASTER_BAND_PX=$bp rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rb$bp -- python tools/raster_bench.py > /dev/null 2>&1
  echo "band_px $bp:"; grep -h "raster_bin" gpurun_out/rb$bp/*/*kernel_stats.csv | cut -d, -f1-4; rm -rf gpurun_out/rb$bp
done
