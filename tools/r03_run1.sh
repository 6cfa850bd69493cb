python -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_augment_gpu.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r03b_tests.log
tail -3 gpurun_out/r03b_tests.log
for i in 1 2; do
echo "== base"; MEMHIP_LIB=mem_amd/exp/base.so python tools/epi_probe.py fc1 fc2 proj qkv
echo "== new prefetch=1"; python tools/epi_probe.py fc1 fc2 proj qkv
echo "== new prefetch=0"; OPTS=gemm_prefetch=0 python tools/epi_probe.py fc1 fc2 proj qkv
done > gpurun_out/r03b_probe.log 2>&1
cat gpurun_out/r03b_probe.log
bash tools/ab_lib.sh mem_amd/exp/base.so "" 2 > gpurun_out/r03b_ab.log 2>&1; cat gpurun_out/r03b_ab.log
