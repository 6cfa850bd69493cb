common="--no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-entrypoint-figure --no-gemm-timer --steps 20 --warmup 5"
run() { MEMHIP_LIB="$1" python bench.py $common $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', d['ms_per_step'], d['ms_per_step_p50'])"; }
for i in 1 2 3; do
run mem_amd/exp/base.so "" base
run "" "--opt gemm_prefetch=0" new_pf0
run "" "--opt gemm_prefetch=0 --opt gemm_stagger=300" new_pf0_stag300
run "" "--opt gemm_prefetch=0 --opt gemm_stagger=-300" new_pf0_stagall300
run mem_amd/exp/base.so "--opt gemm_stagger=300" base_stag300
done
