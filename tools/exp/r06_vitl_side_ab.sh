# ViT-L/16 480x640 (config #5), B = 64: the weight-gradient side stream on / off, interleaved; + register-staged forward check
for rep in 1 2 3; do
  for ns in 0 1; do
    MEMHIP_NO_SIDE=$ns python tools/bench_vitl.py 64 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep no_side=$ns', d['ms_per_step'], 'ms', d['samples_per_sec'], 'samples/s')"
  done
done
WIN_MODES=1,5 python tools/attn_win_check.py fwd 2>&1 | grep -E "mode 5 (out|lse)"
