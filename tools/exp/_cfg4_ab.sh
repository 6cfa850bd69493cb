run() { python bench.py "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config4_end_to_end',{})
print(sys.argv[1:], d.get('ms_per_step'), d.get('ab_value'), c.get('ms_per_step'), c.get('rasterizer_ms_per_step'), (d.get('entrypoint') or {}).get('ms_per_step'))" "$LABEL"; }
LABEL=new_full run
export MEMHIP_LIB=variants/old.so
LABEL=old_full run
unset MEMHIP_LIB
LABEL=new_full_no5 run --no-config5-figure
