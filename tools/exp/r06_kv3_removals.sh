# removal experiments on the dK / dV window kernel (wrong results; timing only): variants/kv3e<n>.so built with -DKV3_EXP=n
for e in 0 1 2 3 4 5 6 7; do
  lib=""; [ $e != 0 ] && lib=variants/kv3e$e.so
  echo "== KV3_EXP=$e"; MEMHIP_LIB=$lib WIN_MODES=3 python tools/attn_win_check.py all time 2>&1 | grep "^mode" | tail -1
done
