for e in 0 1 2 3; do lib=""; [ $e != 0 ] && lib=variants/kvs$e.so
  echo "== KVS_EXP=$e"; MEMHIP_LIB=$lib WIN_TIME_ONLY=1 WIN_MODES=9 python tools/attn_win_check.py all time 2>&1 | grep "^mode" | tail -1; done
