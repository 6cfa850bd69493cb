# interleaved A/B of the window-attention forms at the config-#5 size (B = 64, 16 heads, 30 x 40 + cls): 1 = shipped (LDS-DMA staging,
# recomputing backward), 5 = register-staged forward, 9 = dS-storing backward (3.2 GB workspace)
for rep in 1 2 3; do
  WIN_TIME_ONLY=1 WIN_MODES=1,5,9 python tools/attn_win_check.py all time 2>&1 | grep "^mode" | awk -v r=$rep '{print "rep", r, $0}'
done
tools/attn_win_prof.sh 1,5,9
