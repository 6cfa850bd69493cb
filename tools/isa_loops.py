"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (blocks with >= MIN instructions).
usage: isa_loops.py file.s kernel_name_substring [min]"""
import collections, re, sys
s = open(sys.argv[1]).read(); name = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
i = s.index(name); i = s.index('\n', s.index(name + 'E:', i) if (name + 'E:') in s[i:] else i)
body = s[i:]; body = body[:body.index('s_endpgm')]
blocks = []; cur = ['entry', collections.Counter()]; blocks.append(cur)
for l in body.splitlines():
    if re.match(r'^\.LBB\d+_\d+:', l):
        cur = [l.strip()[:60], collections.Counter()]; blocks.append(cur)
    else:
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if m: cur[1][m.group(1)] += 1
for b in blocks:
    n = sum(b[1].values())
    if n >= mn:
        valu = sum(v for k, v in b[1].items() if k.startswith('v_') and not k.startswith('v_mfma'))
        tr = sum(v for k, v in b[1].items() if k in ('v_exp_f32_e32', 'v_rcp_f32_e32', 'v_log_f32_e32'))
        mf = sum(v for k, v in b[1].items() if k.startswith('v_mfma'))
        print(f"{b[0]}  total {n}  valu {valu} (8-cycle {tr})  mfma {mf}  ~valu-issue cycles {4 * (valu - tr) + 8 * tr + 8 * mf}")
        print('    ' + '  '.join(f"{k} {v}" for k, v in b[1].most_common(30)))
