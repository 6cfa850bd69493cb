"""Instruction-class sequence of a label range of one kernel in a hipcc -S file, runs compressed:
isa_seq.py file.s kernel-substring first-label last-label   (classes: M mfma, V valu, X v_exp, L ds read, W s_waitcnt, B barrier, G vmem)"""
import re, sys
s = open(sys.argv[1]).read()
f = [x for x in re.split(r'\n\t\.type\t', s)[1:] if sys.argv[2] in x.split(',')[0]][0]
on = False; seq = []
for l in f.split('\n'):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        if m.group(1) == sys.argv[3]: on = True
        if m.group(1) == sys.argv[4]: break
        if on: seq.append('|' + m.group(1) + '|')
        continue
    if not on: continue
    m = re.match(r'^\t([a-z_0-9]+)', l)
    if not m: continue
    o = m.group(1)
    if o.startswith('v_mfma'): c = 'M'
    elif o in ('v_exp_f32_e32',): c = 'X'
    elif o.startswith('v_'): c = 'V'
    elif o.startswith('ds_'): c = 'L'
    elif o == 's_waitcnt': c = 'W'
    elif o == 's_barrier': c = 'B'
    elif o.startswith('global_') or o.startswith('buffer_') or o.startswith('scratch_'): c = 'G'
    elif o == 's_nop': c = 'n'
    else: c = 's'
    seq.append(c)
out = []; prev = None; n = 0
for c in seq + [None]:
    if c == prev and c is not None and not c.startswith('|'): n += 1
    else:
        if prev is not None: out.append(prev if n == 1 or prev.startswith('|') else f'{prev}{n}')
        prev = c; n = 1
print(' '.join(out))
