"""Opcode histogram per basic block of one kernel in a hipcc -S file: isa_hist.py file.s kernel-substring [min-instructions]"""
import re, sys, collections
s = open(sys.argv[1]).read()
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for f in re.split(r'\n\t\.type\t', s)[1:]:
    name = f.split(',')[0]
    if sys.argv[2] not in name:
        continue
    print(name)
    lab = 'entry'; blocks = collections.OrderedDict()
    for l in f.split('\n'):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: lab = m.group(1)
        blocks.setdefault(lab, []).append(l)
    for lab, b in blocks.items():
        ops = collections.Counter()
        for l in b:
            m = re.match(r'^\t([a-z_0-9]+)', l)
            if m: ops[m.group(1)] += 1
        tot = sum(ops.values())
        if tot >= minn:
            print(' ', lab, tot)
            print('     ', ', '.join(f'{k}:{v}' for k, v in ops.most_common(40)))
