"""Per-basic-block census of a kernel in a hipcc -save-temps .s file: MFMAs, stores, LDS-DMA, barriers, counted waits, scratch.
usage: isa_blocks.py file.s kernel-substring"""
import re, sys
s = open(sys.argv[1]).read()
funcs = re.split(r'\n\t\.type\t', s)
for f in funcs[1:]:
    name = f.split(',')[0]
    if sys.argv[2] not in name:
        continue
    lines = f.split('\n')
    lab = 'entry'; stats = {}; order = []
    for l in lines:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: lab = m.group(1)
        if lab not in stats:
            stats[lab] = dict(n=0, mfma=0, scr=0, st=0, ld=0, glds=0, bar=0, vm=[], depth=''); order.append(lab)
        d = stats[lab]; d['n'] += 1
        if 'v_mfma' in l: d['mfma'] += 1
        if 'scratch_' in l: d['scr'] += 1
        if re.search(r'global_store|buffer_store', l): d['st'] += 1
        if re.search(r'global_load_dword|global_load_ushort|global_load_ubyte', l) and 'lds' not in l: d['ld'] += 1
        if 'global_load_lds' in l: d['glds'] += 1
        if 'global_atomic' in l: d['st'] += 0; d.setdefault('atom', 0); d['atom'] += 1
        if 's_barrier' in l: d['bar'] += 1
        m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
        if m: d['vm'].append(int(m.group(1)))
        m = re.search(r'Depth[= ](\d)', l)
        if m and not d['depth']: d['depth'] = m.group(1)
    print(name, 'lines', len(lines))
    for lab in order:
        d = stats[lab]
        if d['mfma'] or d['scr'] or d['st'] or d['bar'] or d['glds'] or d['vm'] or d['ld']:
            print(' ', lab, {k: v for k, v in d.items() if v})
