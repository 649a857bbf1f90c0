#!/usr/bin/env python3
"""Static instruction mix of kernels in a hipcc -save-temps .s file: isa_stats.py file.s substring [substring ...]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n', s, re.M):
    name = m.group(1)
    if not any(k in name for k in sys.argv[2:]):
        continue
    j = s.index('.Lfunc_end', m.end())
    c = Counter()
    n = 0
    for l in s[m.end():j].split('\n'):
        l = l.strip()
        if not l or l[0] in '.;/' or l.endswith(':'):
            continue
        op = l.split()[0]
        n += 1
        kind = ('valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'vmem_ld' if 'load' in op and not op.startswith('ds_') else
                'vmem_st' if 'store' in op and not op.startswith('ds_') else 'lds' if op.startswith('ds_') else 'atomic' if 'atomic' in op else 'other')
        c[kind] += 1
        if 'dpp' in l: c['(dpp)'] += 1
        if op == 's_waitcnt': c['(waitcnt)'] += 1
        if op.startswith('s_cbranch'): c['(branch)'] += 1
    print(name[:60], n, dict(c))
