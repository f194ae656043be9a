#!/usr/bin/env python
"""developer tool: instruction mix per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only)"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S):
    name, body = m.group(1), m.group(2)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//'))]
    c = Counter()
    for l in lines:
        op = l.split()[0]
        k = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'ds' if op.startswith('ds_') else 'waitcnt' if op.startswith('s_waitcnt')
             else 'barrier' if op.startswith('s_barrier') else 'vmem' if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')) else 'salu' if op.startswith('s_') else 'other')
        c[k] += 1
    extra = {k: sum(1 for l in lines if l.startswith(k)) for k in ('v_mov', 'v_pk', 'v_cndmask', 'v_readfirstlane', 's_nop', 'scratch_')}
    print(name[:70], dict(c), 'total', len(lines), extra)
