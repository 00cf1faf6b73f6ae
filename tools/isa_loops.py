#!/usr/bin/env python
"""Instruction mix of the MFMA-carrying basic blocks of one kernel in a hipcc -S listing.
    tools/isa_loops.py /tmp/kernels.s k_harm_speech_tile [min_mfma]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
name = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = re.search(r'^(_Z\w*%s\w*):' % name, s, re.M)
body = s[m.end(): s.index('.Lfunc_end', m.end())]
parts = re.split(r'^(\.LBB\d+_\d+):', body, flags=re.M)
for k in range(1, len(parts), 2):
    b = parts[k + 1]
    ins = [l.split()[0] for l in b.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    nm = sum(i.startswith('v_mfma') for i in ins)
    if nm < mn: continue
    def cls(i):
        if i.startswith('v_mfma'): return 'mfma'
        if i.startswith(('buffer_', 'global_', 'flat_')): return 'vmem'
        if i.startswith('ds_'): return 'lds'
        if i.startswith('s_'): return 'salu:' + i if i in ('s_nop', 's_waitcnt') else 'salu'
        return 'valu'
    c = Counter(cls(i) for i in ins)
    # MFMA <-> other switches
    sw = sum(1 for a, b2 in zip(ins, ins[1:]) if a.startswith('v_mfma') != b2.startswith('v_mfma'))
    print(parts[k], 'instr', len(ins), dict(c), 'mfma/other switches', sw)
