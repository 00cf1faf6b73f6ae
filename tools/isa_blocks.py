#!/usr/bin/env python
"""Basic-block instruction mix of one kernel in a hipcc -S listing (all blocks above a size, with opcode histogram on request).
    tools/isa_blocks.py /tmp/isa/kernels.s 'k_noise_filter_olaILi10ELb1' [min_instr] [--ops]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
name = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 40
ops = '--ops' in sys.argv
m = re.search(r'^(_Z\w*%s\w*):' % name, s, re.M)
end = s.index('.Lfunc_end', m.end())
body = s[m.end(): end]
tail = s[end: end + 6000]
for key in ('.sgpr_count', '.vgpr_count', 'ScratchSize', 'Occupancy', 'LDSByteSize', '.vgpr_spill_count', 'NumVgprs', 'NumAgprs'):
    mm = re.search(r'%s:?\s*(\d+)' % re.escape(key), tail)
    if mm: print(key, mm.group(1), end='  ')
print()
parts = re.split(r'^(\.LBB\d+_\d+):', body, flags=re.M)
def cls(i):
    if i.startswith('v_mfma'): return 'mfma'
    if i.startswith(('buffer_', 'global_', 'flat_')): return 'vmem'
    if i.startswith('scratch_'): return 'scratch'
    if i.startswith('ds_'): return 'lds'
    if i == 's_waitcnt': return 'wait'
    if i == 's_barrier': return 'barrier'
    if i == 's_nop': return 'nop'
    if i.startswith('s_'): return 'salu'
    if i.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')): return 'trans'
    if i.startswith('v_pk_'): return 'vpk'
    if 'f64' in i: return 'v64'
    return 'valu'
tot = Counter()
blocks = [('entry', parts[0])] + [(parts[k], parts[k + 1]) for k in range(1, len(parts), 2)]
for lab, b in blocks:
    ins = [l.split()[0] for l in b.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter(cls(i) for i in ins); tot.update(c)
    if len(ins) < mn: continue
    print(lab, 'instr', len(ins), dict(c))
    if ops:
        print('   ', Counter(ins).most_common(40))
print('TOTAL', dict(tot))
