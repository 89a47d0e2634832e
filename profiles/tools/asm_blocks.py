#!/usr/bin/env python3
"""Basic-block instruction mix of one kernel in a hipcc -S listing.
usage: asm_blocks.py lib.s <mangled-name-prefix>  -> per block: label, VALU/SALU/LDS/VMEM/SMEM/branch counts, readlane/writelane"""
import re, sys
src, pref = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and l.rstrip().endswith(':') or (l.startswith(pref) and ':' in l))
blocks, cur = [], ['<entry>', {}]
def cls(op):
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith(('s_load', 's_buffer_load')): return 'smem'
    if op.startswith(('s_cbranch', 's_branch')): return 'br'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    return 'other'
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith('.Lfunc_end'): break
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        blocks.append(cur); cur = [m.group(1), {}]; continue
    if not s or s.startswith((';', '.')): continue
    op = s.split()[0]
    c = cls(op)
    cur[1][c] = cur[1].get(c, 0) + 1
    if c == 'br': cur[1].setdefault('targets', []).append(s.split()[-1])
blocks.append(cur)
tot = {}
for name, d in blocks:
    t = d.pop('targets', [])
    print('%-12s' % name, ' '.join('%s=%d' % kv for kv in sorted(d.items())), '->', ','.join(t))
    for k, v in d.items(): tot[k] = tot.get(k, 0) + v
print('TOTAL', tot)
