#!/usr/bin/env python3
"""Table of hipcc -Rpass-analysis=kernel-resource-usage remarks: resusage.py res.txt [name-filter]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows = []
cur = None
for l in txt.split('\n'):
    m = re.search(r'remark: +(.*?) \[-Rpass', l)
    if not m: continue
    s = m.group(1).strip()
    if s.startswith('Function Name:'):
        cur = {'name': s.split(':', 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ':' in s:
        k, v = s.rsplit(':', 1)
        cur[k.strip()] = v.strip()
def dem(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip().split('(')[0]
    except Exception:
        return n
print('%-44s %5s %5s %7s %7s %7s %4s %7s' % ('kernel', 'SGPR', 'VGPR', 'sSpill', 'vSpill', 'scratch', 'occ', 'LDS'))
for r in rows:
    n = dem(r['name'])
    if flt and flt not in n: continue
    print('%-44s %5s %5s %7s %7s %7s %4s %7s' % (n[:44], r.get('TotalSGPRs', r.get('SGPRs')), r.get('VGPRs'), r.get('SGPRs Spill'), r.get('VGPRs Spill'),
          r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
