#!/usr/bin/env python3
"""Instruction census of one kernel in a hipcc -S listing: isa_count.py listing.s name-substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = None
for mm in re.finditer(r'^(_Z\S+):', s, re.M):
    if pat in mm.group(1):
        m = mm; break
assert m, "kernel not found"
end = s.index('.end_amdhsa_kernel', m.end()) if '.end_amdhsa_kernel' in s[m.end():] else len(s)
body = s[m.end():end]
body = body.split('s_endpgm')[0]
lines = [l.strip() for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
c = Counter(l.split()[0] for l in lines)
tot = sum(c.values())
def grp(pred): return sum(v for k, v in c.items() if pred(k))
print(m.group(1))
print('total', tot, '| VALU', grp(lambda k: k.startswith('v_')), '| DPP', sum(1 for l in lines if 'quad_perm' in l or 'row_' in l),
      '| LDS', grp(lambda k: k.startswith('ds_')), '| VMEM', grp(lambda k: k.startswith(('buffer_', 'global_', 'scratch_'))),
      '| scratch', grp(lambda k: k.startswith('scratch_')), '| SALU', grp(lambda k: k.startswith('s_')), '| waitcnt', c.get('s_waitcnt', 0), '| barrier', c.get('s_barrier', 0))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30): print('   ', k, v)
