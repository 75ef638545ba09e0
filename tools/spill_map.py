#!/usr/bin/env python3
"""Where in a kernel's instruction stream do the spills sit?  One character per event of the ISA of one
function of a `hipcc -S --offload-device-only` listing: m MFMA (runs compressed), S / L scratch store / load,
| barrier, > branch, labels on their own lines.
    python tools/spill_map.py listing.s <mangled-name-substring>"""
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith('_ZN') and key in l and ': ' in l)
end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
out = []
for l in txt[start:end]:
    t = l.strip()
    if t.startswith('scratch_'):
        out.append('S' if 'store' in t else 'L')
    elif 'v_mfma' in t:
        out.append('m')
    elif t.startswith('s_barrier'):
        out.append('|')
    elif re.match(r'\.LBB\d+_\d+:', t):
        out.append('\n' + t.split(':')[0] + ' ')
    elif t.startswith('s_cbranch') or t.startswith('s_branch'):
        out.append('>' + t.split()[-1] + ' ')
    elif t.startswith('buffer_load'):
        out.append('g')
    elif t.startswith('buffer_store'):
        out.append('w')
s = ''.join(out)
s = re.sub(r'(m[gw]*){12,}', lambda m: '[%dm]' % m.group(0).count('m'), s)
print(txt[start].split(':')[0], end - start, 'lines')
print(s)
