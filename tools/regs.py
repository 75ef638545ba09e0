#!/usr/bin/env python3
"""Register / spill table of the kernels in a hipcc log made with -Rpass-analysis=kernel-resource-usage.
    python tools/regs.py build.log [name-filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
blocks = txt.split('Function Name: ')[1:]
rows = []
for b in blocks:
    name = b.split()[0]
    if flt not in name:
        continue
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1))
    rows.append((name, g(' VGPRs'), g('AGPRs'), g('TotalSGPRs'), g('VGPRs Spill'), g('SGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'),
                 g('Occupancy \[waves/SIMD\]')))
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for n, r in zip(names, rows):
    n = re.sub(r'^void stx::', '', n).replace('(stx::WinoArgs)', '')
    print('%-48s VGPR %3d AGPR %3d SGPR %3d  spilled VGPR %2d SGPR %2d  scratch %3d B  occupancy %d' % ((n[:48],) + r[1:]))
