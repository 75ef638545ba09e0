#!/usr/bin/env python3
"""Table of an A/B file made by the tools/r06_*ab*.sh scripts (tools/ubench/h2conv_bench.hip output under '## name: env args' heads).
    python tools/h2ab_table.py gpurun_out/<dir>/ab.txt"""
import re
import sys
from collections import OrderedDict

res = OrderedDict()
for blk in open(sys.argv[1]).read().split('## ')[1:]:
    head, _, body = blk.partition('\n')
    m = re.search(r'((?:\(pooled[^)]*\) )?(?:fwd|bwd) K +\d+ M +\d+ +\d+x\d+ +MB \d): ([\d.]+) ms.*err ([\d.e+-]+) of max \((\d+) bad\)  max \|y\| (\w+)', body)
    if not m:
        print('??', head, body[:200])
        continue
    w = re.search(r'wave 0: prologue +(\d+) +chunk loop +(\d+) \((\d+) per chunk\) +epilogue +(\d+)', body)
    name, spec = head.split(':', 1)
    key = m.group(1) + (' INJECT' if 'INJECT=1' in spec else '')
    res.setdefault(key, OrderedDict()).setdefault(name.strip(), []).append(
        (float(m.group(2)), m.group(3), m.group(5) + ('' if m.group(4) == '0' else ' BAD'), w.groups() if w else ()))
for key, v in res.items():
    print(key)
    for name, rows in v.items():
        ms = [r[0] for r in rows]
        print('    %-8s ms %s  (median %.3f)  err %s %s  cycles prologue/loop/per chunk/epilogue %s'
              % (name, ' '.join('%.3f' % x for x in ms), sorted(ms)[len(ms) // 2], rows[0][1], rows[0][2],
                 ' | '.join('/'.join(r[3]) for r in rows)))
