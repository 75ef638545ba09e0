"""Where a step of bench.py's loop spends its time, from a rocprofv3 --kernel-trace CSV:

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python bench.py --steps 8 --warmup 2 \
        --no-cpu-baseline --no-wall-clock --steady-seconds 0
    python tools/step_timeline.py /tmp/tl/.../*_kernel_trace.csv

A step ends with step_stats_kernel (the host waits for its result).  Per step: its length (end of
the previous step's statistics kernel to the end of this one's), the time NO kernel was running
(union of all dispatch intervals over all streams), the idle gap at its start (statistics kernel
-> first kernel of the next step: host synchronisation + the host queueing the first launches),
and the serial image-op phases before the first / after the last tile kernel.
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: re.sub(r'\(.*', '', re.sub(r'^void ', '', r['Kernel_Name'])).replace('stx::', '')
ends = [i for i, r in enumerate(rows) if 'step_stats' in r['Kernel_Name']]
IMAGE = ('tile_move', 'regularizers', 'adam', 'step_stats', 'reduce_final', 'dsum', 'final')
print('%5s %9s %9s %9s %9s %9s %9s' % ('step', 'len ms', 'idle ms', 'start gap', 'pre-tile', 'post-tile', 'kernels'))
tot = []
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    t0 = int(rows[a]['End_Timestamp'])
    t1 = int(rows[b]['End_Timestamp'])
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    is_img = [any(k in r['Kernel_Name'] for k in IMAGE) for r in seg]
    first_tile = next(i for i, f in enumerate(is_img) if not f)
    last_tile = len(seg) - 1 - next(i for i, f in enumerate(reversed(is_img)) if not f)
    pre = int(seg[first_tile]['Start_Timestamp']) - iv[0][0]
    post = t1 - max(int(r['End_Timestamp']) for r in seg[:last_tile + 1])
    row = ((t1 - t0) / 1e6, (t1 - t0 - busy) / 1e6, (iv[0][0] - t0) / 1e6, pre / 1e6, post / 1e6, len(seg))
    tot.append(row)
    print('%5d %9.3f %9.3f %9.3f %9.3f %9.3f %9d' % ((len(tot),) + row))
n = len(tot)
if n:
    print('%5s %9.3f %9.3f %9.3f %9.3f %9.3f' % (('mean',) + tuple(sum(r[i] for r in tot) / n for i in range(5))))
    seg = rows[ends[-2] + 1:ends[-1] + 1]
    print('\nimage-op kernels of the last step:')
    for r in seg:
        if any(k in r['Kernel_Name'] for k in IMAGE):
            print('  %-50s %8.1f us' % (name(r)[:50], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
