"""Per-dispatch durations (and gaps to the previous dispatch) of the LAST tile evaluation in a
rocprofv3 --kernel-trace CSV of tools/bench_tile.py.   python tools/trace_layers.py kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'conv3x3_m4' in r['Kernel_Name']]
lo, hi = ends[-2], ends[-1]
prev_end = int(rows[lo]['End_Timestamp'])
tot = gap_tot = 0.0
print('%-60s %9s %9s' % ('kernel', 'us', 'gap us'))
for r in rows[lo + 1:hi + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', re.sub(r'^void ', '', r['Kernel_Name'])).replace('stx::', '')[:58]
    print('%-60s %9.1f %9.1f' % (name, (e - s) / 1e3, (s - prev_end) / 1e3))
    tot += (e - s) / 1e3
    gap_tot += max(0, s - prev_end) / 1e3
    prev_end = e
print('kernels %.1f us, gaps %.1f us' % (tot, gap_tot))
