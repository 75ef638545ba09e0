"""Per-kernel summary (calls, total, average, share) from a rocprofv3 --kernel-trace CSV.
    python tools/kernel_stats.py <kernel_trace.csv> [tile evaluations in the trace]
With the second argument the totals are also shown per tile evaluation."""
import collections
import csv
import re
import sys


def main(path, per=1):
    agg = collections.OrderedDict()
    t_min, t_max = None, 0
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:72]
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
        t_min = s if t_min is None else min(t_min, s)
        t_max = max(t_max, e)
    total = sum(v[1] for v in agg.values())
    print('%-74s %7s %11s %10s %7s %12s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'share', 'ms_per_unit'))
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-74s %7d %11.3f %10.2f %6.1f%% %12.4f' % (name, n, ns / 1e6, ns / n / 1e3,
                                                          100.0 * ns / total, ns / 1e6 / per))
    print('%-74s %7d %11.3f %10s %7s %12.4f' % ('SUM of kernel durations', sum(v[0] for v in agg.values()),
                                                 total / 1e6, '', '', total / 1e6 / per))
    print('trace span: %.3f ms' % ((t_max - t_min) / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
