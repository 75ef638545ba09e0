"""Per-dispatch HBM fetch / write of ONE tile evaluation, from rocprofv3 --pmc passes over
tools/bench_tile.py (FETCH_SIZE in one run, WRITE_SIZE in another; --kernel-trace gives the grid).

    python tools/pmc_layers.py fetch_counter.csv write_counter.csv [fetch_factor]

Prints the dispatches of the last tile evaluation in launch order.  fetch_factor: the gfx950
correction of FETCH_SIZE measured on the Adam kernel of bench.py (tools/pmc_traffic.py; 2.0).
"""
import collections
import csv
import re
import sys


def per_dispatch(path, counter):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        rows[int(r['Dispatch_Id'])] = (r['Kernel_Name'], float(r['Counter_Value']) * 1024,
                                       int(r.get('Grid_Size', 0) or 0), int(r.get('Workgroup_Size', 0) or 0))
    return rows


def short(k):
    k = re.sub(r'^void ', '', k)
    k = re.sub(r'\(.*', '', k)
    return k.replace('stx::', '')[:58]


def main(fetch_csv, write_csv, factor=2.0):
    fetch = per_dispatch(fetch_csv, 'FETCH_SIZE')
    write = per_dispatch(write_csv, 'WRITE_SIZE')
    ids = sorted(fetch)
    # a tile evaluation ends with the data gradient of the first layer
    ends = [i for i in ids if 'conv3x3_m4' in fetch[i][0]]
    lo, hi = ends[-2], ends[-1]
    tf = tw = 0.0
    print('%-58s %8s %10s %10s' % ('kernel', 'groups', 'fetch MB', 'write MB'))
    for i in ids:
        if i <= lo or i > hi:
            continue
        k, f, g, wg = fetch[i]
        w = write.get(i, (k, 0.0, 0, 0))[1]
        f *= factor
        tf += f
        tw += w
        print('%-58s %8d %10.1f %10.1f' % (short(k), g // max(wg, 1), f / 1e6, w / 1e6))
    print('total fetch %.1f MB  write %.1f MB' % (tf / 1e6, tw / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], *[float(a) for a in sys.argv[3:4]])
