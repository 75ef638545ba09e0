"""HBM traffic per tile-iteration from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
bench.py.  Both counters are in KiB-units of 1024 B per the rocprofv3 derived-metric definition.

Calibration (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE may report 1/2 of the bytes of a wide
coalesced stream; other access widths are uncalibrated): the fused Adam kernel of the same run
has exactly known traffic -- it reads params, grad, g1, g2, p1 and writes params, g1, g2, p1, avg
(5 + 5 arrays of 3*H*W floats) with 4-byte-per-lane accesses like the rest of this code base --
so measured/expected on that kernel gives the correction factors applied to every other kernel.
"""
import collections
import csv
import json
import re
import sys


def per_dispatch(path, counter):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        rows[int(r['Dispatch_Id'])] = (r['Kernel_Name'], float(r['Counter_Value']))
    return rows


def main(fetch_csv, write_csv, size=2048, tiles_per_step=4):
    """size: edge of the (square) image the Adam kernel runs on; tiles_per_step: tile evaluations
    between two Adam launches (bench.py at N = 1: 2048, 4)."""
    fetch = per_dispatch(fetch_csv, 'FETCH_SIZE')
    write = per_dispatch(write_csv, 'WRITE_SIZE')
    out = {}
    for name, table in (('fetch', fetch), ('write', write)):
        ids = sorted(table)
        adam = [i for i in ids if 'adam_kernel' in table[i][0]]
        lo, hi = adam[-2], adam[-1]                     # one full step between two Adam launches
        step = [table[i] for i in ids if lo < i <= hi]
        expected_adam = 5 * 3 * size * size * 4
        adam_bytes = table[hi][1] * 1024
        factor = expected_adam / adam_bytes
        tile = [(k, v) for k, v in step if not re.search(
            r'adam_kernel|regularizers|tile_move|step_stats|finish_partials|copyBuffer|fillBuffer', k)]
        raw = sum(v for _, v in tile) * 1024 / tiles_per_step
        out[name] = {'adam_measured_bytes': adam_bytes, 'adam_expected_bytes': expected_adam,
                     'correction': factor, 'tile_raw_bytes': raw, 'tile_bytes': raw * factor,
                     'kernels': len(tile)}
        by = collections.Counter()
        for k, v in tile:
            short = re.sub(r'\(.*', '', k)
            short = re.sub(r'^void ', '', short)
            by[short[:70]] += v * 1024 * factor / tiles_per_step
        out[name]['by_kernel_MB'] = {k: round(v / 1e6, 1) for k, v in by.most_common(8)}
    out['hbm_bytes_per_tile_iteration'] = out['fetch']['tile_bytes'] + out['write']['tile_bytes']
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], *[int(a) for a in sys.argv[3:5]])
