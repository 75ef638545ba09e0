"""MFMA utilisation per kernel from one rocprofv3 --kernel-trace --pmc pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv ...
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMD matrix pipes of the chip (a
v_mfma_f32_32x32x2_f32 holds its pipe for 64 cycles), GRBM_GUI_ACTIVE sums the active clocks of
the 8 XCDs; utilisation = busy / (gui_active / 8 * 1024)."""
import collections
import csv
import re
import sys


def main(counter_csv, trace_csv):
    dur = {}
    for r in csv.DictReader(open(trace_csv)):
        dur[int(r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    c = collections.defaultdict(dict)
    name = {}
    for r in csv.DictReader(open(counter_csv)):
        c[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        name[int(r['Dispatch_Id'])] = r['Kernel_Name']
    agg = collections.OrderedDict()
    for i, v in c.items():
        k = re.sub(r'\(.*', '', name[i]).replace('void ', '')[:64]
        a = agg.setdefault(k, [0, 0.0, 0.0, 0])
        a[0] += 1
        a[1] += v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        a[2] += v.get('GRBM_GUI_ACTIVE', 0.0)
        a[3] += dur.get(i, 0)
    print('%-66s %6s %10s %9s %9s' % ('kernel', 'calls', 'total_ms', 'MHz', 'mfma_util'))
    for k, (n, busy, gui, ns) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
        if gui <= 0 or ns <= 0:
            continue
        print('%-66s %6d %10.3f %9.0f %9.3f' % (k, n, ns / 1e6, gui / 8 / ns * 1e3, busy / (gui / 8 * 1024)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
