"""Instruction mix of a kernel's prologue / main loop / epilogue from hipcc -S output
(split at the first and last v_mfma).   python tools/asm_mix.py file.s [name-substring]"""
import collections
import re
import sys


def kind(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')): return 'vmem'
    return 'other'


def main(path, pat=''):
    name, body = None, []
    funcs = []
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name, body = m.group(1), []
            funcs.append((name, body))
        elif name and line.startswith('.Lfunc_end'):
            name = None
        elif name:
            t = line.strip()
            if t and not t.startswith((';', '.')):
                body.append(t.split()[0])
    for name, ops in funcs:
        if pat not in name:
            continue
        idx = [j for j, o in enumerate(ops) if o.startswith('v_mfma')]
        if not idx:
            continue
        parts = (('prologue', ops[:idx[0]]), ('loop', ops[idx[0]:idx[-1] + 1]), ('epilogue', ops[idx[-1] + 1:]))
        print(name)
        for label, seg in parts:
            c = collections.Counter(kind(o) for o in seg)
            print('   %-9s %5d  %s' % (label, len(seg), dict(c)))


if __name__ == '__main__':
    main(*sys.argv[1:3])
