"""Builds libstx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m style_transfer_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared object lands next to the sources
(``style_transfer_amd/csrc/libstx.so``): it is git-ignored but travels with the tree.
"""

import concurrent.futures
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(REPO, 'include')
LIB = os.path.join(CSRC, 'libstx.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'

SOURCES = {
    'conv_mfma.hip': [],
    'conv_wino2.hip': [],
    'conv_first.hip': [],
    'conv_h2.hip': [],
    'gram.hip': [],
    'symm.hip': [],
    'pool.hip': [],
    'reduce.hip': [],
    # one rounding per float32 operation, like the reference's numpy expressions
    'image_ops.hip': ['-ffp-contract=off'],
    'engine.cpp': [],
}
EXTRA = os.environ.get('STX_HIPCC_EXTRA', '').split()
COMMON = EXTRA + ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-I' + INCLUDE, '-I' + CSRC,
          '-Wall', '-Wno-unused-function']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, extra, force):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
    path = os.path.join(CSRC, src)
    deps = [path, os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'bf16x3.h'),
            os.path.join(CSRC, 'f16x2.h'), os.path.join(INCLUDE, 'stx.h')]
    if force or _stale(obj, deps):
        cmd = [HIPCC] + COMMON + extra + ['-c', path, '-o', obj]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, proc.stdout))
        if proc.stdout.strip():
            sys.stderr.write(proc.stdout)
    return obj


def build(force=False, verbose=False):
    """Compiles every translation unit (in parallel) and links libstx.so.  Returns its path."""
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        futs = [ex.submit(_compile, src, extra, force) for src, extra in SOURCES.items()]
        objs = [f.result() for f in futs]
    if force or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise RuntimeError('link failed:\n' + proc.stdout)
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
