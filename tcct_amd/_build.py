"""Build libtcct_hip.so (all HIP kernels + the C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with gpurun snapshots."""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libtcct_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    # (*.inc: loss_classes.inc is compiled three times by loss.hip -- left out of this list until round 6, an edit of it alone did not rebuild loss.o)
    hdrs = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + sorted(glob.glob(os.path.join(CSRC, '*.inc'))) + [os.path.join(CSRC, '..', '..', 'include', 'tcct_hip.h')]
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
            if len(procs) >= 4:
                _wait(procs)
    _wait(procs)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def _wait(procs):
    while procs:
        s, p = procs.pop(0)
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + s)


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
