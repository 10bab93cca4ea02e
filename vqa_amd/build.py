"""Build libvqa_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m vqa_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libvqa_amd.so')
SOURCES = ['schedule.cpp', 'tree.cpp', 'prof.cpp', 'feed.cpp', 'conv.hip', 'conv_f32.hip', 'wgrad.hip', 'gemm.hip', 'misc.hip', 'adam.hip', 'devsched.hip', 'head.hip']
HEADERS = ['vqa_dev.h', 'plan.h', 'prof.h', os.path.join('..', '..', 'include', 'vqa_amd.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wall', '-Wno-unused-function']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _stale(OUT, [os.path.join(CSRC, s) for s in SOURCES] + hdrs):
        return OUT          # up to date (object files need not exist: they do not travel)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + (['-x', 'hip'] if s.endswith('.cpp') else []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('compiling %s failed' % s)
    if force or procs or _stale(OUT, objs):
        cmd = [HIPCC, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', OUT] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
