"""
Builds surreal_amd/libsurreal_amd.so (the C-ABI of include/surreal_amd.h) for gfx950 with
hipcc.  hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the tree snapshot.

    python -m surreal_amd.build          # incremental
    python -m surreal_amd.build --force
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsurreal_amd.so')
SOURCES = ['smx_scan.hip', 'smx_mlp3_fused.hip', 'smx_gemm.hip', 'smx_ppo.hip', 'smx_replay.hip',
           'smx_ddpg.hip', 'smx_lstm.hip', 'smx_conv.hip', 'smx_epoch.hip', 'smx_mlp3_rows16.hip', 'smx_xchg.hip',
           'smx_rollout.hip', 'smx_wgrad.hip', 'smx_mlp3_bwd16.hip', 'smx_ddpg_rows.hip']
# -ffp-contract=off: the reference issues separate ATen mul/add ops; contraction into FMAs would
# change roundings that the parity tests pin (the MFMA path is unaffected).
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-Wno-unused-result']


# e.g. SMX_EXTRA_FLAGS=-DSMX_EPOCH_TIMING python -m surreal_amd.build --force  (scripts/bench_epoch.py)
FLAGS += [f for f in os.environ.get('SMX_EXTRA_FLAGS', '').split() if f]


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libsurreal_amd.so')
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    cc = hipcc()
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')] + \
        [os.path.join(os.path.dirname(HERE), 'include', 'surreal_amd.h')]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            cmd = [cc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    if force or procs or _newer(LIB, objs):
        cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


HOST_SRC = os.path.join(CSRC, 'host', 'smx_host.c')
HOST_LIB = os.path.join(HERE, '_smx_host.so')


def build_host(force=False, verbose=True):
    """surreal_amd/_smx_host.so: the CPython extension that assembles host-fed batches in place (csrc/host/smx_host.c;
    gcc + OpenMP, CPython and numpy headers -- no device code)"""
    if not (force or _newer(HOST_LIB, [HOST_SRC])):
        return HOST_LIB
    import sysconfig
    import numpy
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc is None:
        raise RuntimeError('gcc not found: cannot build _smx_host.so')
    cmd = [cc, '-O3', '-fPIC', '-shared', '-fopenmp', '-Wall', '-I' + sysconfig.get_paths()['include'],
           '-I' + numpy.get_include(), HOST_SRC, '-o', HOST_LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    build_host(force='--force' in sys.argv)
    print(LIB)
    print(HOST_LIB)
