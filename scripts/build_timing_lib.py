"""builds surreal_amd/libsurreal_amd_timing.so: the library with the kernels' phase timestamps compiled in
(-DSMX_ROLLOUT_TIMING -DSMX_EPOCH_TIMING), next to the product library (objects in a scratch directory).  Load it with
SMX_LIB_PATH=.../libsurreal_amd_timing.so (scripts/bench_rollout.py, scripts/bench_epoch.py); never used by tests."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surreal_amd import build as B
flags = B.FLAGS + ['-DSMX_ROLLOUT_TIMING', '-DSMX_EPOCH_TIMING'] + sys.argv[1:]
cc, tmp = B.hipcc(), tempfile.mkdtemp()
out = os.path.join(ROOT, 'surreal_amd', 'libsurreal_amd_timing.so')
procs, objs = [], []
for src in B.SOURCES:
    o = os.path.join(tmp, src.replace('.hip', '.o'))
    objs.append(o)
    procs.append(subprocess.Popen([cc] + flags + ['-c', os.path.join(B.CSRC, src), '-o', o]))
assert all(p.wait() == 0 for p in procs)
subprocess.check_call([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
print('built', out)
