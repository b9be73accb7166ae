"""builds surreal_amd/libsurreal_amd_<name>.so: the product library with extra -D flags on SOME sources (A/B runs of a
kernel variant on one GPU box, next to the product library; the other objects are the product build's).  Load it with
SMX_LIB_PATH=surreal_amd/libsurreal_amd_<name>.so.  Never used by tests or by bench.py.

    python scripts/build_variant_lib.py prio smx_mlp3_rows16.hip -DSMX_FUSED_PRIO=1
"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surreal_amd import build as B
name = sys.argv[1]
srcs = [a for a in sys.argv[2:] if a.endswith('.hip')]
flags = B.FLAGS + [a for a in sys.argv[2:] if not a.endswith('.hip')]
B.build(verbose=False)
cc, tmp = B.hipcc(), tempfile.mkdtemp()
objs, procs = [], []
for src in B.SOURCES:
    if src in srcs:
        o = os.path.join(tmp, src.replace('.hip', '.o'))
        procs.append(subprocess.Popen([cc] + flags + ['-c', os.path.join(B.CSRC, src), '-o', o]))
    else:
        o = os.path.join(B.CSRC, src.replace('.hip', '.o'))
    objs.append(o)
assert all(p.wait() == 0 for p in procs)
out = os.path.join(ROOT, 'surreal_amd', 'libsurreal_amd_%s.so' % name)
subprocess.check_call([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
print('built', out)
