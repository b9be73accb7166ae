#!/usr/bin/env python
"""
TEST INFRASTRUCTURE ONLY -- the recipe for oracle/_ref/ (git-ignored; it ships to the GPU box with the tree snapshot
like the built .so files do, and is never imported by surreal_amd/).

The reference is pure Python, so "building the reference where it lies" means byte-compiling it: every module of
/root/reference/surreal/ is compiled, from the source where it lies, to a SOURCELESS .pyc under oracle/_ref/surreal/
(py_compile; same interpreter here and on the GPU box: this image's /usr/bin/python3).  No reference source is
copied anywhere.  With oracle/_ref/ present, oracle/ref_shims.py can import the reference's own learner on a host
where /root/reference does not exist -- which is what lets bench.py time the REFERENCE'S code (cpu_baseline.kind
"reference") on the GPU box's cores next to the restatement (kind "port").

    python oracle/make_ref.py            # idempotent; a no-op (keeps what is there) when /root/reference is absent
"""
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('SURREAL_REFERENCE_SRC', '/root/reference')
OUT = os.path.join(HERE, '_ref')


def build(verbose=True):
    src_root = os.path.join(REFERENCE, 'surreal')
    if not os.path.isdir(src_root):
        if verbose:
            print('oracle/make_ref.py: %s absent -- keeping the prebuilt oracle/_ref (%s)'
                  % (src_root, 'present' if os.path.isdir(os.path.join(OUT, 'surreal')) else 'ABSENT'))
        return None
    stamp = os.path.join(OUT, 'BUILD_INFO')
    tag = '%s python %d.%d magic %r' % (src_root, sys.version_info[0], sys.version_info[1],
                                         __import__('importlib.util').util.MAGIC_NUMBER)
    if os.path.exists(stamp) and open(stamp).read().splitlines()[:1] == [tag]:
        return OUT
    shutil.rmtree(OUT, ignore_errors=True)
    n = skipped = 0
    for d, _, files in os.walk(src_root):
        rel = os.path.relpath(d, REFERENCE)
        for f in files:
            if not f.endswith('.py'):
                continue
            dst = os.path.join(OUT, rel, f + 'c')          # sourceless layout: module.pyc where module.py would be
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            try:
                # dfile: the path tracebacks show -- the reference's own (for file:line citations)
                py_compile.compile(os.path.join(d, f), cfile=dst, dfile=os.path.join('/root/reference', rel, f),
                                   doraise=True, invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
                n += 1
            except py_compile.PyCompileError as e:         # a module this interpreter cannot parse is not on the path
                skipped += 1
                if verbose:
                    print('  skipped', os.path.join(rel, f), '--', str(e).splitlines()[-1][:100])
    with open(stamp, 'w') as fp:
        fp.write(tag + '\n%d modules byte-compiled, %d skipped\n' % (n, skipped))
    if verbose:
        print('oracle/_ref: %d modules byte-compiled from %s (%d skipped)' % (n, src_root, skipped))
    return OUT


if __name__ == '__main__':
    build()
