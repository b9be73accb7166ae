"""CPU tier: the C-ABI library builds/loads and exports every symbol include/surreal_amd.h
declares (no compute calls -- there is no GPU here), and the product refuses to run without
its HIP extension instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'surreal_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(smx_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from surreal_amd import build, _lib
    lib_path = build.build(verbose=False)          # hipcc cross-compiles without a GPU
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), 'include/surreal_amd.h declares %s but the .so does not export it' % n
    # and the ctypes binding covers exactly the header
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    assert _lib.load().smx_abi_version() == 1


def test_ctrl_struct_layout_matches_binding():
    """smx_ppo_ctrl_t is addressed as 16 4-byte words from Python"""
    src = open(os.path.join(ROOT, 'include', 'surreal_amd.h')).read()
    end = src.index('} smx_ppo_ctrl_t;')
    body = src[src.rindex('typedef struct {', 0, end) + len('typedef struct {'):end]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1]
        for nm in names.split(','):
            nm = nm.strip()
            m = re.match(r'(\w+)\[(\d+)\]', nm)
            fields += [m.group(1)] * int(m.group(2)) if m else [nm]
    from surreal_amd import _lib as L
    assert len(fields) == L.CTRL_WORDS
    assert fields[L.C_LR_ACTOR] == 'lr_actor' and fields[L.C_KL_TARGET] == 'kl_target'
    assert fields[L.C_STEP_ACTOR] == 'adam_step_actor' and fields[L.C_STOP] == 'stop_flag'
    assert fields[L.C_EPOCHS_DONE] == 'epochs_done'


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from surreal_amd import _lib, kernels
    with pytest.raises(_lib.SmxError):
        kernels.HipKernels()
    import helpers as H
    g, case = H.load_golden('tiny_clip')
    _, params, z = H.case_inputs(case)
    with pytest.raises(_lib.SmxError):
        H.make_learner(case, params, z)


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'surreal_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+(oracle|ppo_oracle|ref_shims|cpu_kernels)\b', txt, re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


STRUCTS = {               # C typedef -> ctypes mirror in surreal_amd/_lib.py
    'smx_mlp3_t': 'Mlp3', 'smx_mlp3_job_t': 'Mlp3Job', 'smx_lstm_t': 'Lstm', 'smx_adam_group_t': 'AdamGroup',
    'smx_ppo_losses_t': 'PpoLosses', 'smx_ppo_combine_t': 'PpoCombine', 'smx_synth_act_step_t': 'SynthActStep',
    'smx_epoch_job_t': 'EpochJob', 'smx_epoch_pack_t': 'EpochPack', 'smx_epoch_prep_t': 'EpochPrep', 'smx_learn_epilogue_t': 'LearnEpilogue',
    'smx_xchg_t': 'Xchg', 'smx_synth_rollout_t': 'SynthRollout', 'smx_linear_job_t': 'LinearJob',
    'smx_gather_job_t': 'GatherJob', 'smx_ddpg_net_t': 'DdpgNet', 'smx_ddpg_rows_t': 'DdpgRows', 'smx_ddpg_update_t': 'DdpgUpdate',
}


def test_struct_layouts_match_the_ctypes_mirrors(tmp_path):
    """every struct that crosses the C ABI by pointer: sizeof and the offset of every field as the C
    compiler lays them out (gcc on include/surreal_amd.h) against the ctypes.Structure the product
    passes -- a silent mismatch would hand the kernels garbage"""
    import subprocess
    from surreal_amd import _lib as L
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "surreal_amd.h"', 'int main(void) {']
    for cname, pyname in STRUCTS.items():
        cls = getattr(L, pyname)
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {}
    for ln in out.splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, pyname in STRUCTS.items():
        cls = getattr(L, pyname)
        assert got[(cname, 'sizeof')] == ctypes.sizeof(cls), (cname, got[(cname, 'sizeof')], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
    # every struct typedef of the header has a mirror (anonymous `typedef struct {` and named ones)
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'surreal_amd.h')).read(), flags=re.S)
    typedefs = set(re.findall(r'}\s*(smx_\w+_t)\s*;', hdr)) - {'smx_ppo_ctrl_t'}      # (addressed as words)
    assert typedefs == set(STRUCTS), (sorted(typedefs), sorted(STRUCTS))


def test_ctypes_signatures_match_the_header():
    """every entry point: number and kind of parameters (and the return type) the header declares
    against the argtypes / restype the ctypes binding installs"""
    from surreal_amd import _lib as L
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'surreal_amd.h')).read(), flags=re.S)
    decls = re.findall(r'\n\s*([A-Za-z_][\w\s\*]*?)\b(smx_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', hdr)
    assert len(decls) == len(declared_symbols())
    kinds = {ctypes.c_int32: 'i32', ctypes.c_int64: 'i64', ctypes.c_float: 'f32', ctypes.c_double: 'f64',
             ctypes.c_uint64: 'u64', ctypes.c_void_p: 'ptr', ctypes.c_char_p: 'ptr'}
    kinds[ctypes.c_size_t] = 'u64'                         # (the same 8-byte class as c_uint64 on this ABI)

    def kind_of_c(t):
        t = t.strip()
        if '*' in t or re.search(r'\bsmx_stream_t\b', t):
            return 'ptr'
        for pat, k in ((r'\bint32_t\b|\bint\b', 'i32'), (r'\buint64_t\b', 'u64'), (r'\bint64_t\b', 'i64'),
                       (r'\bsize_t\b', 'u64'), (r'\bfloat\b', 'f32'), (r'\bdouble\b', 'f64')):
            if re.search(pat, t):
                return k
        raise AssertionError('unknown C type %r' % t)

    def kind_of_ct(t):
        if t in kinds:
            return kinds[t]
        assert hasattr(t, '_type_') or t is None, t        # POINTER(struct)
        return 'ptr'
    for ret, name, params in decls:
        restype, argtypes = L._SIGS[name]
        plist = [p for p in (q.strip() for q in params.replace('\n', ' ').split(',')) if p and p != 'void']
        assert len(plist) == len(argtypes), (name, plist, argtypes)
        for p, a in zip(plist, argtypes):
            assert kind_of_c(p) == kind_of_ct(a), (name, p, a)
        assert kind_of_c(ret) == kind_of_ct(restype), (name, ret, restype)


def test_epoch_supported_mirrors_the_launch_lds_budget():
    """smx_epoch_supported() is what makes the learner choose the fused row-block epochs; it must refuse what
    smx_epoch_forward_f32 cannot place in 128 KB of LDS (x, h1 and h2 tiles of 16 rows), or such configs
    raise at launch instead of taking the layered schedule (host-side arithmetic: no GPU needed)"""
    from surreal_amd import _lib as L
    lib = L.load()
    assert lib.smx_epoch_supported(376, 300, 200, 17) == 1          # the benchmark shape
    assert lib.smx_epoch_supported(17, 300, 200, 6) == 1
    assert lib.smx_epoch_supported(1200, 300, 200, 17) == 0         # ~133 KB: was accepted, failed at launch
    assert lib.smx_epoch_supported(1024, 640, 640, 17) == 0
    assert lib.smx_epoch_supported(376, 302, 200, 17) == 0          # hidden sizes must be multiples of 4
    # the largest observation the default hidden sizes leave room for is accepted and one 64-column step more is not
    ok = [d for d in range(64, 2049, 64) if lib.smx_epoch_supported(d, 300, 200, 17)]
    assert ok and ok == list(range(64, ok[-1] + 1, 64)) and ok[-1] < 1200


def test_peer_exchange_buffer_holds_the_padded_chunks():
    """smx_xchg_bytes(): the staging holds the vector cut into `world` chunks of ceil(n / world) rounded up to whole
    256-byte lines -- INCLUDING the padding of the last chunk, which the all-reduce's first step really writes -- plus
    the two halves of the reduced-chunk buffer behind a 16 KB header.  (Sized by the capacity alone the padding ran
    into the first reduced chunk: 28 floats at 8 ranks; host-side arithmetic, no GPU needed.)"""
    from surreal_amd import _lib as L
    lib = L.load()
    line = 64                                          # floats
    for world in range(2, 9):
        for cap in (1, 3, 63, 64, 65, 4096, 540001, 536278, 10 ** 7 + 1):
            chunk = -(-(-(-cap // world)) // line) * line
            assert chunk * world >= cap and chunk % line == 0
            assert lib.smx_xchg_bytes(cap, world) == 16384 + 4 * (world * chunk + 2 * chunk), (cap, world)
    assert lib.smx_xchg_bytes(0, 2) == 0 and lib.smx_xchg_bytes(10, 1) == 0 and lib.smx_xchg_bytes(10, 9) == 0
