import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the host-side CPython extension (csrc/host/smx_host.c -> surreal_amd/_smx_host.so): built here, incrementally, so
    # that a fresh checkout's first test run has it (gcc, two seconds; test_abi_symbols does the same for the HIP library)
    try:
        from surreal_amd import build
        build.build_host(verbose=False)
    except Exception as e:                     # the tests that need it say so themselves
        import warnings
        warnings.warn('could not build surreal_amd/_smx_host.so: %r' % (e,))


@pytest.fixture
def cpu_double():
    """install the torch-CPU kernel test double for the duration of a test"""
    from surreal_amd import kernels as KN
    from cpu_kernels import TorchCpuKernels
    prev = KN.set_default_kernels(TorchCpuKernels(), 'cpu')
    yield
    KN.set_default_kernels(*prev)


def _tier():
    """'gpu' when the HIP path ran in this session (a device is visible), else 'cpu' (the kernel test double): the
    report files carry it, so a CPU-tier run never overwrites the GPU box's evidence"""
    try:
        import torch
        return 'gpu' if torch.cuda.is_available() else 'cpu'
    except Exception:
        return 'cpu'


def pytest_sessionfinish(session, exitstatus):
    """how tight the final-parameter comparison was (helpers.assert_final_params): per golden case and tensor, the
    fraction of elements further than 1e-5 from the reference's and the largest difference -> one summary line, and
    gpurun_out/final_params_report_<tier>.json (tier = gpu | cpu) when that directory exists (the GPU box)"""
    try:
        import helpers
    except Exception:
        return
    arb = getattr(helpers, 'FP64_REPORT', {})
    if arb:
        worst = max(arb.items(), key=lambda kv: kv[1][0] / kv[1][2])
        print('\nfp64 arbiter: %d gradient-norm traces; closest to its bound: %s -- %.3g from float64, the reference %.3g, '
              'bound %.3g' % (len(arb), worst[0], worst[1][0], worst[1][1], worst[1][2]))
        out = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(out):
            import json
            json.dump({k: {'path_vs_fp64': v[0], 'reference_vs_fp64': v[1], 'bound': v[2], 'share_of_bound': v[0] / v[2],
                           'test': v[3] if len(v) > 3 else ''} for k, v in arb.items()},
                      open(os.path.join(out, 'fp64_arbiter_report_%s.json' % _tier()), 'w'), indent=0)
    seeds = getattr(helpers, 'FP64_SEED_REPORT', {})
    if seeds:
        for k, v in seeds.items():
            if 'hip' not in v:
                continue
            print('fp64 arbiter over %d seeds, %s: HIP worst %.3g (%.0f %% of its bound), median %.3g (%.0f %%); the reference '
                  'worst %.3g, median %.3g' % (len(v['hip']), k, v['hip_max'], 100 * v['share_max'], v['hip_median'],
                                               100 * v['share_median'], max(v['reference']), sorted(v['reference'])[len(v['reference']) // 2]))
        out = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(out):
            import json
            json.dump(seeds, open(os.path.join(out, 'fp64_seed_report_%s.json' % _tier()), 'w'), indent=1)
    try:
        import ddpg_helpers
        drep = ddpg_helpers.DDPG_PARAM_REPORT
    except Exception:
        drep = {}
    if drep:
        import json
        tens = {k: v for k, v in drep.items() if isinstance(v, tuple)}
        if tens:
            w = max(tens.items(), key=lambda kv: kv[1][0] / kv[1][3])
            print('\nDDPG parameters vs the reference: %d tensors, %d elements; closest to its bound: %s -- max diff %.2e '
                  '(bound %.0e)' % (len(tens), sum(v[2] for v in tens.values()), w[0], w[1][0], w[1][3]))
        out = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(out):
            json.dump({k: (dict(max_diff=v[0], frac_off_1e5=v[1], elements=v[2], bound=v[3]) if isinstance(v, tuple) else v)
                       for k, v in drep.items()}, open(os.path.join(out, 'ddpg_params_report_%s.json' % _tier()), 'w'), indent=0)
    rep = helpers.FINAL_PARAM_REPORT
    if not rep:
        return
    import json
    worst = max(rep.items(), key=lambda kv: kv[1][0])
    total = sum(v[2] for v in rep.values())
    off = sum(v[0] * v[2] for v in rep.values())
    print('\nfinal parameters vs the reference: %d tensors, %d elements, %.5f %% further than 1e-5 overall; worst tensor '
          '%s: %.4f %% (max diff %.2e)' % (len(rep), total, 100.0 * off / max(total, 1), worst[0], 100 * worst[1][0],
                                           worst[1][1]))
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        json.dump({k: {'frac_off': v[0], 'max_diff': v[1], 'elements': v[2]} for k, v in rep.items()},
                  open(os.path.join(out, 'final_params_report_%s.json' % _tier()), 'w'), indent=0)
