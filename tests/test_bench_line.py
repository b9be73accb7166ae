"""
bench.py's stdout contract: the driver keeps an 8 KB tail of stdout and parses the last line, so the
line must stay under 4 KB whatever the run measured; the complete record goes to --full-out.  The stub is a real full
record (profiles/r04_bench.json: every configuration's entry, ~25 KB) with extra rows piled on.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stub():
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench.json')))
    full.pop('secondary_summary', None)
    # more secondaries than any run has, with long keys, errors and skips among them
    for i in range(12):
        k = 'configs[4] LSTM, adapt: extra row %d %s' % (i, 'x' * 60)
        full['secondary'][k] = dict(full['secondary'][next(iter(full['secondary']))])
    full['secondary']['a failing entry'] = {'error': 'RuntimeError(%r)' % ('boom ' * 200)}
    full['secondary']['a skipped entry'] = {'skipped': 'time budget (--budget-s) spent before this entry'}
    full['cpu_baseline']['reference'] = {'value': 3.1e5, 'cores': 32, 's_per_learn': 0.42, 'learns': 3}
    full['cpu_baseline']['port'] = {'value': 3.3e5, 'cores': 32, 's_per_learn': 0.40}
    full['bench_wall_s'] = 61.2
    return full


def test_line_is_short_and_complete():
    import bench
    full = _stub()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, 'gpurun_out/bench_full.json')
    assert len(line) < bench.LINE_LIMIT == 4096
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'full_record'):
        assert k in d, k
    assert d['config']['workload'] and 'model' not in d['config']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in d['roofline'], k
    assert abs(d['roofline']['frac'] - d['roofline']['achieved'] / d['roofline']['peak']) < 1e-3
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in d['cpu_baseline'], k
    assert d['cpu_baseline']['reference']['value'] == pytest.approx(3.1e5)
    assert abs(d['value'] - full['value']) / full['value'] < 1e-4          # rounded to 5 significant digits, no more
    for row in d.get('secondary_summary', {}).values():
        assert len(row) <= 6, row


def test_eight_gpu_line_carries_the_exchange_fields_and_both_scalings():
    """what `bench.py --gpus 8` hands the driver: the process-group size, which exchange ran and why a
    fallback happened, the weak AND the strong row -- all inside the 4 KB line, with every secondary row it can keep"""
    import bench
    full = _stub()
    full.update(n_gpus=8, scaling='weak')
    full['config'].update(parallelism='dp8', rccl_ranks=8, exchange='process group (RCCL)',
                          exchange_fallback_reason='peer exchange self-check failed: ' + 'hipIpcOpenMemHandle ' * 20,
                          collectives_per_step=25, epoch_all_reduce_us=41.0, epoch_all_reduce_bytes=1402000, graph_segments=True)
    full['strong'] = {'value': 5.1e8, 'unit': 'env-steps/s', 'ms_per_step': 0.26, 'global_batch': 1024, 'B_per_gpu': 128,
                      'hip_graph': True}
    line = bench.compact_line(full, 'gpurun_out/bench_full.json')
    d = json.loads(line)
    assert len(line) < bench.LINE_LIMIT
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp8'
    assert d['config']['rccl_ranks'] == 8 and d['config']['exchange'].startswith('process group')
    assert d['config']['exchange_fallback_reason'].startswith('peer exchange self-check failed')
    assert d['strong']['global_batch'] == 1024 and d['strong']['value'] > 0 and d['value'] > 0
    assert 'traffic_measured_in_run' in d['roofline'] or 'traffic' in d['roofline']


def test_diagnostic_line_is_short():
    import bench
    line = bench.compact_line({'metric': bench.METRIC, 'value': None, 'unit': 'env-steps/s', 'n_gpus': 8, 'steps': 20, 'warmup': 5,
                               'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                               'dtype': 'f32', 'data': 'synthetic', 'config': {'workload': 'w', 'parallelism': 'dp8'},
                               'error': 'e' * 1500})
    d = json.loads(line)
    assert len(line) < 4096 and d['value'] is None and 'ms_per_step' in d and d['error']


def test_last_4k_of_stdout_parses(tmp_path):
    """what the driver does: keep a tail of stdout, parse its last line"""
    full = tmp_path / 'full.json'
    code = ('import sys, json; sys.path.insert(0, %r); sys.argv=["bench.py"]\n'
            'import bench, tests.test_bench_line as T\n'
            'print("noise " * 3000)\n'
            'rec = T._stub()\n'
            'print(bench.compact_line(rec, bench.write_full(rec, %r)), flush=True)\n' % (ROOT, str(full)))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    tail = r.stdout[-4096:]
    d = json.loads(tail.splitlines()[-1])
    assert d['value'] > 0 and d['roofline']['frac'] > 0 and d['cpu_baseline']['value'] > 0
    rec = json.load(open(full))                    # the side file holds everything the line dropped
    assert 'secondary' in rec and len(json.dumps(rec)) > 20000


def test_reference_and_port_cpu_legs_agree(monkeypatch):
    """bench.py's two CPU baselines on a small shape: the REFERENCE'S OWN learner (/root/reference under the
    shims; `kind: reference`) and the restatement (`kind: port`) are the same arithmetic -- cpu_ppo times both, and the
    statistics of one learn agree bit for bit (build container only: skipped where the reference tree does not exist)"""
    import copy
    import numpy as np
    import bench
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ref_shims
    if not ref_shims.reference_available():
        pytest.skip('no reference tree on this host')
    monkeypatch.setenv('SMX_BENCH_REFERENCE', '1')
    import gen_golden as G
    import ppo_oracle
    from surreal_amd import synthetic
    Bs, Ns, Ds, As = 6, 12, 9, 3
    for rnn in (False, True):
        F = 12 if rnn else 0
        batch = synthetic.make_ppo_batch(Bs, Ns, Ds, As, seed=3, rnn_hidden=F)
        params = synthetic.make_ppo_params(Ds, As, hidden=(24, 16), seed=1, rnn_hidden=F)
        hyper = dict(n_step=Ns, kl_target=1e9, ppo_mode='adapt')
        if rnn:
            hyper.update(if_rnn_policy=True, horizon=4)
        ref = ref_shims.import_reference()
        Lr = G.build_reference_learner(ref, params, None, Bs, Ns, Ds, As, hyper)
        bd = ref_shims.BeneDict(copy.deepcopy(batch))
        bd = Lr._preprocess_batch_ppo(bd)
        rs = Lr._optimize(bd.obs, bd.actions, bd.rewards, bd.obs_next, bd.persistent_infos, bd.onetime_infos, bd.dones)
        O = ppo_oracle.OraclePPOLearner(params, As, Bs, **hyper)
        os_ = O.learn(copy.deepcopy(batch))
        for k, v in os_.items():
            if k != '_lr':
                assert float(rs[k]) == float(v) or (np.isnan(float(rs[k])) and np.isnan(float(v))), (rnn, k, rs[k], v)
        # the timing leg itself: both kinds reported, the quoted value is the faster one
        monkeypatch.setattr(bench, '_CPU_THREADS', [2])
        out = bench.cpu_ppo(Bs, Ns, Ds, As, rnn, None, params, batch, budget_s=4.0)
        assert out['port']['value'] > 0 and out['reference']['value'] > 0
        assert out['value'] == max(out['port']['value'], out['reference']['value'])
        assert out['kind'] == ('reference' if out['reference']['value'] >= out['port']['value'] else 'port')
