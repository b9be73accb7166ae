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
