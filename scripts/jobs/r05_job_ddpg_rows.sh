#!/bin/bash
# the DDPG row-block schedule: tests, then sample + learn timed on both schedules
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_ddpg.py -q -m gpu -x 2>&1 | tail -25
python - <<'P'
import time, torch, sys
sys.path.insert(0, '.')
import bench
for rows in (True, False):
    import surreal_amd.main.ddpg_configs as C
    orig = C.ddpg_session_config
    def sc(*a, **k):
        s = orig(*a, **k)
        s.learner['ddpg_row_schedule'] = rows
        return s
    C.ddpg_session_config = sc
    r = bench.secondary_ddpg(steps=300, cpu=False)
    C.ddpg_session_config = orig
    print('row_schedule', rows, 'ms_per_iteration %.4f' % r['ms_per_iteration'])
    dk = r.get('dominant_kernel', {})
    for t in dk.get('top5', []):
        print('   ', t)
    print('    device_us_per_learn', dk.get('device_us_per_learn'))
P
} > gpurun_out/r05_ddpg_rows.log 2>&1
