cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_ddpg.py -q -m gpu -x 2>&1 | tail -3
python - <<'P'
import sys; sys.path.insert(0, '.')
import bench
r = bench.secondary_ddpg(steps=300, cpu=False)
print('ms_per_iteration %.4f' % r['ms_per_iteration'])
import json
print(json.dumps(r['dominant_kernel'], indent=0)[:3000])
P
} > gpurun_out/r05_stage.log 2>&1
