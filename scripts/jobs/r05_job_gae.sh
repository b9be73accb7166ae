cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "epilogue or zfilter or zupdate or z_" 2>&1 | tail -3
python -m pytest tests/test_gpu_learner.py tests/test_gpu_sequences.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"; done
} > gpurun_out/r05_gae.log 2>&1
