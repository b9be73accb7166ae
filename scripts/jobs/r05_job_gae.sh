cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gae or moments or norm" 2>&1 | tail -3
python -m pytest tests/test_gpu_learner.py -q -m gpu -x -k "golden" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -c 700
} > gpurun_out/r05_gae.log 2>&1
