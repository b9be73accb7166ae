cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -5
python -m pytest tests/test_gpu_learner.py tests/test_gpu_sequences.py tests/test_gpu_dist.py -q -m gpu -x -k "rnn or lstm or sequence or cfg5" 2>&1 | tail -6
} > gpurun_out/r05_mrows2.log 2>&1
