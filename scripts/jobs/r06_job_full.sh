#!/bin/bash
# round 6: the whole GPU tier, smoke(), the bench line of the tree as it stands
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r06_gputest.log 2>&1; tail -8 gpurun_out/r06_gputest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py --full-out gpurun_out/r06_bench_full.json > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -c 600 gpurun_out/r06_bench.json
