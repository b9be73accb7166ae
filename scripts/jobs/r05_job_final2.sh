# round 5, the tree after the final job: the GPU tier and the bench line again (DDPG staging / level-1 fold, unused imports)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r05_gputest_final2.log 2>&1; tail -3 gpurun_out/r05_gputest_final2.log
(time python bench.py > gpurun_out/r05_bench_final2.json 2> gpurun_out/r05_bench_final2.err); tail -c 300 gpurun_out/r05_bench_final2.json
python bench.py --secondary all --budget-s 420 --full-out gpurun_out/r05_bench_full_all2.json > gpurun_out/r05_bench_all_line2.json 2> gpurun_out/r05_bench_all2.err; tail -c 200 gpurun_out/r05_bench_all_line2.json
