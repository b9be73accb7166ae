# round 5: kernel table + bench line of the final tree (after the GAE / epilogue load batching)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r05_gputest_final3.log 2>&1; tail -2 gpurun_out/r05_gputest_final3.log
python bench.py > gpurun_out/r05_bench_final3.json 2> gpurun_out/r05_bench_final3.err; tail -c 200 gpurun_out/r05_bench_final3.json
rm -rf gpurun_out/prof_b
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/prof_b.log 2>&1
for d in prof_b; do find gpurun_out/$d -mindepth 2 -name '*.csv' -exec mv {} gpurun_out/$d/ \; ; done
