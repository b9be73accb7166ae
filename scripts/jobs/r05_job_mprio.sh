cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo "== product"; python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
for n in 8 15; do
echo "== waves 4-7 sleep 64 x $n cycles in front of their MFMA loop"; SMX_LIB_PATH=$PWD/surreal_amd/libsurreal_amd_mstag$n.so python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
done
echo "== product"; python scripts/bench_rnn_one.py 1024 128 17 6 2>&1 | tail -1
} > gpurun_out/r05_mprio.log 2>&1
