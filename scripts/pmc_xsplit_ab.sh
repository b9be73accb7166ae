cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_fa gpurun_out/pmc_fb
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fa -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_fa.log 2>&1
SMX_EPOCH_NO_XSPLIT=1 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fb -o bench -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > gpurun_out/pmc_fb.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ('pmc_fa', 'pmc_fb'):
    f = glob.glob('gpurun_out/%s/**/bench_counter_collection.csv' % tag, recursive=True)
    if not f:
        print(tag, 'no csv'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        n = r['Kernel_Name']
        if 'epoch_f' in n or 'gemm32' in n:
            agg[(n.split('(')[0][-40:], r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(tag, k, 'launches %d  mean %.1f (x2 corrected: %.2f MB)' % (len(v), sum(v) / len(v), 2 * sum(v) / len(v) * 1e-3 * 1.0))
PY
