#!/bin/bash
# where an iteration's 108 us go: kernels and the gaps between them (rocprofv3 kernel trace of scripts/bench_ddpg.py's first loop)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ddpg_gaps
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ddpg_gaps -o t -- python $R/scripts/bench_ddpg.py > $R/gpurun_out/ddpg_gaps.log 2>&1
cd $R
f=$(find gpurun_out/ddpg_gaps -name "*_kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
# iterations: from one critic chain launch to the next
idx = [i for i, e in enumerate(ev) if 'ddpg_rows4_kernel<0>' in e[2]]
idx = idx[40:240]
per = collections.defaultdict(list)
for a, b in zip(idx[:-1], idx[1:]):
    seg = ev[a:b]
    period = ev[b][0] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg)
    per['period'].append(period); per['busy'].append(busy); per['n'].append(len(seg))
    for k in range(len(seg)):
        nxt = seg[k + 1][0] if k + 1 < len(seg) else ev[b][0]
        nm = seg[k][2].split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[:40]
        per['gap after %d %s' % (k, nm)].append(nxt - seg[k][1])
        per['dur %d %s' % (k, nm)].append(seg[k][1] - seg[k][0])
import statistics
for k, v in per.items():
    print('%-60s median %8.0f ns' % (k, statistics.median(v)))
PY
find gpurun_out/ddpg_gaps -type f -delete
