#!/bin/bash
# tools/round_timeline.sh <workload>: per-round wall-clock anatomy of stage 3 from a rocprofv3 kernel trace of ONE pass:
# growth kernel ms, the gap until the next growth launch (validation, owner passes, host round trips), owner passes in it
export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
mkdir -p gpurun_out/r03; rm -rf gpurun_out/r03/kt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/kt -- python3 bench.py --workload $1 --steps 1 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0 > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/r03/kt/*/*kernel_trace.csv')[0]
rows = sorted(({'n': r['Kernel_Name'], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp'])} for r in csv.DictReader(open(f))), key=lambda r: r['s'])
# keep the LAST pass: starts at the last cellkey/grid kernel burst before the last build_records
br = [i for i, r in enumerate(rows) if 'build_records' in r['n']]
rows = rows[br[-1]:]
t0 = rows[0]['s']
gs = [i for i, r in enumerate(rows) if 'grow_spec' in r['n']]
print('setup (build_records .. first growth launch) ms', round((rows[gs[0]]['s'] - t0) / 1e6, 2))
tot_g = tot_gap = tot_busy = 0
for k, i in enumerate(gs):
    j = gs[k + 1] if k + 1 < len(gs) else len(rows)
    seg = rows[i + 1:j]
    g = (rows[i]['e'] - rows[i]['s']) / 1e6
    end = rows[j]['s'] if j < len(rows) else rows[-1]['e']
    gap = (end - rows[i]['e']) / 1e6
    busy = sum(r['e'] - r['s'] for r in seg if 'validate3' not in r['n']) / 1e6
    npull = sum('pull_pass' in r['n'] for r in seg)
    pull = sum(r['e'] - r['s'] for r in seg if 'pull_pass' in r['n']) / 1e6
    tot_g += g; tot_gap += gap; tot_busy += busy
    print('round %2d grow %7.2f ms | gap %6.2f ms: kernels %5.2f (pull %2d passes %5.2f ms, max %4d us) other launches %3d' % (
        k + 1, g, gap, busy, npull, pull, max([0] + [(r['e'] - r['s']) // 1000 for r in seg if 'pull_pass' in r['n']]), len(seg) - npull))
print('sum grow %.1f ms, sum gaps %.1f ms (kernel-busy %.1f ms)' % (tot_g, tot_gap, tot_busy))
PY
rm -rf gpurun_out/r03/kt
