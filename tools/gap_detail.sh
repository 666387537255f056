#!/bin/bash
# tools/gap_detail.sh <workload> <round>: every kernel between the end of growth launch <round> and the next one (last pass)
export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
mkdir -p gpurun_out/r03; rm -rf gpurun_out/r03/kt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/kt -- python3 bench.py --workload $1 --steps 1 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0 > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, glob, re
f = glob.glob('gpurun_out/r03/kt/*/*kernel_trace.csv')[0]
rows = sorted(({'n': r['Kernel_Name'], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp'])} for r in csv.DictReader(open(f))), key=lambda r: r['s'])
br = [i for i, r in enumerate(rows) if 'build_records' in r['n']]
rows = rows[br[-1]:]
gs = [i for i, r in enumerate(rows) if 'grow_spec' in r['n']]
k = int("$2") - 1
i = gs[k]; j = gs[k + 1] if k + 1 < len(gs) else len(rows)
t0 = rows[i]['e']
for r in rows[i + 1:j + (1 if j < len(rows) else 0)]:
    m = re.search(r'(\w+_kernel|__amd_\w+|DeviceRadixSort\w*|trampoline_kernel)', r['n']); nm = m.group(1) if m else r['n'][:40]
    print('%9.1f us  +%8.1f us  %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, nm))
PY
rm -rf gpurun_out/r03/kt
