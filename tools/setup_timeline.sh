#!/bin/bash
# tools/setup_timeline.sh <workload>: every kernel from the start of the last pass's grid build to its first growth launch
export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
mkdir -p gpurun_out/r03; rm -rf gpurun_out/r03/kt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/kt -- python3 bench.py --workload $1 --steps 1 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0 > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, glob, re
f = glob.glob('gpurun_out/r03/kt/*/*kernel_trace.csv')[0]
rows = sorted(({'n': r['Kernel_Name'], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp'])} for r in csv.DictReader(open(f))), key=lambda r: r['s'])
br = [i for i, r in enumerate(rows) if 'build_records' in r['n']]
gs = [i for i, r in enumerate(rows) if 'grow_spec' in r['n'] and i > br[-1]]
# back from build_records to the first kernel of the pass (a gap > 2 ms separates passes)
i0 = br[-1]
while i0 > 0 and rows[i0]['s'] - rows[i0 - 1]['e'] < 2_000_000:
    i0 -= 1
t0 = rows[i0]['s']
prev = t0
for r in rows[i0:gs[0] + 1]:
    m = re.search(r'(\w+_kernel|__amd_\w+|DeviceRadixSort\w*|trampoline_kernel)', r['n']); nm = m.group(1) if m else r['n'][:40]
    print('%9.2f ms  +%8.1f us  (idle before %6.1f us)  %s' % ((r['s'] - t0) / 1e6, (r['e'] - r['s']) / 1e3, (r['s'] - prev) / 1e3, nm))
    prev = max(prev, r['e'])
PY
rm -rf gpurun_out/r03/kt
