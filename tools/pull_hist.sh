#!/bin/bash
# histogram of pull_pass_kernel durations for one pass of a workload (rocprofv3 kernel trace)
export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
rm -rf gpurun_out/r03/kt; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/kt -- python3 bench.py --workload $1 --steps 1 --warmup 0 --secondary= --no-cpu-baseline --no-audit > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/r03/kt/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
t0=min(int(r['Start_Timestamp']) for r in rows)
pp=[(int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows if 'pull_pass' in r['Kernel_Name']]
gs=[(int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows if 'grow_spec' in r['Kernel_Name']]
print('pull passes', len(pp), 'total ms', sum(d for _,d in pp)/1e6)
import bisect
starts=[s for s,_ in gs]
per={}
for s,d in pp:
    k=bisect.bisect_right(starts,s)
    per.setdefault(k,[0,0.0,0]); per[k][0]+=1; per[k][1]+=d/1e6; per[k][2]=max(per[k][2],d)
for k in sorted(per): print('after grow launch', k, 'passes', per[k][0], 'ms', round(per[k][1],2), 'max_us', per[k][2]//1000)
PY
rm -rf gpurun_out/r03/kt
