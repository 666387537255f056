#!/bin/bash
# tools/kstats.sh <workload> <tag>: rocprofv3 kernel stats of one bench run (3 passes: 1 warmup + 2 timed) -> gpurun_out/r03/ks_<tag>.csv
wl=$1; tag=$2
export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
mkdir -p gpurun_out/r03; rm -rf gpurun_out/r03/ks_tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/ks_tmp -- python3 bench.py --workload $wl --steps 2 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0 > gpurun_out/r03/ks_${tag}.json 2> gpurun_out/r03/ks_${tag}.err || exit 1
cp gpurun_out/r03/ks_tmp/*/*kernel_stats.csv gpurun_out/r03/ks_${tag}.csv
rm -rf gpurun_out/r03/ks_tmp
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r03/ks_${tag}.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:16]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls'])//3:>5d}/pass {float(r['TotalDurationNs'])/1e6/3:8.2f} ms/pass avg {float(r['AverageNs'])/1e3:9.1f} us")
print("kernel total per pass ms", tot/1e6/3)
PY
