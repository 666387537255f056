#!/bin/bash
# per-kernel time of a few setup kernels via rocprofv3 for a lib
lib=$1; wl=$2
export TMPDIR=/tmp; rm -rf gpurun_out/pk; BS_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk -- python3 bench.py --workload $wl --steps 1 --warmup 1 --secondary "" --no-cpu-baseline --no-audit > /dev/null 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob('gpurun_out/pk/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    m=re.search(r'(static_mask_kernel|rev_count_kernel|rev_fill_kernel|build_records_kernel)', r['Name'])
    if m: print("$lib $wl", m.group(1), round(float(r['AverageNs'])/1e6,2), "ms")
PY
