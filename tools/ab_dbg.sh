#!/bin/bash
# tools/ab_dbg.sh <workload> lib1.so ...: one debug bench per library: kernel ms and critical-path steps (sum of per-round maxsteps)
wl=$1; shift
export BS_CLOUD_CACHE=/tmp
for lib in "$@"; do
  BS_DEBUG=1 BS_LIB_PATH=$PWD/$lib python bench.py --workload $wl --steps 1 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0 2> /tmp/ab_dbg.err | python -c "
import sys,json,re
d=json.loads(sys.stdin.readline())
err=open('/tmp/ab_dbg.err').read()
ms=[int(m) for m in re.findall(r'maxsteps=(\d+)', err)]
half=len(ms)//2
print('$lib', '$wl', 'value', round(d['value'],3), 'grow_kernel_ms', round(d['stages_ms']['grow_kernel_ms'],1), 'rounds', d['config']['rg_rounds'], 'critical steps (last pass)', sum(ms[half:]), 'us/step', round(1e3*d['stages_ms']['grow_kernel_ms']/max(1,sum(ms[half:])),3))"
done
