#!/bin/bash
# tools/ab_plain.sh <workload> lib1.so ...: one plain bench (3 timed passes) per library; prints the end-to-end rate and
# the part of a pass that is NOT the growth kernel (setup, owner passes, validation, host round trips)
wl=$1; shift
export BS_CLOUD_CACHE=/tmp
for lib in "$@"; do
  BS_LIB_PATH=$PWD/$lib python bench.py --workload $wl --steps 3 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); s=d['stages_ms']
print('$lib', '$wl', 'value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'grow_kernel', round(s['grow_kernel_ms'],1), 'setup', round(s['grow_setup_ms'],1), 'rest of stage 3', round(s['grow_ms']-s['grow_kernel_ms']-s['grow_setup_ms'],1), 'rounds', d['config']['rg_rounds'])"
done
