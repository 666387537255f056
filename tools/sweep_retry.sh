#!/bin/bash
export BS_CLOUD_CACHE=/tmp
for v in 16384 65536 1000000; do for w in urban_50m urban_10m facade_1m; do
 BS_RETRY_MAX_LIST=$v python bench.py --workload $w --steps 2 --warmup 1 --secondary= --no-cpu-baseline --no-audit 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('retry_max_list', $v, '$w', round(d['value'],2), 'kernel_ms', round(d['stages_ms']['grow_kernel_ms'],1), 'rounds', d['config']['rg_rounds'])"
done; done
