#!/bin/bash
# A/B of several builds of the HIP library on ONE box: tools/ab_bench.sh <workload> <steps> lib1.so lib2.so ...
# (each library is benchmarked twice, interleaved, so that drift shows up)
wl=$1; steps=$2; shift 2
for rep in 1 2; do
  for lib in "$@"; do
    BS_LIB_PATH=$PWD/$lib python bench.py --workload $wl --steps $steps --warmup 1 --secondary "" --no-cpu-baseline --no-audit 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', '$wl', 'value', round(d['value'],3), 'grow_kernel_ms', round(d['stages_ms']['grow_kernel_ms'],1), 'rounds', d['config']['rg_rounds'])"
  done
done
