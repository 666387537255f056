#!/bin/bash
# round-2 baseline measurements on the GPU box: 50 M round log, kernel stats, bench lines
export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
mkdir -p gpurun_out/r02
BS_DEBUG=1 python3 bench.py --workload urban_50m --steps 2 --warmup 1 --secondary "" --no-cpu-baseline --no-audit > gpurun_out/r02/b50_dbg.json 2> gpurun_out/r02/b50_dbg.err || exit 1
echo "50m debug bench done"; cat gpurun_out/r02/b50_dbg.json
rm -rf gpurun_out/r02/ks50
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/ks50 -- python3 bench.py --workload urban_50m --steps 2 --warmup 1 --secondary "" --no-cpu-baseline --no-audit > gpurun_out/r02/b50_prof.json 2> gpurun_out/r02/b50_prof.err || exit 1
cp gpurun_out/r02/ks50/*/*kernel_stats.csv gpurun_out/r02/ks50_kernel_stats.csv
rm -rf gpurun_out/r02/ks50
echo "50m kernel stats done"
python3 bench.py --workload facade_1m --steps 3 --warmup 1 --secondary "" --no-cpu-baseline --no-audit > gpurun_out/r02/b1.json 2> gpurun_out/r02/b1.err || exit 1
cat gpurun_out/r02/b1.json
