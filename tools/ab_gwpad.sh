# tools/ab_gwpad.sh: static_mask_kernel with different geometry-tile paddings (variant builds), setup and kernel time at 50 M
mkdir -p gpurun_out/r03
for v in 384 192 128; do
  if [ $v = 384 ]; then unset BS_LIB_PATH; else export BS_LIB_PATH=/root/repo/build/libgw$v.so; fi
  export TMPDIR=/tmp; rm -rf gpurun_out/r03/gwtmp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/gwtmp -- python3 bench.py --workload urban_50m --steps 2 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0 > gpurun_out/r03/gw_$v.json 2> gpurun_out/r03/gw_$v.err || exit 1
  python3 - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/r03/gwtmp/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'static_mask_kernel' in r['Name'] or 'rev_fill' in r['Name']:
        print('GW_PAD $v', r['Name'].split('(')[0][-22:], round(float(r['AverageNs'])/1e6,2), 'ms')
d=json.load(open('gpurun_out/r03/gw_$v.json')); print('GW_PAD $v', round(d['value'],1), round(d['stages_ms']['grow_setup_ms'],1))
PY
done
rm -rf gpurun_out/r03/gwtmp
