mkdir -p gpurun_out/r03
for rep in 1 2; do
for v in default 1; do
  if [ $v = default ]; then unset BS_GROW_V2; else export BS_GROW_V2=$v; fi
  timeout -k 10 300 python bench.py --workload urban_50m --secondary= --no-cpu-baseline --concurrent 0 --steps 6 --no-audit > gpurun_out/r03/r1_$v.json 2> gpurun_out/r03/r1_$v.err || exit 1
  python -c "
import json,sys; d=json.load(open('gpurun_out/r03/r1_$v.json')); print('engine=$v', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()})"
done
done
