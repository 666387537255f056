mkdir -p gpurun_out/r03
for w in facade_1m urban_10m urban_50m; do
  for v in 1 0; do
    BS_GROW_V2=$v python bench.py --workload $w --secondary= --no-cpu-baseline --concurrent 0 --steps 3 --no-audit > gpurun_out/r03/ab_${w}_v2_$v.json 2> gpurun_out/r03/ab_${w}_v2_$v.err
    python -c "
import json,sys; d=json.load(open('gpurun_out/r03/ab_${w}_v2_$v.json')); print('$w V2=$v', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()}, d['config']['rg_rounds'], d['config']['validation_rejects'])"
  done
done
