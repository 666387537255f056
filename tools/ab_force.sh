# tools/ab_force.sh: bench lines with one step engine forced (BS_GROW_V2=0 first engine, 1 second engine)
mkdir -p gpurun_out/r03
for w in ${WORKLOADS:-urban_10m urban_50m}; do
  for v in ${ENGINES:-1 0}; do
    BS_GROW_V2=$v timeout -k 10 300 python bench.py --workload $w --secondary= --no-cpu-baseline --concurrent 0 --steps 3 --no-audit > gpurun_out/r03/f_${w}_$v.json 2> gpurun_out/r03/f_${w}_$v.err || { tail -20 gpurun_out/r03/f_${w}_$v.err; exit 1; }
    python -c "
import json,sys; d=json.load(open('gpurun_out/r03/f_${w}_$v.json')); print('$w engine=$v', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()}, d['config']['rg_rounds'], d['config']['validation_rejects'])"
  done
done
