mkdir -p gpurun_out/r03
BS_GROW_V2=3 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or fuzz or stress or forged" > gpurun_out/r03/v3_parity.log 2>&1 || { tail -30 gpurun_out/r03/v3_parity.log; exit 1; }
tail -3 gpurun_out/r03/v3_parity.log
for w in facade_1m urban_10m; do
  for v in 3 1; do
    BS_GROW_V2=$v timeout -k 10 300 python bench.py --workload $w --secondary= --no-cpu-baseline --concurrent 0 --steps 3 > gpurun_out/r03/ab_${w}_v3_$v.json 2> gpurun_out/r03/ab_${w}_v3_$v.err || { tail -20 gpurun_out/r03/ab_${w}_v3_$v.err; exit 1; }
    python -c "
import json,sys; d=json.load(open('gpurun_out/r03/ab_${w}_v3_$v.json')); print('$w V2=$v', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()}, d['config']['rg_rounds'], d['config']['validation_rejects'], d.get('rank0_kernel_ms') or d.get('kernel_ms'))"
  done
done
