mkdir -p gpurun_out/r03
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_errors.py -x -q > gpurun_out/r03/gen_parity.log 2>&1 || { tail -30 gpurun_out/r03/gen_parity.log; exit 1; }
tail -2 gpurun_out/r03/gen_parity.log
for w in uniform_1m uniform_10m; do
  for v in wave thread; do
    if [ $v = thread ]; then export BS_KNN_GENERAL_THREAD=1; else unset BS_KNN_GENERAL_THREAD; fi
    timeout -k 10 300 python bench.py --workload $w --secondary= --no-cpu-baseline --concurrent 0 --steps 3 --no-audit > gpurun_out/r03/gen_${w}_$v.json 2> gpurun_out/r03/gen_${w}_$v.err || { tail -20 gpurun_out/r03/gen_${w}_$v.err; exit 1; }
    python -c "
import json,sys; d=json.load(open('gpurun_out/r03/gen_${w}_$v.json')); print('$w $v', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()}, d['config'].get('fallback_queries'), d.get('tie_rows') or d['config'].get('tie_rows'))"
  done
done
