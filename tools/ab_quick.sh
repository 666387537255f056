# tools/ab_quick.sh: parity subset, then facade_1m / urban_10m / urban_50m bench lines of the current build
mkdir -p gpurun_out/r03
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or fuzz or stress or forged" > gpurun_out/r03/q_parity.log 2>&1 || { tail -30 gpurun_out/r03/q_parity.log; exit 1; }
tail -2 gpurun_out/r03/q_parity.log
for w in ${WORKLOADS:-facade_1m urban_10m urban_50m}; do
    timeout -k 10 300 python bench.py --workload $w --secondary= --no-cpu-baseline --concurrent 0 --steps 3 > gpurun_out/r03/q_$w.json 2> gpurun_out/r03/q_$w.err || { tail -20 gpurun_out/r03/q_$w.err; exit 1; }
    python -c "
import json,sys; d=json.load(open('gpurun_out/r03/q_$w.json')); print('$w', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()}, d['config']['rg_rounds'], d['config']['validation_rejects'])"
done
