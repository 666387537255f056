mkdir -p gpurun_out/r03
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r03/knn_parity.log 2>&1 || { tail -30 gpurun_out/r03/knn_parity.log; exit 1; }
tail -2 gpurun_out/r03/knn_parity.log
for w in ${WORKLOADS:-facade_1m urban_10m uniform_10m urban_50m}; do
  for v in 1 0; do
    BS_KNN_BUFFERED=$v timeout -k 10 300 python bench.py --workload $w --secondary= --no-cpu-baseline --concurrent 0 --steps 3 --no-audit > gpurun_out/r03/knn_${w}_$v.json 2> gpurun_out/r03/knn_${w}_$v.err || { tail -20 gpurun_out/r03/knn_${w}_$v.err; exit 1; }
    python -c "
import json,sys; d=json.load(open('gpurun_out/r03/knn_${w}_$v.json')); print('$w buffered=$v', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()}, d.get('tie_rows'))"
  done
done
