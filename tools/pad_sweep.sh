mkdir -p gpurun_out/r03
for pad in 0 16000 36000 64000; do
  BS_GROW_LDS_PAD=$pad timeout -k 10 300 python bench.py --workload urban_50m --secondary= --no-cpu-baseline --concurrent 0 --steps 2 --no-audit > gpurun_out/r03/pad_$pad.json 2> gpurun_out/r03/pad_$pad.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r03/pad_$pad.json')); print('pad $pad', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()})"
done
