# tools/ab_lib.sh <variant.so> [workloads...]: bench lines of the in-tree library and of a variant build on the same box
V=$1; shift
mkdir -p gpurun_out/r03
for w in ${@:-urban_10m urban_50m uniform_10m}; do
  for lib in tree variant; do
    if [ $lib = variant ]; then export BS_LIB_PATH=$V; else unset BS_LIB_PATH; fi
    timeout -k 10 300 python bench.py --workload $w --secondary= --no-cpu-baseline --concurrent 0 --steps 3 --no-audit > gpurun_out/r03/lib_${w}_$lib.json 2> gpurun_out/r03/lib_${w}_$lib.err || { tail -20 gpurun_out/r03/lib_${w}_$lib.err; exit 1; }
    python -c "
import json,sys; d=json.load(open('gpurun_out/r03/lib_${w}_$lib.json')); print('$w $lib', round(d['value'],2), {k:round(x,1) for k,x in d['stages_ms'].items()})"
  done
done
