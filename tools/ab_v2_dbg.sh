mkdir -p gpurun_out/r03
for v in 1 0; do
  BS_DEBUG=1 BS_GROW_V2=$v python bench.py --workload ${1:-urban_50m} --secondary= --no-cpu-baseline --concurrent 0 --steps 1 --warmup 1 --no-audit > gpurun_out/r03/dbg_v2_$v.json 2> gpurun_out/r03/dbg_v2_$v.err
  echo "== V2=$v"; grep "launch span" gpurun_out/r03/dbg_v2_$v.err | cut -c1-230 | tail -26 | head -8
  grep "^\[bs\] round" gpurun_out/r03/dbg_v2_$v.err | tail -26 | head -8 | cut -c1-260
done
