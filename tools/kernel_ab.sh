export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
for lib in "$@"; do
rm -rf gpurun_out/r02/ks_tmp
BS_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/ks_tmp -- python3 bench.py --workload urban_50m --steps 2 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/r02/ks_tmp/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("cand_flag", "pull_pass", "decide_", "rev_fill", "static_mask")):
        print("$lib", r['Name'][28:50], int(r['Calls'])//3, 'calls/pass', round(float(r['TotalDurationNs'])/3e6,2), 'ms/pass')
PY
done
rm -rf gpurun_out/r02/ks_tmp
