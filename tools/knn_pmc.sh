export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
E=gpurun_out/r03/knnpmc; mkdir -p $E
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf $E/tmp
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $E/tmp -- python3 bench.py --workload ${WL:-urban_50m} --k 16 --steps 1 --warmup 0 --secondary= --no-cpu-baseline --no-audit --concurrent 0 > /dev/null 2> $E/e$i.err || { tail -3 $E/e$i.err; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$E/tmp/*/*counter_collection.csv")[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "knn_fast" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print("knn_fast*", k, "%.0f" % v)
PY
done
rm -rf $E/tmp
