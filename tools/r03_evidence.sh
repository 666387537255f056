#!/bin/bash
# Round-3 evidence on the GPU box (everything the BENCH line's roofline fields can be recomputed from):
#   per workload: rocprofv3 --kernel-trace --stats, PMC passes FETCH_SIZE and WRITE_SIZE (separate runs),
#   facade_1m: SQ instruction-mix passes, in-kernel phase probes.  Output: gpurun_out/r03/ev/
export TMPDIR=/tmp BS_CLOUD_CACHE=/tmp
E=gpurun_out/r03/ev; mkdir -p $E
B="--steps 2 --warmup 1 --secondary= --no-cpu-baseline --no-audit --concurrent 0"
for wl in ${WORKLOADS:-facade_1m urban_10m urban_50m uniform_10m}; do
  rm -rf $E/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $E/tmp -- python3 bench.py --workload $wl $B > $E/ks_$wl.json 2> $E/ks_$wl.err || exit 1
  cp $E/tmp/*/*kernel_stats.csv $E/${wl}_kernel_stats.csv; echo "kernel stats $wl done"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $E/tmp; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $E/tmp_$c -- python3 bench.py --workload $wl $B > $E/pmc_${c}_$wl.json 2> $E/pmc_${c}_$wl.err || exit 1
  done
  python3 tools/pmc_traffic.py $E/tmp_FETCH_SIZE $E/tmp_WRITE_SIZE 3 $wl $E/pmc_traffic.json > $E/pmc_traffic.new && mv $E/pmc_traffic.new $E/pmc_traffic.json
  python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$E/tmp_%s/*/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:100]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    with open("$E/${wl}_pmc_%s_by_kernel.csv" % c.lower(), "w") as o:
        o.write("kernel,dispatches,sum_%s_KB\n" % c)
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write('"%s",%d,%.1f\n' % (k, n, v))
PY
  rm -rf $E/tmp_FETCH_SIZE $E/tmp_WRITE_SIZE; echo "pmc $wl done"
done
# SQ instruction mix of the growth kernel (facade)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1)); rm -rf $E/tmp
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $E/tmp -- python3 bench.py --workload facade_1m --steps 1 --warmup 0 --secondary= --no-cpu-baseline --no-audit --concurrent 0 > /dev/null 2> $E/sq$i.err || exit 1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$E/tmp/*/*counter_collection.csv")[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "grow_spec" in r["Kernel_Name"]:  # both step engines
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
with open("$E/facade_1m_sq_counters.csv", "a") as o:
    for k, v in sorted(agg.items()):
        o.write("grow_spec_kernel,%s,%.0f\n" % (k, v))
PY
done
rm -rf $E/tmp; echo "sq done"
