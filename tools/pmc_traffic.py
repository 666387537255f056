#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in
SEPARATE runs as MI355X_MICROARCH.md prescribes: the two counters do not fit
one TCC pass) of `bench.py` into profiles/pmc_traffic.json.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <n_segment_calls> <workload> [profiles/pmc_traffic.json] > new.json
(the JSON is keyed by workload; an existing file given as 5th argument is updated)

gfx950 correction: FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports half of
the bytes of wide reads -> HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (counters are in KB).
The stage 3 accesses are 16-B loads of 128-B records (one request per line), the same
request shape the guide calibrated; other widths are uncalibrated, so treat the
absolute value as +-2x and the per-kernel ratios as exact.
"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        agg[m.group(1) if m else "other"] += float(r["Counter_Value"]) * 1024.0
    return agg


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    calls = int(sys.argv[3])
    workload = sys.argv[4]
    merge_into = sys.argv[5] if len(sys.argv) > 5 else None  # existing per-workload JSON to update
    stage3 = ["grow_spec_kernel", "grow_spec2_kernel", "pull_pass_kernel", "decide_pass_kernel", "decide_finish_kernel", "static_mask_kernel", "rev_fill_kernel",
              "refresh_records_kernel", "build_records_kernel", "validate1_kernel", "validate2_kernel", "validate3_kernel",
              "plane_apply_kernel", "cand_flag_kernel", "copy_lists_kernel", "label_kernel", "fill_i32_kernel",
              "reset_tags_kernel"]
    out = {"workload": workload, "segment_calls_in_run": calls,
           "formula": "2*FETCH_SIZE + WRITE_SIZE (KB counters), per bs_segment_dev call", "kernels": {}}
    tot = 0.0
    for k in sorted(set(fetch) | set(write)):
        b = (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) / calls
        out["kernels"][k] = {"fetch_x2_bytes": 2 * fetch.get(k, 0.0) / calls, "write_bytes": write.get(k, 0.0) / calls,
                             "hbm_bytes": b}
        if k in stage3:
            tot += b
    out["region_grow_stage_bytes_per_call"] = tot
    # both step engines (grow_spec_kernel, grow_spec2_kernel) are "the growth kernel" of bench.py's roofline block
    out["grow_spec_kernel_bytes_per_call"] = sum(out["kernels"].get(kn, {}).get("hbm_bytes", 0.0)
                                                 for kn in ("grow_spec_kernel", "grow_spec2_kernel"))
    out["knn_fast_kernel_bytes_per_call"] = out["kernels"].get("knn_fast_kernel", {}).get("hbm_bytes")
    db = {}
    if merge_into:
        try:
            db = json.load(open(merge_into))
            if "workload" in db:  # round-1 single-workload layout
                db = {db["workload"]: db}
        except (OSError, ValueError):
            db = {}
    db[workload] = out
    print(json.dumps(db, indent=1))


if __name__ == "__main__":
    main()
