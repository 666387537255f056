#!/usr/bin/env python3
"""Checks the generated ISA of grow_spec2_kernel: between the partial wait of the gather (s_waitcnt vmcnt(CH + 2)) and the
wait for the flags and the candidates' rows (s_waitcnt vmcnt(0)) no instruction may read or write their destination registers --
the loads are still writing them (csrc/bs_grow_spec.hip, rows_wait).  Usage: check_rows_wait.py [file.s]; without an
argument the device assembly is produced with hipcc -S (about a minute)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "buildingsegment_amd", "csrc", "bs_grow_spec.hip")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "--cuda-device-only", "-S"]


def device_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="bs_isa_"), "bs_grow_spec.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return out


def regs_of(line):
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        regs.add(int(m.group(1)))
    return regs


def check(path):
    """Returns (number of waits checked, list of offending lines)."""
    lines = open(path).read().split("\n")
    bad, checked = [], 0
    for kc, ch in ((16, 4), (32, 8)):
        start = [i for i, l in enumerate(lines) if re.match(r"^_ZN2bs12_GLOBAL__N_117grow_spec2_kernelILi%dE.*:" % kc, l)]
        if not start:
            raise RuntimeError("grow_spec2_kernel<%d> not found in %s" % (kc, path))
        st = start[0]
        en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
        labels = {m.group(1): i for i in range(st, en) for m in [re.match(r"^(\.LBB\d+_\d+):", lines[i])] if m}
        for i in range(st, en):
            if "s_waitcnt vmcnt(%d)" % (ch + 2) not in lines[i]:
                continue
            rows, j = [], i
            while len(rows) < ch + 2 and j > st:  # the loads issued last before the wait: two flags, ch row chunks
                j -= 1
                m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", lines[j])
                if m:
                    rows.append((int(m.group(1)), int(m.group(2))))
                m = re.search(r"global_load_dword v(\d+),", lines[j])
                if m:
                    rows.append((int(m.group(1)), int(m.group(1))))
            rowregs = {r for a, b in rows for r in range(a, b + 1)}
            checked += 1
            # walk every path from the wait until an s_waitcnt vmcnt(0)
            seen, work = set(), [i + 1]
            while work:
                k = work.pop()
                while k < en and k not in seen:
                    seen.add(k)
                    l = lines[k].split(";")[0].strip()
                    if not l or l.endswith(":"):
                        k += 1
                        continue
                    if "s_waitcnt" in l and "vmcnt(0)" in l:
                        break
                    if regs_of(l) & rowregs:
                        bad.append("%s:%d: %s" % (os.path.basename(path), k + 1, l))
                    m = re.match(r"s_(c?branch\w*)\s+(\.LBB\d+_\d+)", l)
                    if m:
                        work.append(labels[m.group(2)])
                        if m.group(1) == "branch":
                            break
                    if l.startswith("s_endpgm"):
                        break
                    k += 1
    return checked, bad


if __name__ == "__main__":
    n, bad = check(sys.argv[1] if len(sys.argv) > 1 else device_asm())
    print("%d partial waits checked, %d offending instructions" % (n, len(bad)))
    for b in bad:
        print("  " + b)
    sys.exit(1 if bad or n < 4 else 0)
