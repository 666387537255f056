#!/usr/bin/env python3
"""PLY golden fixtures from the reference's OWN reader / writer (build container only).

oracle/_ref/ref_ply is /root/reference/tmc3/ply.cpp compiled where it lies plus the driver
oracle/ref/ref_ply_driver.cpp.  For a handful of small input files (binary / ascii, float32 /
float64 coordinates, negative and sub-millimetre values, colours, extra scalar properties, a face
element) this script stores in tests/golden/ply_cases.npz: the input file bytes, the quantised
positions and colour slots the reference reads (ply.cpp:407-415,436-477) and the bytes of the file
it writes back (binary and ascii; ply.cpp:88-186).  tests/test_host_ply_golden.py requires
host/bs_ply.cpp to reproduce every output byte.  Fixtures are data, not source.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_ply")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ply_cases.npz")


def make_inputs():
    rs = np.random.default_rng(11)
    cases = {}

    def ply(fmt, props, rows, faces=True, cr=False):
        nl = "\r\n" if cr else "\n"
        hdr = ["ply", f"format {fmt} 1.0", f"element vertex {len(rows)}"] + [f"property {t} {n}" for t, n, _ in props]
        if faces:
            hdr += ["element face 0", "property list uint8 int32 vertex_index"]
        hdr += ["end_header"]
        head = (nl.join(hdr) + nl).encode()
        if fmt == "ascii":
            body = "".join(" ".join(("%d" % v) if np.issubdtype(np.dtype(d), np.integer) else ("%.17g" % v)
                                    for v, (_, _, d) in zip(r, props)) + "\n" for r in rows).encode()
        else:
            rec = np.zeros(len(rows), dtype=[(n, d) for _, n, d in props])
            for j, (_, n, _) in enumerate(props):
                rec[n] = [r[j] for r in rows]
            body = rec.tobytes()
        return head + body

    vals = [1.2345, -1.2345, -0.0005, 2.9999, 100.0004, 0.001, 0.0, -7.0, 2147.4836, 12345.678901]
    xyz = np.array([[v, vals[-1 - i], (i * 0.37) % 5 - 2.5] for i, v in enumerate(vals)])
    rgb = (np.arange(len(vals) * 3).reshape(-1, 3) * 7 + 3) % 256
    p64 = [("float64", "x", "<f8"), ("float64", "y", "<f8"), ("float64", "z", "<f8")]
    p32 = [("float", "x", "<f4"), ("float32", "y", "<f4"), ("float", "z", "<f4")]
    col = [("uchar", "red", "u1"), ("uchar", "green", "u1"), ("uchar", "blue", "u1")]
    cases["bin_f64_rgb"] = ply("binary_little_endian", p64 + col, [list(a) + list(c) for a, c in zip(xyz, rgb)])
    cases["bin_f32_rgb_extra"] = ply("binary_little_endian",
                                     p32[:1] + [("float32", "intensity", "<f4")] + p32[1:] + [("uint16", "ring", "<u2")] + col,
                                     [[a[0], 0.5, a[1], a[2], 9] + list(c) for a, c in zip(xyz, rgb)])
    cases["ascii_f64_rgb"] = ply("ascii", p64 + col, [list(a) + list(c) for a, c in zip(xyz, rgb)])
    cases["ascii_crlf_header"] = ply("ascii", p64 + col, [list(a) + list(c) for a, c in zip(xyz, rgb)], cr=True)
    cases["bin_f64_nocolor"] = ply("binary_little_endian", p64, [list(a) for a in xyz], faces=False)
    k = rs.integers(0, 1 << 22, (300, 3))
    cases["bin_half_mm_300"] = ply("binary_little_endian", p64 + col,
                                   [list((kk + 0.5) / 1000.0) + [0, 0, 0] for kk in k])
    big = rs.uniform(-500.0, 500.0, (200, 3))
    cases["bin_f32_random_200"] = ply("binary_little_endian", p32 + col[::-1],  # blue, green, red property order
                                      [list(a) + [1, 2, 3] for a in big])
    return cases


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_ply missing: run `make -C oracle ref` in the build container")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, data in make_inputs().items():
            fi = os.path.join(td, name + ".ply")
            open(fi, "wb").write(data)
            out[name + "/in"] = np.frombuffer(data, np.uint8)
            for mode, flag in (("bin", "0"), ("ascii", "1")):
                fo, fd = os.path.join(td, "o.ply"), os.path.join(td, "d.bin")
                subprocess.check_call([REF, fi, "1000", fo, flag, fd])
                out[f"{name}/out_{mode}"] = np.frombuffer(open(fo, "rb").read(), np.uint8)
            raw = open(fd, "rb").read()
            n = int(np.frombuffer(raw, np.int64, 1)[0])
            hc = int(np.frombuffer(raw, np.int32, 1, 8)[0])
            out[name + "/xyz"] = np.frombuffer(raw, np.int32, 3 * n, 12).reshape(n, 3).copy()
            out[name + "/colors"] = (np.frombuffer(raw, np.uint16, 3 * n, 12 + 12 * n).reshape(n, 3).copy()
                                     if hc else np.zeros((0, 3), np.uint16))
            print(name, n, "points, colours" if hc else "points, no colours", len(out[name + "/out_bin"]), "B written")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
