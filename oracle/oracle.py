"""ctypes binding of oracle/libbs_oracle.so (the CPU restatement) and a runner
for oracle/_ref/ref_stage3 (the reference's own stage-3 code, compiled verbatim
in the build container).  TEST INFRASTRUCTURE ONLY -- see oracle/bs_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Planes(C.Structure):
    _fields_ = [("n_planes", C.c_int32), ("id", C.POINTER(C.c_int32)),
                ("normal", C.POINTER(C.c_double)), ("center", C.POINTER(C.c_int32)),
                ("offset", C.POINTER(C.c_int64)), ("point_idx", C.POINTER(C.c_int32))]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libbs_oracle.so")
    src = os.path.join(_HERE, "bs_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbs_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        ip, dp, lp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
        L.bso_knn_normals.argtypes = [ip, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double,
                                      C.c_int, C.c_int, ip, dp]
        L.bso_knn_brute.argtypes = [ip, C.c_int64, C.c_int64, C.c_int64, C.c_int, ip]
        L.bso_normal_from_list.argtypes = [ip, ip, C.c_int, dp]
        L.bso_fast_eigen3x3.argtypes = [dp, dp]
        L.bso_region_grow.argtypes = [ip, dp, ip, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double,
                                      ip, C.POINTER(_Planes), lp]
        L.bso_region_grow_owner.argtypes = [ip, dp, ip, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double,
                                            ip, C.POINTER(_Planes), lp, ip]
        L.bso_planes_free.argtypes = [C.POINTER(_Planes)]
        L.bso_det_acos.argtypes = [C.c_double]
        L.bso_det_acos.restype = C.c_double
        L.bso_det_cos.argtypes = [C.c_double]
        L.bso_det_cos.restype = C.c_double
        L.bso_det_log.argtypes = [C.c_double]
        L.bso_det_log.restype = C.c_double
        L.bso_grid_dims.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.bso_grid_picture.argtypes = [ip, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int, dp,
                                       C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def knn_normals(xyz, k=15, radius=100.0, max_nn=50, cell=0, q0=0, q1=None, want_normals=True):
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    n = len(xyz)
    q1 = n if q1 is None else q1
    neigh = np.empty((q1 - q0, k), dtype=np.int32)
    normals = np.empty((q1 - q0, 3), dtype=np.float64) if want_normals else None
    rc = lib().bso_knn_normals(_ip(xyz), n, q0, q1, k, radius, max_nn, cell, _ip(neigh),
                               _dp(normals) if want_normals else None)
    if rc != 0:
        raise ValueError(f"bso_knn_normals failed: {rc}")
    return neigh, normals


def knn_brute(xyz, k=15, q0=0, q1=None):
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    n = len(xyz)
    q1 = n if q1 is None else q1
    neigh = np.empty((q1 - q0, k), dtype=np.int32)
    rc = lib().bso_knn_brute(_ip(xyz), n, q0, q1, k, _ip(neigh))
    if rc != 0:
        raise ValueError(f"bso_knn_brute failed: {rc}")
    return neigh


def normal_from_list(xyz, idx):
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.empty(3, dtype=np.float64)
    lib().bso_normal_from_list(_ip(xyz), _ip(idx), len(idx), _dp(out))
    return out


def fast_eigen3x3(c6):
    c6 = np.ascontiguousarray(c6, dtype=np.float64)
    out = np.empty(3, dtype=np.float64)
    lib().bso_fast_eigen3x3(_dp(c6), _dp(out))
    return out


def _planes_to_py(P):
    npl = P.n_planes
    if npl == 0:
        return {"id": np.zeros(0, np.int32), "normal": np.zeros((0, 3)), "center": np.zeros((0, 3), np.int32),
                "offset": np.zeros(1, np.int64), "point_idx": np.zeros(0, np.int32)}
    off = np.ctypeslib.as_array(P.offset, (npl + 1,)).copy()
    return {"id": np.ctypeslib.as_array(P.id, (npl,)).copy(),
            "normal": np.ctypeslib.as_array(P.normal, (npl * 3,)).reshape(npl, 3).copy(),
            "center": np.ctypeslib.as_array(P.center, (npl * 3,)).reshape(npl, 3).copy(),
            "offset": off,
            "point_idx": (np.ctypeslib.as_array(P.point_idx, (int(off[-1]),)).copy()
                          if off[-1] > 0 else np.zeros(0, np.int32))}


def region_grow(xyz, normals, neigh, th_thickness=300, th_point_count=400, cos_th=0.88, want_owner=False):
    """(plane_idx, planes) -- with want_owner also owner [n]: the seed attempt that left each point labelled."""
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    normals = np.ascontiguousarray(normals, dtype=np.float64)
    neigh = np.ascontiguousarray(neigh, dtype=np.int32)
    n, k = neigh.shape
    plane_idx = np.empty(n, dtype=np.int32)
    owner = np.empty(n, dtype=np.int32) if want_owner else None
    P = _Planes()
    att = C.c_int64(0)
    if want_owner:
        rc = lib().bso_region_grow_owner(_ip(xyz), _dp(normals), _ip(neigh), n, k, th_thickness, th_point_count,
                                         cos_th, _ip(plane_idx), C.byref(P), C.byref(att), _ip(owner))
    else:
        rc = lib().bso_region_grow(_ip(xyz), _dp(normals), _ip(neigh), n, k, th_thickness, th_point_count,
                                   cos_th, _ip(plane_idx), C.byref(P), C.byref(att))
    if rc != 0:
        raise ValueError(f"bso_region_grow failed: {rc}")
    planes = _planes_to_py(P)
    planes["n_seed_attempts"] = att.value
    lib().bso_planes_free(C.byref(P))
    if want_owner:
        return plane_idx, planes, owner
    return plane_idx, planes


def det_acos(x):
    return lib().bso_det_acos(float(x))


def det_cos(x):
    return lib().bso_det_cos(float(x))


def det_log(x):
    return lib().bso_det_log(float(x))


def grid_dims(extent, bin=100):
    """(width, height) of the 2-D raster (TMC3.cpp:75-76)."""
    ext = (C.c_int32 * 3)(*[int(v) for v in extent])
    w, h = C.c_int32(0), C.c_int32(0)
    if lib().bso_grid_dims(ext, bin, C.byref(w), C.byref(h)) != 0:
        raise ValueError("bso_grid_dims failed")
    return w.value, h.value


def grid_picture(xyz, extent=None, bin=100, bin_height=1000, libm_log=False):
    """buildingSeg::compute_gird_picture on a cloud that is already shifted to the
    origin.  Returns (image [height][width][3] f64, ground_th)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    if extent is None:
        extent = xyz.max(0)
    w, h = grid_dims(extent, bin)
    ext = (C.c_int32 * 3)(*[int(v) for v in extent])
    img = np.empty((h, w, 3), dtype=np.float64)
    th = C.c_double(0)
    rc = lib().bso_grid_picture(_ip(xyz), len(xyz), ext, bin, bin_height, 1 if libm_log else 0, _dp(img), C.byref(th))
    if rc != 0:
        raise ValueError(f"bso_grid_picture failed: {rc}")
    return img, th.value


# --------------------------------------------------------------------------
# verbatim reference stage 3 (oracle/_ref/ref_stage3): build container only
# --------------------------------------------------------------------------

def ref_stage3_path():
    p = os.path.join(_HERE, "_ref", "ref_stage3")
    return p if os.path.exists(p) else None


def ref_region_grow(xyz, normals, neigh, timeout=3600):
    """Run the reference's own seg_plane::get_planes (fixed thresholds 300 /
    400 / 0.88) on (xyz, normals, neigh).  Returns (plane_idx, planes, colors)."""
    exe = ref_stage3_path()
    if exe is None:
        raise FileNotFoundError("oracle/_ref/ref_stage3 not built (run `make -C oracle ref` where /root/reference exists)")
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    normals = np.ascontiguousarray(normals, dtype=np.float64)
    neigh = np.ascontiguousarray(neigh, dtype=np.int32)
    n, k = neigh.shape
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(np.int64(n).tobytes())
            f.write(np.int32(k).tobytes())
            f.write(xyz.tobytes())
            f.write(normals.tobytes())
            f.write(neigh.tobytes())
        subprocess.check_call([exe, fi, fo], timeout=timeout)
        buf = open(fo, "rb").read()
    pos = 0
    plane_idx = np.frombuffer(buf, np.int32, n, pos).copy()
    pos += 4 * n
    npl = int(np.frombuffer(buf, np.int32, 1, pos)[0])
    pos += 4
    ids, nrm, ctr, off, pidx = [], [], [], [0], []
    for _ in range(npl):
        ids.append(int(np.frombuffer(buf, np.int32, 1, pos)[0])); pos += 4
        nrm.append(np.frombuffer(buf, np.float64, 3, pos).copy()); pos += 24
        ctr.append(np.frombuffer(buf, np.int32, 3, pos).copy()); pos += 12
        sz = int(np.frombuffer(buf, np.int64, 1, pos)[0]); pos += 8
        pidx.append(np.frombuffer(buf, np.int32, sz, pos).copy()); pos += 4 * sz
        off.append(off[-1] + sz)
    colors = np.frombuffer(buf, np.uint16, 3 * n, pos).reshape(n, 3).copy()
    planes = {"id": np.array(ids, np.int32), "normal": np.array(nrm, np.float64).reshape(-1, 3),
              "center": np.array(ctr, np.int32).reshape(-1, 3), "offset": np.array(off, np.int64),
              "point_idx": np.concatenate(pidx).astype(np.int32) if pidx else np.zeros(0, np.int32)}
    return plane_idx, planes, colors


def ref_raster_path():
    p = os.path.join(_HERE, "_ref", "ref_raster")
    return p if os.path.exists(p) else None


def ref_grid_picture(xyz, png_prefix=None, timeout=3600):
    """Run the reference's own buildingSeg (constructor shift, groundTH,
    compute_gird_picture; TMC3.cpp:44-200) on an UNSHIFTED cloud.  Returns
    dict(width, height, min, max, ground_th, image[height][width][3])."""
    exe = ref_raster_path()
    if exe is None:
        raise FileNotFoundError("oracle/_ref/ref_raster not built (run `make -C oracle ref` where /root/reference exists)")
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(np.int64(len(xyz)).tobytes())
            f.write(xyz.tobytes())
        cmd = [exe, fi, fo] + ([png_prefix] if png_prefix else [])
        subprocess.run(cmd, check=True, timeout=timeout)
        raw = open(fo, "rb").read()
    w, h = np.frombuffer(raw, np.int32, 2, 0)
    mn = np.frombuffer(raw, np.int32, 3, 8).copy()
    mx = np.frombuffer(raw, np.int32, 3, 20).copy()
    th = float(np.frombuffer(raw, np.float64, 1, 32)[0])
    img = np.frombuffer(raw, np.float64, int(w) * int(h) * 3, 40).reshape(int(h), int(w), 3).copy()
    return {"width": int(w), "height": int(h), "min": mn, "max": mx, "ground_th": th, "image": img}
