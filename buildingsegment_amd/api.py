"""Python host mirror of the reference's call sites for the hot path.

Reference (paths relative to /root/reference/tmc3/):
    get_Normal_and_K_neighbor<15>(pointCloud, normal, neigh);   TMC3.cpp:215, my_function.h:48-85
    seg_plane h(pointCloud, normal, neigh, 15);                  TMC3.cpp:216, my_function.h:98-104
    vector<plane> planes = h.get_planes();                       TMC3.cpp:217, my_function.cpp:180-217
    h.set_plane_color(planes);                                   TMC3.cpp:218, my_function.cpp:260-275

Everything here goes through the C ABI (include/bs_api.h) into the HIP
library; nothing is computed in Python and there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import BsError, Params, Planes, Timings


def default_params(**kw) -> Params:
    p = Params()
    _lib.load().bs_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown parameter {k}")
        setattr(p, k, v)
    return p


@dataclass
class Plane:
    """struct plane (my_function.h:25-30)."""
    id: int
    normal: np.ndarray
    center: np.ndarray
    pointIdx: np.ndarray = field(repr=False)


def _planes_to_list(P: Planes):
    out = []
    n = P.n_planes
    if n == 0:
        return out
    off = np.ctypeslib.as_array(P.offset, (n + 1,))
    ids = np.ctypeslib.as_array(P.id, (n,))
    nrm = np.ctypeslib.as_array(P.normal, (3 * n,)).reshape(n, 3)
    ctr = np.ctypeslib.as_array(P.center, (3 * n,)).reshape(n, 3)
    pidx = np.ctypeslib.as_array(P.point_idx, (int(off[n]),)) if off[n] > 0 else np.zeros(0, np.int32)
    for i in range(n):
        out.append(Plane(int(ids[i]), nrm[i].copy(), ctr[i].copy(), pidx[off[i]:off[i + 1]].copy()))
    return out


class Context:
    """bs_ctx wrapper: one HIP device, its stream and scratch buffers."""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        rc = self._L.bs_create(device, C.byref(self._h))
        if rc != 0:
            raise BsError(rc, "bs_create failed (is a gfx950 GPU visible?)")
        self.device = device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.bs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise BsError(rc, self._L.bs_last_error(self._h).decode())

    def set_stream(self, stream_handle):
        self._check(self._L.bs_set_stream(self._h, C.c_void_p(stream_handle)))

    def timings(self) -> dict:
        t = Timings()
        self._check(self._L.bs_get_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timings._fields_}

    # ---- host-buffer API ----------------------------------------------------
    def knn_normals(self, xyz, params: Params | None = None):
        p = params or default_params()
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise ValueError("xyz must be [n, 3]")
        n = xyz.shape[0]
        neigh = np.empty((n, p.k), dtype=np.int32)
        normals = np.empty((n, 3), dtype=np.float64)
        self._check(self._L.bs_knn_normals(self._h, xyz.ctypes.data, n, C.byref(p), neigh.ctypes.data,
                                           normals.ctypes.data))
        return neigh, normals

    def knn_normals_halo(self, xyz_local, gidx, n_query, params: Params, cert_radius: float):
        """Slab form: the first n_query points are queries, the rest halo; returns
        (neigh with GLOBAL indices, normals, number of uncertified k-lists)."""
        xyz_local = np.ascontiguousarray(xyz_local, dtype=np.int32)
        gidx = np.ascontiguousarray(gidx, dtype=np.int32)
        n = xyz_local.shape[0]
        neigh = np.empty((n_query, params.k), dtype=np.int32)
        normals = np.empty((n_query, 3), dtype=np.float64)
        unc = C.c_int64(0)
        self._check(self._L.bs_knn_normals_halo(self._h, xyz_local.ctypes.data, gidx.ctypes.data, n, n_query,
                                                C.byref(params), neigh.ctypes.data, normals.ctypes.data,
                                                float(cert_radius), C.byref(unc)))
        return neigh, normals, unc.value

    def region_grow(self, xyz, normals, neigh, params: Params | None = None):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        normals = np.ascontiguousarray(normals, dtype=np.float64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        p = params or default_params(k=neigh.shape[1])
        if neigh.shape[1] != p.k:
            raise ValueError("neigh width != params.k")
        n = xyz.shape[0]
        plane_idx = np.empty(n, dtype=np.int32)
        P = Planes()
        self._check(self._L.bs_region_grow(self._h, xyz.ctypes.data, normals.ctypes.data, neigh.ctypes.data,
                                           n, C.byref(p), plane_idx.ctypes.data, C.byref(P)))
        planes = _planes_to_list(P)
        self._L.bs_planes_free(C.byref(P))
        return plane_idx, planes

    def segment(self, xyz, params: Params | None = None):
        p = params or default_params()
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        n = xyz.shape[0]
        neigh = np.empty((n, p.k), dtype=np.int32)
        normals = np.empty((n, 3), dtype=np.float64)
        plane_idx = np.empty(n, dtype=np.int32)
        P = Planes()
        self._check(self._L.bs_segment(self._h, xyz.ctypes.data, n, C.byref(p), neigh.ctypes.data,
                                       normals.ctypes.data, plane_idx.ctypes.data, C.byref(P)))
        planes = _planes_to_list(P)
        self._L.bs_planes_free(C.byref(P))
        return neigh, normals, plane_idx, planes

    # ---- device-buffer API (raw device pointers, e.g. torch.Tensor.data_ptr()) ----
    def knn_normals_dev(self, d_xyz, n, d_neigh, d_normals, params, q_begin=0, q_end=None, d_gidx=0,
                        cert_radius=0.0):
        q_end = n if q_end is None else q_end
        unc = C.c_int64(0)
        self._check(self._L.bs_knn_normals_dev(self._h, d_xyz, d_gidx or None, n, q_begin, q_end,
                                               C.byref(params), d_neigh, d_normals or None, cert_radius,
                                               C.byref(unc)))
        return unc.value

    def region_grow_dev(self, d_xyz, d_normals, d_neigh, n, d_plane_idx, params):
        self._check(self._L.bs_region_grow_dev(self._h, d_xyz, d_normals, d_neigh, n, C.byref(params),
                                               d_plane_idx))

    def segment_dev(self, d_xyz, n, d_plane_idx, params, d_neigh=0, d_normals=0):
        self._check(self._L.bs_segment_dev(self._h, d_xyz, n, C.byref(params), d_neigh or None,
                                           d_normals or None, d_plane_idx))

    def shift_to_origin_dev(self, d_xyz, n):
        """buildingSeg ctor shift (TMC3.cpp:55-73) in place on the device; returns the minimum."""
        mn = np.zeros(3, dtype=np.int32)
        self._check(self._L.bs_shift_to_origin_dev(self._h, d_xyz, n, mn.ctypes.data))
        return mn

    def ingest_dev(self, d_records, n, stride, offsets, is_f64, d_xyz, scale=1000.0, shift_to_origin=True):
        """ply::read's position quantisation (ply.cpp:436-465) on a device-resident binary vertex body,
        optionally followed by the buildingSeg shift; returns the subtracted minimum."""
        mn = np.zeros(3, dtype=np.int32)
        self._check(self._L.bs_ingest_dev(self._h, d_records, n, stride, offsets[0], offsets[1], offsets[2],
                                          1 if is_f64 else 0, float(scale), 1 if shift_to_origin else 0, d_xyz,
                                          mn.ctypes.data))
        return mn

    def plane_colors_dev(self, plane_rgb, n, d_colors):
        rgb = np.ascontiguousarray(plane_rgb, dtype=np.int32).reshape(-1, 3)
        self._check(self._L.bs_plane_colors_dev(self._h, rgb.ctypes.data, len(rgb), n, d_colors))

    def grid_picture(self, xyz, extent=None, bin=100, bin_height=1000):
        """buildingSeg::compute_gird_picture (TMC3.cpp:123-174) of a cloud already shifted to
        its bounding-box origin.  Returns (image [height][width][3] f64, ground_th)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        ext = np.ascontiguousarray(xyz.max(0) if extent is None else extent, dtype=np.int32)
        w, h = grid_dims(ext, bin)
        img = np.empty((h, w, 3), dtype=np.float64)
        th = C.c_double(0)
        self._check(self._L.bs_grid_picture(self._h, xyz.ctypes.data, len(xyz), ext.ctypes.data, bin, bin_height,
                                            img.ctypes.data, C.byref(th)))
        return img, th.value

    def grid_picture_dev(self, d_xyz, n, extent, d_image, bin=100, bin_height=1000):
        """Device-resident variant: d_xyz / d_image are device pointers (ints)."""
        ext = np.ascontiguousarray(extent, dtype=np.int32)
        th = C.c_double(0)
        self._check(self._L.bs_grid_picture_dev(self._h, d_xyz, n, ext.ctypes.data, bin, bin_height, d_image, C.byref(th)))
        return th.value

    def selftest_center_div(self, c, n):
        """Device evaluation of (int32)((uint64)(int64)c / n) through csrc/bs_centerdiv.h."""
        c = np.ascontiguousarray(c, dtype=np.int32)
        n = np.ascontiguousarray(n, dtype=np.uint32)
        out = np.empty_like(c)
        self._check(self._L.bs_selftest_center_div(self._h, c.ctypes.data, n.ctypes.data, out.ctypes.data, len(c)))
        return out

    def selftest_forge_next(self, mode):
        """The next region grow corrupts one finished plane (1: duplicated entry, 2: normal off by an ulp);
        timings()['validation_rejects'] must then be >= 1 and the result still exact."""
        self._check(self._L.bs_selftest_forge_next(self._h, int(mode)))

    def set_audit(self, on=True):
        """Replay every plane attempt against the final owners after each region grow (see bs_set_audit);
        timings()['audit_mismatches'] must be 0."""
        self._check(self._L.bs_set_audit(self._h, 1 if on else 0))

    # ---- building blocks of the component-sharded stage 3 (device pointers) ----
    def cc_hook_dev(self, d_rows, d_gidx, m, k, d_parent, n_total) -> int:
        """One hooking step of the distributed union-find (bs_cc_hook_dev); returns the unions performed."""
        h = C.c_int64(0)
        self._check(self._L.bs_cc_hook_dev(self._h, d_rows or None, d_gidx or None, m, k, d_parent, n_total, C.byref(h)))
        return h.value

    def owner_fetch_dev(self, d_owner):
        """owner[p] of the last speculative grow on this context, in the caller's index order (-1: unlabelled)."""
        self._check(self._L.bs_owner_fetch_dev(self._h, d_owner))

    def plane_seeds_dev(self, d_seeds, cap) -> int:
        """Seeds of the committed planes of the last speculative grow (ascending) into d_seeds; returns their number."""
        npl = C.c_int32(0)
        self._check(self._L.bs_plane_seeds_dev(self._h, d_seeds or None, cap, C.byref(npl)))
        return npl.value

    def sync(self):
        self._check(self._L.bs_stream_sync(self._h))

    def labels_from_owner_dev(self, d_owner, n, d_seeds, n_seeds, d_plane_idx):
        self._check(self._L.bs_labels_from_owner_dev(self._h, d_owner or None, n, d_seeds or None, n_seeds, d_plane_idx or None))

    def remap_rows_dev(self, d_rows, n_rows, k, d_sorted_gidx, n, d_out) -> int:
        miss = C.c_int32(0)
        self._check(self._L.bs_remap_rows_dev(self._h, d_rows or None, n_rows, k, d_sorted_gidx or None, n, d_out or None,
                                              C.byref(miss)))
        return miss.value

    # ---- multi-GPU entry point of the C ABI (bs_segment_sharded) ----
    def segment_sharded(self, comm_ops, d_xyz, d_gidx, m, n_total, d_plane_idx, params, halo=0.0) -> dict:
        """ONE cloud spread over the ranks of `comm_ops` (a _lib.CommOps: bs_comm_rccl / bs_comm_local_create);
        this rank passes its m points (device pointers).  d_plane_idx [n_total] receives the labels of the whole
        cloud.  Returns bs_shard_info as a dict."""
        inf = _lib.ShardInfo()
        self._check(self._L.bs_segment_sharded(self._h, C.byref(comm_ops) if comm_ops is not None else None, d_xyz or None,
                                               d_gidx or None, m, n_total, C.byref(params), float(halo), d_plane_idx,
                                               C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in _lib.ShardInfo._fields_}

    def sharded_planes_fetch(self):
        """The planes THIS rank grew in the last segment_sharded, with global ids and global point indices."""
        P = Planes()
        self._check(self._L.bs_sharded_planes_fetch(self._h, C.byref(P)))
        planes = _planes_to_list(P)
        self._L.bs_planes_free(C.byref(P))
        return planes

    def planes_fetch(self):
        P = Planes()
        self._check(self._L.bs_planes_fetch(self._h, C.byref(P)))
        planes = _planes_to_list(P)
        self._L.bs_planes_free(C.byref(P))
        return planes


def grid_dims(extent, bin=100):
    """(width, height) of the 2-D raster (TMC3.cpp:75-76)."""
    ext = np.ascontiguousarray(extent, dtype=np.int32)
    w, h = C.c_int32(0), C.c_int32(0)
    rc = _lib.load().bs_grid_dims(ext.ctypes.data, bin, C.byref(w), C.byref(h))
    if rc != 0:
        raise ValueError("bs_grid_dims: invalid extent / bin")
    return w.value, h.value


def save_image(image, prefix):
    """buildingSeg::save_image (TMC3.cpp:81-117): three 8-bit RGB PNGs -- mean height in the
    red channel, density in the green channel, and the (never written, all zero) third
    channel in green.  Each channel is scaled by its maximum and truncated to uint8.
    The reference's file names are prefix + a Chinese caption; this port uses ASCII
    suffixes (height / density / density_height).  Returns the three uint8 arrays."""
    img = np.asarray(image, dtype=np.float64)
    h, w, _ = img.shape
    mx = [max(0.0, float(img[..., c].max())) for c in range(3)]  # `max` starts at 0 (TMC3.cpp:85)
    outs = []
    for c, slot, name in ((0, 0, "height"), (1, 1, "density"), (2, 1, "density_height")):
        out = np.zeros((h, w, 3), dtype=np.uint8)
        if mx[c] != 0:
            out[..., slot] = (255.0 * (1.0 * img[..., c] / mx[c])).astype(np.uint8)
        write_png(prefix + name + ".png", out)
        outs.append(out)
    return outs


def write_png(path, rgb):
    """Minimal 8-bit RGB PNG writer (zlib from the standard library)."""
    import struct
    import zlib
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


_DEFAULT_CTX = None


def _ctx() -> Context:
    global _DEFAULT_CTX
    if _DEFAULT_CTX is None:
        _DEFAULT_CTX = Context(0)
    return _DEFAULT_CTX


def get_normal_and_k_neighbor(point_cloud, k: int = 15, ctx: Context | None = None):
    """get_Normal_and_K_neighbor<K>(pointCloud, normal, neigh) (my_function.h:48-85).
    Returns (normal [n,3] f64, neigh [n,K] int32).  The reference's stray
    output.ply side effect (my_function.h:81) is not reproduced."""
    neigh, normals = (ctx or _ctx()).knn_normals(point_cloud, default_params(k=k))
    return normals, neigh


class SegPlane:
    """class seg_plane (my_function.h:89-123)."""

    def __init__(self, point_cloud, normal, neigh, num_neigh: int, ctx: Context | None = None):
        self.cloud = np.ascontiguousarray(point_cloud, dtype=np.int32)
        self.normal = np.ascontiguousarray(normal, dtype=np.float64)
        self.neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        self.K = int(num_neigh)
        self.planeIdx = np.full(len(self.cloud), -1, dtype=np.int32)  # my_function.h:103
        self._ctx = ctx or _ctx()

    def get_planes(self):
        """seg_plane::get_planes (my_function.cpp:180-217)."""
        self.planeIdx, planes = self._ctx.region_grow(self.cloud, self.normal, self.neigh,
                                                      default_params(k=self.K))
        return planes

    def set_plane_color(self, planes, rand=None):
        """seg_plane::set_plane_color (my_function.cpp:260-275): colours [n,3]
        uint16 in the reference's G,B,R slots.  ``rand`` is a callable returning
        the next rand() value (default: glibc rand() seeded with 1, as an
        unseeded C program)."""
        if rand is None:
            libc = C.CDLL(None)
            libc.srand(1)
            rand = libc.rand
        colors = np.zeros((len(self.cloud), 3), dtype=np.uint16)
        for p in planes:
            col = [55 + rand() % 200, 55 + rand() % 200, 55 + rand() % 200]
            colors[p.pointIdx] = col
        return colors
