"""ctypes loader of libbuildingsegment_hip.so (the C ABI of include/bs_api.h).

Fails loudly when the HIP library is missing or cannot be loaded: there is no
CPU fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# BS_LIB_PATH: developer override (A/B runs of two builds of the same HIP library on one box)
LIB_PATH = os.environ.get("BS_LIB_PATH") or os.path.join(HERE, "libbuildingsegment_hip.so")

BS_OK = 0
STATUS = {0: "BS_OK", -1: "BS_ERR_INVALID", -2: "BS_ERR_RANGE", -3: "BS_ERR_NOMEM", -4: "BS_ERR_HIP",
          -5: "BS_ERR_NO_DEVICE", -6: "BS_ERR_INTERNAL", -7: "BS_ERR_UNCERTIFIED"}

# every symbol include/bs_api.h declares
EXPORTS = ["bs_api_version", "bs_sizeof_timings", "bs_strerror", "bs_params_default", "bs_create", "bs_destroy", "bs_last_error",
           "bs_set_stream", "bs_get_timings", "bs_knn_normals", "bs_knn_normals_halo", "bs_region_grow", "bs_segment",
           "bs_planes_free", "bs_plane_colors", "bs_knn_normals_dev", "bs_region_grow_dev",
           "bs_segment_dev", "bs_planes_fetch", "bs_shift_to_origin_dev", "bs_plane_colors_dev",
           "bs_selftest_center_div", "bs_selftest_forge_next", "bs_set_audit", "bs_ingest_dev", "bs_grid_dims", "bs_grid_picture", "bs_grid_picture_dev",
           "bs_cc_hook_dev", "bs_owner_fetch_dev", "bs_labels_from_owner_dev", "bs_remap_rows_dev",
           "bs_plane_seeds_dev", "bs_stream_sync", "bs_comm_rccl", "bs_comm_rccl_unique_id", "bs_comm_rccl_init",
           "bs_comm_rccl_destroy", "bs_comm_local_create", "bs_comm_local_destroy", "bs_segment_sharded",
           "bs_sharded_planes_fetch"]


class Params(C.Structure):
    _fields_ = [("k", C.c_int32), ("max_nn", C.c_int32), ("radius", C.c_double),
                ("th_thickness", C.c_int32), ("th_point_count", C.c_int32), ("cos_th", C.c_double),
                ("cell_size", C.c_int32), ("rg_mode", C.c_int32)]


class Planes(C.Structure):
    _fields_ = [("n_planes", C.c_int32), ("id", C.POINTER(C.c_int32)), ("normal", C.POINTER(C.c_double)),
                ("center", C.POINTER(C.c_int32)), ("offset", C.POINTER(C.c_int64)),
                ("point_idx", C.POINTER(C.c_int32))]


class Timings(C.Structure):
    _fields_ = [("grid_ms", C.c_double), ("knn_ms", C.c_double), ("grow_ms", C.c_double),
                ("total_ms", C.c_double), ("largest_plane", C.c_int64), ("n_seed_attempts", C.c_int64),
                ("n_fallback_queries", C.c_int64), ("rg_rounds", C.c_int64), ("grow_kernel_ms", C.c_double),
                ("grow_kernel_launches", C.c_int64), ("grow_setup_ms", C.c_double),
                ("validation_rejects", C.c_int64), ("forged_seed", C.c_int64), ("forged_refused", C.c_int64),
                ("audit_attempts", C.c_int64), ("audit_mismatches", C.c_int64), ("audit_ms", C.c_double),
                ("tie_rows", C.c_int64), ("rej_v1_robbed", C.c_int64), ("rej_v1_tag", C.c_int64), ("rej_v1_dup", C.c_int64),
                ("rej_v3_state", C.c_int64), ("incons_seed", C.c_int64), ("incons_list", C.c_int64),
                ("incons_log", C.c_int64)]

API_VERSION = 5  # BS_API_VERSION of include/bs_api.h this loader mirrors


class CommOps(C.Structure):
    """bs_comm_ops (include/bs_api.h): the three collectives bs_segment_sharded is built on."""
    _fields_ = [("handle", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32),
                ("all_reduce", C.c_void_p), ("all_gather", C.c_void_p), ("all_to_all_v", C.c_void_p)]


class ShardInfo(C.Structure):
    _fields_ = [("n_own", C.c_int64), ("n_local", C.c_int64), ("n_grow", C.c_int64), ("components", C.c_int64),
                ("planes_total", C.c_int64), ("cc_iterations", C.c_int32), ("halo_retries", C.c_int32),
                ("halo_mm", C.c_double), ("ms_partition", C.c_double), ("ms_halo", C.c_double), ("ms_knn", C.c_double),
                ("ms_components", C.c_double), ("ms_redistribute", C.c_double), ("ms_grow", C.c_double),
                ("ms_labels", C.c_double)]


class BsError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"{STATUS.get(status, status)}: {detail}")


_LIB = None


def load():
    """dlopen the HIP library; raise if it is absent (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m buildingsegment_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    L.bs_api_version.restype = C.c_int
    L.bs_sizeof_timings.restype = C.c_int64
    if L.bs_api_version() != API_VERSION or L.bs_sizeof_timings() != C.sizeof(Timings):
        raise ImportError(f"{LIB_PATH} is API version {L.bs_api_version()} (bs_timings: {L.bs_sizeof_timings()} bytes), this "
                          f"loader mirrors version {API_VERSION} ({C.sizeof(Timings)} bytes): rebuild the library")
    vp, ip, dp, lp = C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)
    pp = C.POINTER(Params)
    L.bs_api_version.restype = C.c_int
    L.bs_strerror.restype = C.c_char_p
    L.bs_strerror.argtypes = [C.c_int]
    L.bs_params_default.argtypes = [pp]
    L.bs_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.bs_destroy.argtypes = [vp]
    L.bs_destroy.restype = None
    L.bs_last_error.argtypes = [vp]
    L.bs_last_error.restype = C.c_char_p
    L.bs_set_stream.argtypes = [vp, vp]
    L.bs_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.bs_knn_normals.argtypes = [vp, ip, C.c_int64, pp, ip, dp]
    L.bs_knn_normals_halo.argtypes = [vp, ip, ip, C.c_int64, C.c_int64, pp, ip, dp, C.c_double, lp]
    L.bs_region_grow.argtypes = [vp, ip, dp, ip, C.c_int64, pp, ip, C.POINTER(Planes)]
    L.bs_segment.argtypes = [vp, ip, C.c_int64, pp, ip, dp, ip, C.POINTER(Planes)]
    L.bs_planes_free.argtypes = [C.POINTER(Planes)]
    L.bs_planes_free.restype = None
    L.bs_plane_colors.argtypes = [C.POINTER(Planes), ip, C.c_int64, vp]
    L.bs_knn_normals_dev.argtypes = [vp, ip, ip, C.c_int64, C.c_int64, C.c_int64, pp, ip, dp, C.c_double, lp]
    L.bs_region_grow_dev.argtypes = [vp, ip, dp, ip, C.c_int64, pp, ip]
    L.bs_segment_dev.argtypes = [vp, ip, C.c_int64, pp, ip, dp, ip]
    L.bs_planes_fetch.argtypes = [vp, C.POINTER(Planes)]
    L.bs_shift_to_origin_dev.argtypes = [vp, ip, C.c_int64, ip]
    L.bs_ingest_dev.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                C.c_int32, ip, ip]
    L.bs_plane_colors_dev.argtypes = [vp, ip, C.c_int32, C.c_int64, vp]
    L.bs_selftest_forge_next.argtypes = [vp, C.c_int]
    L.bs_set_audit.argtypes = [vp, C.c_int]
    L.bs_selftest_center_div.argtypes = [vp, ip, vp, ip, C.c_int64]
    L.bs_grid_dims.argtypes = [ip, C.c_int32, ip, ip]
    L.bs_grid_picture.argtypes = [vp, ip, C.c_int64, ip, C.c_int32, C.c_int32, dp, dp]
    L.bs_grid_picture_dev.argtypes = [vp, ip, C.c_int64, ip, C.c_int32, C.c_int32, dp, dp]
    L.bs_cc_hook_dev.argtypes = [vp, ip, ip, C.c_int64, C.c_int32, ip, C.c_int64, lp]
    L.bs_owner_fetch_dev.argtypes = [vp, ip]
    L.bs_plane_seeds_dev.argtypes = [vp, ip, C.c_int64, C.POINTER(C.c_int32)]
    L.bs_stream_sync.argtypes = [vp]
    L.bs_labels_from_owner_dev.argtypes = [vp, ip, C.c_int64, ip, C.c_int32, ip]
    L.bs_remap_rows_dev.argtypes = [vp, ip, C.c_int64, C.c_int32, ip, C.c_int64, ip, C.POINTER(C.c_int32)]
    L.bs_comm_rccl.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(CommOps)]
    L.bs_comm_rccl_unique_id.argtypes = [C.c_char_p]
    L.bs_comm_rccl_init.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.bs_comm_rccl_destroy.argtypes = [vp]
    L.bs_comm_local_create.argtypes = [C.c_int32, C.POINTER(CommOps)]
    L.bs_comm_local_destroy.argtypes = [C.POINTER(CommOps)]
    L.bs_comm_local_destroy.restype = None
    L.bs_segment_sharded.argtypes = [vp, C.POINTER(CommOps), ip, ip, C.c_int64, C.c_int64, pp, C.c_double, ip,
                                     C.POINTER(ShardInfo)]
    L.bs_sharded_planes_fetch.argtypes = [vp, C.POINTER(Planes)]
    _LIB = L
    return L
