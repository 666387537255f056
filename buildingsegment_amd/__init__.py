"""MI355X-native buildingSegment hot path (kNN -> PCA normals -> region grow).

The compute lives in ``csrc/`` (hand-written HIP for gfx950) behind the C ABI of
``include/bs_api.h``; this package is the thin Python host mirror of the
reference's call sites (/root/reference/tmc3/TMC3.cpp:213-218).
"""
__version__ = "0.1.0"
