"""Loader of libstratum_hip.so (the C ABI declared in include/sthip.h).

The library is built in-tree by __graft_entry__.build() / `make -C stratum_amd/csrc`. There is no
fallback of any kind: if the shared object is missing or a HIP device is absent, every entry point
of this package raises.
"""
import ctypes as C
import importlib.util
import os
import sys

from . import wire

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STHIP_LIB") or os.path.join(_HERE, "libstratum_hip.so")  # STHIP_LIB: another build of the same library (kernel experiments)

# every symbol include/sthip.h declares
EXPORTS = [
    "sthip_abi_version",
    "sthip_create",
    "sthip_destroy",
    "sthip_last_error",
    "sthip_set_stream",
    "sthip_scene_upload",
    "sthip_scene_update_transforms",
    "sthip_render",
    "sthip_set_shard",
    "sthip_trace_rays",
    "sthip_get_stats",
    "sthip_set_option",
    "sthip_shard_slot_count",
    "sthip_assemble_tiles",
    "sthip_assemble_tiles_bytes",
    "sthip_pack_tiles",
    "sthip_radiance_to_sums",
    "sthip_accumulate",
    "sthip_tonemap",
    "sthip_image_compare",
    "sthip_write_hdr",
    "sthip_measure_ceiling",
]

_lib = None


class StratumHipError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own libamdhip64.so / libhsa-runtime64.so under torch/lib (sonames
    libamdhip64.so.7, libhsa-runtime64.so.1, the ones this library links). If torch is imported first the loader
    gives this library torch's runtime and the two share one; if this library came first the system runtime is
    loaded, torch then loads its own copy by file name, and the second HSA runtime in the process finds no GPU.
    Loading torch's copy first (without importing torch) makes the order irrelevant: one HIP runtime either way.
    Without torch installed nothing happens and the system ROCm runtime is used."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(hip):
        C.CDLL(hip, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    if os.environ.get("STHIP_LIB"):  # development: A/B of kernel variants built side by side (tools/ab_variants.py)
        LIB_PATH = os.environ["STHIP_LIB"]
    if not os.path.exists(LIB_PATH):
        raise StratumHipError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
        )
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    L.sthip_abi_version.restype = C.c_int
    L.sthip_create.restype = C.c_int
    L.sthip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.sthip_destroy.restype = None
    L.sthip_destroy.argtypes = [C.c_void_p]
    L.sthip_last_error.restype = C.c_char_p
    L.sthip_last_error.argtypes = [C.c_void_p]
    L.sthip_set_stream.restype = C.c_int
    L.sthip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.sthip_scene_upload.restype = C.c_int
    L.sthip_scene_upload.argtypes = [C.c_void_p, C.POINTER(wire.SceneDesc)]
    L.sthip_render.restype = C.c_int
    L.sthip_render.argtypes = [
        C.c_void_p,
        C.POINTER(wire.BDPTPushConstants),
        C.c_uint32,
        C.c_uint32,
        C.POINTER(wire.FrameDesc),
        C.c_uint32,
        C.c_uint32,
        C.POINTER(wire.Outputs),
    ]
    L.sthip_set_shard.restype = C.c_int
    L.sthip_set_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.sthip_trace_rays.restype = C.c_int
    L.sthip_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    L.sthip_get_stats.restype = C.c_int
    L.sthip_get_stats.argtypes = [C.c_void_p, C.POINTER(wire.Stats)]
    L.sthip_set_option.restype = C.c_int
    L.sthip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.sthip_shard_slot_count.restype = C.c_uint32
    L.sthip_shard_slot_count.argtypes = [C.c_uint32] * 6
    L.sthip_assemble_tiles.restype = C.c_int
    L.sthip_assemble_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sthip_assemble_tiles_bytes.restype = C.c_int
    L.sthip_assemble_tiles_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sthip_radiance_to_sums.restype = C.c_int
    L.sthip_radiance_to_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    L.sthip_pack_tiles.restype = C.c_int
    L.sthip_pack_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sthip_accumulate.restype = C.c_int
    L.sthip_accumulate.argtypes = [C.c_void_p, C.POINTER(wire.AccumulateDesc)]
    L.sthip_tonemap.restype = C.c_int
    L.sthip_tonemap.argtypes = [C.c_void_p, C.POINTER(wire.TonemapDesc)]
    L.sthip_image_compare.restype = C.c_int
    L.sthip_image_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.sthip_write_hdr.restype = C.c_int
    L.sthip_write_hdr.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sthip_measure_ceiling.restype = C.c_int
    L.sthip_measure_ceiling.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
    _lib = L
    return L
