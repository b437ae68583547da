"""Host-side mirror of the reference renderer component `stm::BDPT` (src/Node/BDPT.{hpp,cpp})
above the C ABI of libstratum_hip.so.

Same knobs, same defaults, same flag resolution: the constructor sets the default sampling
flags and push constants of BDPT.cpp:55-76 and parses the reference's `--key=value` arguments
(`minPathVertices`, `maxPathVertices`, `maxDiffuseVertices`, repeated `bdptFlag=[~]name`,
BDPT.cpp:78-127); `update()` is where the reference binds the scene descriptors
(BDPT.cpp:341-421) and here uploads the scene arrays; `render()` replaces the recorded
dispatch sequence (BDPT.cpp:423-838) by one sthip_render call.
"""
import ctypes as C
import os

import numpy as np

from . import wire
from ._lib import StratumHipError, lib


def _flag_key(name):
    # to_string(BDPTFlagBits) lower-cased with spaces removed (BDPT.cpp:107-116)
    table = {
        "ePerformanceCounters": "performancecounters",
        "eRemapThreads": "remapthreads",
        "eCoherentRR": "coherentrr",
        "eCoherentSampling": "coherentsampling",
        "eFlipTriangleUVs": "fliptriangleuvs",
        "eFlipNormalMaps": "flipnormalmaps",
        "eAlphaTest": "alphatest",
        "eNormalMaps": "normalmaps",
        "eShadingNormalShadowFix": "shadingnormalshadowfix",
        "eRayCones": "raycones",
        "eSampleBSDFs": "samplebsdfs",
        "eNEE": "nee",
        "eNEEReservoirs": "neereservoirs",
        "eNEEReservoirReuse": "neereservoirreuse",
        "eMIS": "mis",
        "eSampleLightPower": "samplelightpower",
        "eUniformSphereSampling": "uniformspheresampling",
        "ePresampleLights": "presamplelights",
        "eDeferShadowRays": "defershadowrays",
        "eConnectToViews": "connecttoviews",
        "eConnectToLightPaths": "connecttolightpaths",
        "eLVC": "lightvertexcache",
        "eLVCReservoirs": "lvcreservoirs",
        "eLVCReservoirReuse": "lvcreservoirreuse",
        "eHashGridJitter": "jitterhashgridlookups",
        "eSampleEnvironmentMapDirectly": "sampleenvironmentmapdirectly",
    }
    return table[name]


_FLAG_BY_KEY = {_flag_key(n): i for i, n in enumerate(wire.FLAG_NAMES)}


def known_flag(arg):
    """Whether --bdptFlag would act on `arg` (BDPT.cpp:94-127 ignores a name it does not know; a tool that measures may not)."""
    return bool(arg) and arg.lstrip("~!").lower() in _FLAG_BY_KEY


class BDPT:
    def __init__(self, device=0, args=None):
        self._lib = lib()
        h = C.c_void_p()
        rc = self._lib.sthip_create(device, C.byref(h))
        if rc != 0:
            raise StratumHipError("sthip_create(%d) failed (%d): %s" % (device, rc, self._lib.sthip_last_error(None).decode()))
        self._h = h
        self.device = device
        self.mSamplingFlags = wire.DEFAULT_SAMPLING_FLAGS
        self.mPushConstants = wire.default_push_constants(0, 0, 0)
        self.mPushConstants.gLightPathCount = 64  # BDPT.cpp:70: the size of the light vertex cache unless --lightPathCount says otherwise (without eLVC: one path per pixel, :469-470)
        self._scene = None
        self._prev_result = None
        args = args or {}
        for key, field in (
            ("minPathVertices", "gMinPathVertices"),
            ("maxPathVertices", "gMaxPathVertices"),
            ("maxDiffuseVertices", "gMaxDiffuseVertices"),
            ("maxNullCollisions", "gMaxNullCollisions"),
            ("lightPresampleTileSize", "gLightPresampleTileSize"),
            ("lightPresampleTileCount", "gLightPresampleTileCount"),
            ("lightPathCount", "gLightPathCount"),
            ("reservoirM", "gReservoirM"),
            ("reservoirMaxM", "gReservoirMaxM"),
            ("reservoirSpatialM", "gReservoirSpatialM"),
            ("hashGridBucketCount", "gHashGridBucketCount"),
        ):
            if key in args:
                setattr(self.mPushConstants, field, int(args[key]))
        if "environmentSampleProbability" in args:  # BDPT.cpp:82
            self.mPushConstants.gEnvironmentSampleProbability = float(args["environmentSampleProbability"])
        for a in args.get("bdptFlag", []):
            self.set_flag(a)

    # BDPT.cpp:94-127
    def set_flag(self, arg):
        if not arg:
            return
        on = True
        if arg[0] in "~!":
            on, arg = False, arg[1:]
        bit = _FLAG_BY_KEY.get(arg.lower())
        if bit is None:
            if os.environ.get("STHIP_STRICT_FLAGS"):  # the measuring tools and bench.py set it: a misspelt flag is an error there
                raise ValueError("unknown --bdptFlag name %r" % (arg,))
            return  # the reference silently ignores unknown names
        if on:
            self.mSamplingFlags |= 1 << bit
        else:
            self.mSamplingFlags &= ~(1 << bit)

    def _check(self, rc, what):
        if rc != 0:
            raise StratumHipError("%s failed (%d): %s" % (what, rc, self._lib.sthip_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sthip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- BDPT::update: (re)bind the scene ----
    def update(self, scene):
        d = scene.desc()
        self._check(self._lib.sthip_scene_upload(self._h, C.byref(d)), "sthip_scene_upload")
        self._scene = scene

    def update_transforms(self, scene):
        """Only instance transforms changed since update(scene) (SceneData.set_instance_transform): the top level of the
        acceleration structure is rebuilt, the bottom levels stay in HBM (the reference's cached BLASes, Scene.cpp:435-459).
        Raises StratumHipError (unsupported) if an instance of the merged identity-transform mesh moved: call update()."""
        rc = self._lib.sthip_scene_update_transforms(
            self._h, wire.ptr(scene.transforms), wire.ptr(scene.inverse_transforms), wire.ptr(scene.motion_transforms), scene.instances.shape[0]
        )
        self._check(rc, "sthip_scene_update_transforms")
        self._scene = scene

    def set_stream(self, stream_handle):
        self._check(self._lib.sthip_set_stream(self._h, C.c_void_p(stream_handle)), "sthip_set_stream")

    def set_option(self, name, value):
        self._check(self._lib.sthip_set_option(self._h, name.encode(), int(value)), "sthip_set_option")

    CEILINGS = {"triad": 0, "node_gather_table": 1, "node_gather_l2": 2, "node_gather_l1": 3}

    def measure_ceiling(self, kind):
        """Measured memory-system ceiling in GB/s (include/sthip.h: sthip_measure_ceiling)."""
        v = C.c_double(0.0)
        self._check(self._lib.sthip_measure_ceiling(self._h, self.CEILINGS[kind], C.byref(v)), "sthip_measure_ceiling")
        return float(v.value)

    def stats(self):
        s = wire.Stats()
        self._check(self._lib.sthip_get_stats(self._h, C.byref(s)), "sthip_get_stats")
        return {f: (list(getattr(s, f)) if hasattr(getattr(s, f), "__len__") else getattr(s, f)) for f, _ in wire.Stats._fields_}

    def push_constants(self, frame):
        pc = wire.BDPTPushConstants.from_buffer_copy(self.mPushConstants)
        pc.gOutputExtent[0], pc.gOutputExtent[1] = frame.width, frame.height
        pc.gViewCount = frame.views.shape[0]
        pc.gLightCount = self._scene.light_count
        if not (self.mSamplingFlags >> wire.FLAG_NAMES.index("eLVC")) & 1:  # BDPT.cpp:469-470: with the cache on it stays the user's value
            pc.gLightPathCount = frame.width * frame.height
        # BDPT.cpp:393,486-496
        pc.gEnvironmentMaterialAddress = self._scene.environment_address
        if self._scene.environment_address == 0xFFFFFFFF:
            pc.gEnvironmentSampleProbability = 0.0
        if pc.gLightCount == 0:
            pc.gEnvironmentSampleProbability = 1.0
        if not (self._scene.scene_flags & wire.BDPT_FLAG_HAS_MEDIA):  # BDPT.cpp:497-500
            pc.gMaxNullCollisions = 0
        return pc

    # ---- BDPT::render ----
    def render(self, frame, seed_begin=0, seed_count=1, aovs=True, device_outputs=None, packed_tiles=False, debug_mode=0, debug_image=None, host_outputs=None):
        """Host outputs by default (dict of numpy arrays; `host_outputs` = such a dict from an earlier call: its arrays are
        written in place instead of fresh ones — a caller's own buffers). `device_outputs` = dict of raw device pointers
        {"radiance": ptr, ["albedo", "visibility", "depth", "prev_uv", "ray_count"]} renders in place on the
        GPU without synchronising. packed_tiles: "radiance" holds only this shard's tiles in slot order
        (shard_slot_count() float4 entries) — the form ranks exchange, see assemble_tiles."""
        if self._scene is None:
            raise StratumHipError("BDPT.render before BDPT.update(scene)")
        pc = self.push_constants(frame)
        if self._scene.volumes:  # gViewMediumInstances, BDPT.cpp:456-466
            frame.view_medium_instances = self._scene.view_medium_instances(frame.view_transforms)
        fd = frame.desc()
        o = wire.Outputs()
        out = None
        o.radiance_layout = wire.LAYOUT_SHARD_TILES if packed_tiles else wire.LAYOUT_IMAGE
        if device_outputs is not None:
            o.device_ptrs = 1
            o.gRadiance = device_outputs["radiance"]
            o.gAlbedo = device_outputs.get("albedo")
            o.gVisibility = device_outputs.get("visibility")
            o.gDepth = device_outputs.get("depth")
            o.gPrevUVs = device_outputs.get("prev_uv")
            o.gRayCount = device_outputs.get("ray_count")
            if debug_mode:
                o.debug_mode = debug_mode
                o.gDebugImage = device_outputs["debug"]
        else:
            W, H = frame.width, frame.height
            if host_outputs is None:
                out = {"radiance": np.zeros((self.shard_slot_count(frame), 4) if packed_tiles else (H, W, 4), np.float32), "ray_count": np.zeros(2, np.uint64)}
            else:
                out = {k: host_outputs[k] for k in ("radiance", "ray_count")}
                if out["radiance"].nbytes != (self.shard_slot_count(frame) if packed_tiles else H * W) * 16 or out["radiance"].dtype != np.float32 or not out["radiance"].flags["C_CONTIGUOUS"]:
                    raise ValueError("host_outputs['radiance'] does not fit the frame")
            o.device_ptrs = 0
            o.gRadiance = wire.ptr(out["radiance"])
            o.gRayCount = wire.ptr(out["ray_count"])
            if debug_mode:  # BDPTDebugMode -> gDebugImage (in / out: a copy of what the caller passes, or zeros)
                out["debug"] = np.ascontiguousarray(debug_image, np.float32).copy() if debug_image is not None else np.zeros((H, W, 4), np.float32)
                o.debug_mode = debug_mode
                o.gDebugImage = wire.ptr(out["debug"])
            if aovs and host_outputs is not None:
                for k, dtype, per_pixel in (("albedo", np.float32, 16), ("visibility", wire.VisibilityInfo, None), ("depth", wire.DepthInfo, None), ("prev_uv", np.float32, 8)):
                    a = host_outputs[k]
                    if a.dtype != dtype or a.nbytes != H * W * (per_pixel or a.dtype.itemsize) or not a.flags["C_CONTIGUOUS"]:
                        raise ValueError("host_outputs[%r] does not fit the frame" % k)
                    out[k] = a
            elif aovs:
                out["albedo"] = np.zeros((H, W, 4), np.float32)
                out["visibility"] = np.zeros((H, W), wire.VisibilityInfo)
                out["depth"] = np.zeros((H, W), wire.DepthInfo)
                out["prev_uv"] = np.zeros((H, W, 2), np.float32)
            if aovs:
                o.gAlbedo = wire.ptr(out["albedo"])
                o.gVisibility = wire.ptr(out["visibility"])
                o.gDepth = wire.ptr(out["depth"])
                o.gPrevUVs = wire.ptr(out["prev_uv"])
        rc = self._lib.sthip_render(self._h, C.byref(pc), self.mSamplingFlags, self._scene.scene_flags, C.byref(fd), seed_begin, seed_count, C.byref(o))
        self._check(rc, "sthip_render")
        if out is not None:
            self._prev_result = out["radiance"]
        return out

    # ---- multi-GPU assembly (include/sthip.h: sthip_shard_slot_count / sthip_assemble_tiles) ----
    def set_shard(self, rank, count, tile_w=64, tile_h=32):
        self._check(self._lib.sthip_set_shard(self._h, rank, count, tile_w, tile_h), "sthip_set_shard")
        self._shard = (rank, count, tile_w, tile_h)

    def shard_slot_count(self, frame, rank=None):
        r, n, tw, th = getattr(self, "_shard", (0, 1, 64, 32))
        return int(self._lib.sthip_shard_slot_count(frame.width, frame.height, r if rank is None else rank, n, tw, th))

    def assemble_tiles(self, frame, packed_ptr, rank_stride, frame_ptr):
        """packed_ptr: device buffer with rank r's tiles at r * rank_stride float4 entries; frame_ptr: W x H RGBA32F."""
        _, n, tw, th = getattr(self, "_shard", (0, 1, 64, 32))
        self._check(self._lib.sthip_assemble_tiles(self._h, packed_ptr, rank_stride, n, tw, th, frame.width, frame.height, frame_ptr), "sthip_assemble_tiles")

    def radiance_to_sums(self, image_ptr, entries, back=False):
        """(mean over the seeds, their number) -> (sum, number) in place, or back: what the seed-split replica mode reduces
        (device pointer; include/sthip.h: sthip_radiance_to_sums)."""
        self._check(self._lib.sthip_radiance_to_sums(self._h, image_ptr, entries, 1 if back else 0), "sthip_radiance_to_sums")

    def pack_tiles(self, frame, image_ptr, entry_bytes, packed_ptr):
        """This shard's tiles of a W x H image of entry_bytes per pixel (a G-buffer output of render) in slot order: what the
        ranks exchange (device pointers; include/sthip.h: sthip_pack_tiles)."""
        self._check(self._lib.sthip_pack_tiles(self._h, image_ptr, frame.width, frame.height, entry_bytes, packed_ptr), "sthip_pack_tiles")

    def assemble_tiles_bytes(self, frame, packed_ptr, rank_stride, frame_ptr, entry_bytes):
        """assemble_tiles for entries of entry_bytes (albedo / depth 16, visibility / prev-uv 8)."""
        n, tw, th = self._shard[1], self._shard[2], self._shard[3]
        self._check(self._lib.sthip_assemble_tiles_bytes(self._h, packed_ptr, rank_stride, n, tw, th, frame.width, frame.height, entry_bytes, frame_ptr), "sthip_assemble_tiles_bytes")

    def prev_result(self):  # BDPT.hpp:18
        return self._prev_result

    # ---- the traversal contract on its own ----
    def trace(self, rays, any_hit=False, alpha_test=False, flip_uvs=False):
        rays = np.ascontiguousarray(rays, dtype=wire.Ray)
        hits = np.zeros(rays.shape[0], wire.Hit)
        mode = (1 if any_hit else 0) | (2 if alpha_test else 0) | (4 if flip_uvs else 0)
        rc = self._lib.sthip_trace_rays(self._h, wire.ptr(rays), rays.shape[0], wire.ptr(hits), mode, 0)
        self._check(rc, "sthip_trace_rays")
        return hits
