"""What the reference does with the renderer's output, on the GPU: the tone-map block of BDPT::render
(src/Node/BDPT.cpp:783-815 -> kernels/tonemap.hlsl), the ImageComparer node (src/Node/ImageComparer.cpp:61-90 ->
kernels/image_compare.hlsl) and the "Export HDR" button (BDPT.cpp:313-337). Thin host code over the C ABI
(sthip_tonemap / sthip_image_compare / sthip_write_hdr); there is no CPU implementation behind it."""
import ctypes as C

import numpy as np

from . import _lib, wire


def _mode(table, m):
    if isinstance(m, str):
        if m not in table:
            raise ValueError("unknown mode %r (one of %s)" % (m, ", ".join(table)))
        return table[m]
    return int(m)


def _rgba(a, name):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("%s must be an (H, W, 4) float32 image" % name)
    return a


def accumulate_desc(frame_out, history, views, reprojection, demodulate_albedo, history_limit, instance_index_map, keep):
    """Fills a wire.AccumulateDesc from a render's output dict (`frame_out`: radiance, albedo, visibility, depth,
    prev_uv) and the previous state (`history`: accum_color, accum_moments, visibility, depth). `keep` receives the
    arrays the descriptor points at. Returns (desc, accum_color, accum_moments)."""
    rad = _rgba(frame_out["radiance"], "radiance")
    H, W = rad.shape[0], rad.shape[1]
    d = wire.AccumulateDesc()
    d.width, d.height = W, H
    v = np.ascontiguousarray(views, dtype=wire.ViewData)
    d.view_count = v.shape[0]
    d.reprojection = int(bool(reprojection))
    d.demodulate_albedo = int(bool(demodulate_albedo))
    d.history_limit = float(history_limit)
    d.device_ptrs = 0
    out_c = np.zeros((H, W, 4), np.float32)
    out_m = np.zeros((H, W, 2), np.float32)
    arrays = {
        "gViews": v,
        "gRadiance": rad,
        "gAlbedo": np.ascontiguousarray(frame_out["albedo"], np.float32) if "albedo" in frame_out else None,
        "gVisibility": np.ascontiguousarray(frame_out["visibility"], wire.VisibilityInfo) if "visibility" in frame_out else None,
        "gDepth": np.ascontiguousarray(frame_out["depth"], wire.DepthInfo) if "depth" in frame_out else None,
        "gPrevUVs": np.ascontiguousarray(frame_out["prev_uv"], np.float32) if "prev_uv" in frame_out else None,
        "gPrevVisibility": np.ascontiguousarray(history["visibility"], wire.VisibilityInfo) if history.get("visibility") is not None else None,
        "gPrevDepth": np.ascontiguousarray(history["depth"], wire.DepthInfo) if history.get("depth") is not None else None,
        "gPrevAccumColor": np.ascontiguousarray(history["accum_color"], np.float32),
        "gPrevAccumMoments": np.ascontiguousarray(history["accum_moments"], np.float32),
        "gInstanceIndexMap": np.ascontiguousarray(instance_index_map, np.uint32) if instance_index_map is not None else None,
        "gAccumColor": out_c,
        "gAccumMoments": out_m,
    }
    for k, a in arrays.items():
        setattr(d, k, a.ctypes.data if a is not None else None)
    d.instance_count = arrays["gInstanceIndexMap"].shape[0] if instance_index_map is not None else 0
    keep.append(arrays)
    return d, out_c, out_m


class TemporalAccumulation:
    """The accumulation state Denoiser keeps between frames (Denoiser.cpp:66-77,176-213): accumulated colour (rgb =
    mean, a = sample count) and luminance moments, plus last frame's visibility / depth for the reprojection tests."""

    def __init__(self, bdpt, reprojection=True, demodulate_albedo=False, history_limit=0.0):
        self._bdpt = bdpt
        self.reprojection = reprojection
        self.demodulate_albedo = demodulate_albedo
        self.history_limit = history_limit
        self.history = None

    def reset(self):  # Denoiser::reset_accumulation
        self.history = None

    def __call__(self, frame_out, views, instance_index_map=None):
        rad = _rgba(frame_out["radiance"], "radiance")
        H, W = rad.shape[0], rad.shape[1]
        history = self.history
        if history is None:  # first frame: nothing accumulated yet (sample count 0 everywhere)
            vis = np.zeros((H, W), wire.VisibilityInfo)
            vis["instance_primitive_index"] = wire.MISS
            history = {"accum_color": np.zeros((H, W, 4), np.float32), "accum_moments": np.zeros((H, W, 2), np.float32), "visibility": vis, "depth": np.zeros((H, W), wire.DepthInfo)}
        keep = []
        d, out_c, out_m = accumulate_desc(frame_out, history, views, self.reprojection, self.demodulate_albedo, self.history_limit, instance_index_map, keep)
        self._bdpt._check(_lib.lib().sthip_accumulate(self._bdpt._h, C.byref(d)), "sthip_accumulate")
        self.history = {"accum_color": out_c, "accum_moments": out_m, "visibility": frame_out.get("visibility"), "depth": frame_out.get("depth")}
        return out_c, out_m


class Tonemapper:
    """The tone-map state BDPT keeps (BDPT.cpp:44-54,190-196,304-309): mode, exposure, gamma correction."""

    def __init__(self, bdpt, mode="Raw", exposure=0.0, gamma_correction=True, exposure_alpha=0.0):
        self._bdpt = bdpt
        self.mode = mode
        self.exposure = float(exposure)
        self.gamma_correction = bool(gamma_correction)
        self.exposure_alpha = float(exposure_alpha)  # gExposureAlpha: blend the maxima with the previous frame's (BDPT.cpp:51,192)
        self.state = np.zeros(6, np.float32)  # what the reference keeps in mTonemapMax from frame to frame

    def __call__(self, radiance, albedo=None, modulate_albedo=False, return_max=False):
        img = _rgba(radiance, "radiance")
        alb = _rgba(albedo, "albedo") if albedo is not None else None
        if alb is not None and alb.shape != img.shape:
            raise ValueError("albedo and radiance extents differ")
        out = np.empty_like(img)
        mx = np.zeros(4, np.float32)
        d = wire.TonemapDesc(
            img.shape[1],
            img.shape[0],
            _mode(wire.TONEMAP, self.mode),
            int(bool(modulate_albedo)),
            int(self.gamma_correction),
            self.exposure,
            0,
            self.exposure_alpha,
            img.ctypes.data,
            alb.ctypes.data if alb is not None else None,
            out.ctypes.data,
            mx.ctypes.data,
            self.state.ctypes.data,
        )
        self._bdpt._check(_lib.lib().sthip_tonemap(self._bdpt._h, C.byref(d)), "sthip_tonemap")
        return (out, mx) if return_max else out

    def device(self, width, height, input_ptr, albedo_ptr, output_ptr, modulate_albedo=False):
        """Device-pointer form (RGBA32F buffers in HBM); enqueues on the context's stream and returns."""
        d = wire.TonemapDesc(
            width, height, _mode(wire.TONEMAP, self.mode), int(bool(modulate_albedo)), int(self.gamma_correction), self.exposure, 1, 0.0, input_ptr, albedo_ptr, output_ptr, None, None
        )
        self._bdpt._check(_lib.lib().sthip_tonemap(self._bdpt._h, C.byref(d)), "sthip_tonemap")


class ImageComparer:
    """ImageComparer node: error of image1 against image2. `value` is what the GUI prints (ImageComparer.cpp:86-89)."""

    def __init__(self, bdpt, mode="SMAPE", quantization=1024):
        self._bdpt = bdpt
        self.mode = mode
        self.quantization = int(quantization)

    def raw(self, image1, image2):
        a, b = _rgba(image1, "image1"), _rgba(image2, "image2")
        if a.shape != b.shape:
            raise ValueError("image extents differ")
        s, o = C.c_uint32(0), C.c_uint32(0)
        rc = _lib.lib().sthip_image_compare(
            self._bdpt._h, a.ctypes.data, b.ctypes.data, a.shape[1], a.shape[0], _mode(wire.COMPARE, self.mode), self.quantization, 0, C.byref(s), C.byref(o)
        )
        self._bdpt._check(rc, "sthip_image_compare")
        return s.value, bool(o.value)

    def value(self, image1, image2):
        s, overflow = self.raw(image1, image2)
        if overflow:
            return float("inf")
        v = s / float(self.quantization)
        return float(np.sqrt(v)) if _mode(wire.COMPARE, self.mode) == wire.COMPARE["MSE"] else v


def write_hdr(path, image):
    """Radiance .hdr of an (H, W, 4) float image, byte-compatible with the reference's stbi_write_hdr export."""
    img = _rgba(image, "image")
    rc = _lib.lib().sthip_write_hdr(str(path).encode(), img.shape[1], img.shape[0], img.ctypes.data)
    if rc != 0:
        raise _lib.StratumHipError("sthip_write_hdr(%s) failed (%d)" % (path, rc))
