"""Camera / view construction as the reference host does it.

make_perspective: src/Shaders/transform.h:159-168; Camera::view: src/Node/Scene.hpp:19-28;
the aspect passed by FlyCamera is height/width (src/Node/FlyCamera.cpp:29-32);
the default camera looks down -Z with near_plane = -1/1024 (src/main.cpp:86).
"""
import numpy as np

from . import wire


def make_perspective(fovy, aspect, offset=(0.0, 0.0), znear=-1.0 / 1024.0):
    p = np.zeros((), dtype=wire.ProjectionData)
    sy = np.float32(1.0) / np.float32(np.tan(np.float32(fovy) / np.float32(2)))
    p["scale"] = [np.float32(aspect) * sy, sy]
    p["offset"] = offset
    p["near_plane"] = znear
    p["far_plane"] = 0
    p["vertical_fov"] = fovy
    return p


def _back_project(p, v):
    near = np.float32(p["near_plane"])
    if p["vertical_fov"] < 0:
        return np.array([(v[0] - p["offset"][0]) / p["scale"][0], (v[1] - p["offset"][1]) / p["scale"][1], near], np.float32)
    s = np.sign(near)
    return np.array(
        [near * (v[0] * s - p["offset"][0]) / p["scale"][0], near * (v[1] * s - p["offset"][1]) / p["scale"][1], near], np.float32
    )


def make_view(width, height, fovy):
    """ViewData for a full-frame camera (Camera::view with mImageRect = the whole image)."""
    v = np.zeros(1, dtype=wire.ViewData)
    proj = make_perspective(fovy, height / float(width))
    ext = _back_project(proj, (1.0, 1.0))[:2] - _back_project(proj, (-1.0, -1.0))[:2]
    if proj["vertical_fov"] >= 0:
        ext = ext / np.float32(proj["near_plane"])
    proj["sensor_area"] = abs(ext[0] * ext[1])
    v["projection"][0] = proj
    v["image_min"][0] = [0, 0]
    v["image_max"][0] = [width, height]
    return v


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """Camera-to-world TransformData (node_to_world of the camera node); camera looks down its -Z."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64))
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    m = np.zeros((3, 4))
    m[:, 0], m[:, 1], m[:, 2], m[:, 3] = r, u, -f, eye
    t = np.zeros(1, dtype=wire.TransformData)
    t["m"][0] = m.astype(np.float32)
    return t


def inverse_transform(t):
    from .scene import transform_inverse

    out = np.zeros_like(t)
    for i in range(t.shape[0]):
        out["m"][i] = transform_inverse(t["m"][i])  # views[i].second.inverse(), BDPT.cpp:452
    return out


class Frame:
    """gFrameParams view arrays for one camera. `prev` = the Frame of the previous render call: its views and
    inverse view transforms become gPrevViews / gPrevInverseViewTransforms (BDPT.cpp:453-467), which is what the
    prev-uv / prev_z outputs and the temporal reprojection use; without it the camera is static."""

    def __init__(self, width, height, fovy, eye, target, up=(0.0, 1.0, 0.0), prev=None):
        self.width, self.height = width, height
        self.views = make_view(width, height, fovy)
        self.view_transforms = look_at(eye, target, up)
        self.inverse_view_transforms = inverse_transform(self.view_transforms)
        self.prev = prev

    @classmethod
    def stereo(cls, width, height, fovy, eye, target, eye_separation=0.1, up=(0.0, 1.0, 0.0)):
        """Two views side by side in one image (what the XR node hands BDPT::render: one ViewData per eye with its own
        image rectangle, BDPT.cpp:444-467): left half = left eye, right half = right eye."""
        half = width // 2
        f = np.asarray(target, np.float64) - np.asarray(eye, np.float64)
        r = np.cross(f / np.linalg.norm(f), np.asarray(up, np.float64))
        r /= np.linalg.norm(r)
        frames = [cls(half, height, fovy, tuple(np.asarray(eye) + s * 0.5 * eye_separation * r), target, up) for s in (-1.0, 1.0)]
        self = cls.__new__(cls)
        self.width, self.height, self.prev = width, height, None
        self.views = np.concatenate([fr.views for fr in frames])
        self.views["image_min"][1] = [half, 0]
        self.views["image_max"][1] = [2 * half, height]
        self.view_transforms = np.concatenate([fr.view_transforms for fr in frames])
        self.inverse_view_transforms = np.concatenate([fr.inverse_view_transforms for fr in frames])
        return self

    def desc(self):
        d = wire.FrameDesc()
        d.gViews = wire.ptr(self.views)
        d.gViewTransforms = wire.ptr(self.view_transforms)
        d.gInverseViewTransforms = wire.ptr(self.inverse_view_transforms)
        d.gPrevViews = wire.ptr(self.prev.views) if self.prev is not None else None
        d.gPrevInverseViewTransforms = wire.ptr(self.prev.inverse_view_transforms) if self.prev is not None else None
        d.view_count = self.views.shape[0]
        self._view_media = getattr(self, "view_medium_instances", None)
        d.gViewMediumInstances = wire.ptr(self._view_media) if self._view_media is not None else None
        return d
