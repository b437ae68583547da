"""Host-side scene packing: what Scene::update (src/Node/Scene.cpp:299-684) produces.

SceneBuilder collects meshes, materials and mesh instances and emits the arrays the
shaders bind as gSceneParams.* — PackedVertexData vertices, the byte index buffer,
InstanceData records, instance / inverse / motion transforms, the material byte buffer
and the light-instance list — in exactly the reference's layouts (include/sthip_wire.h).
"""
import ctypes as C

import math

import numpy as np

from . import wire

_IDENTITY = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], dtype=np.float32)


def translate(t):
    m = np.eye(4)
    m[:3, 3] = t
    return m


def scale(s):
    s = np.broadcast_to(np.asarray(s, dtype=np.float64), (3,))
    return np.diag([s[0], s[1], s[2], 1.0])


def rotate_y(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def rotate_x(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def transform_inverse(m32):
    """TransformData::inverse (transform.h:26-31). The reference takes Eigen's 4x4 inverse; pinned here — and
    identically in stratum_amd/host/stratum_hip.hpp — as the adjugate of the 3x3 block over its determinant,
    evaluated in double in this exact order, then 0 - (inv * t). An identity goes in, an identity comes out."""
    a = [[float(m32[i][j]) for j in range(4)] for i in range(3)]
    a00, a01, a02 = a[0][0], a[0][1], a[0][2]
    a10, a11, a12 = a[1][0], a[1][1], a[1][2]
    a20, a21, a22 = a[2][0], a[2][1], a[2][2]
    c00 = a11 * a22 - a12 * a21
    c01 = a12 * a20 - a10 * a22
    c02 = a10 * a21 - a11 * a20
    det = a00 * c00 + a01 * c01 + a02 * c02
    i = [
        [c00 / det, (a02 * a21 - a01 * a22) / det, (a01 * a12 - a02 * a11) / det],
        [c01 / det, (a00 * a22 - a02 * a20) / det, (a02 * a10 - a00 * a12) / det],
        [c02 / det, (a01 * a20 - a00 * a21) / det, (a00 * a11 - a01 * a10) / det],
    ]
    tx, ty, tz = a[0][3], a[1][3], a[2][3]
    out = np.zeros((3, 4), np.float32)
    for r in range(3):
        out[r, 0], out[r, 1], out[r, 2] = i[r][0], i[r][1], i[r][2]
        out[r, 3] = 0.0 - (i[r][0] * tx + i[r][1] * ty + i[r][2] * tz)
    return out


def tmul(a, b):
    """tmul (transform.h:88-104) in binary32, rows dotted left to right: a * [b; 0 0 0 1]."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    r = np.zeros((3, 4), np.float32)
    for i in range(3):
        for j in range(4):
            s = np.float32(a[i, 0] * b[0, j]) + np.float32(a[i, 1] * b[1, j])
            s = np.float32(s + np.float32(a[i, 2] * b[2, j]))
            if j == 3:
                s = np.float32(s + a[i, 3])
            r[i, j] = s
    return r


class SceneData:
    """Packed arrays + the descriptor handed to sthip_scene_upload / the oracle."""

    def __init__(self, vertices, indices, instances, xf, inv_xf, motion_xf, materials, lights, name=""):
        self.name = name
        self.vertices = vertices
        self.indices = indices
        self.instances = instances
        self.transforms = xf
        self.inverse_transforms = inv_xf
        self.motion_transforms = motion_xf
        self.materials = materials
        self.lights = lights
        self.images = []  # RGBA32F arrays (H, W, 4): Texture2D<float4> gImages[]
        self.images1 = []  # float32 arrays (H, W): Texture2D<float> gImage1s[] (alpha masks)
        self.distributions = np.zeros(0, np.float32)  # StructuredBuffer<float> gDistributions
        self.environment_address = 0xFFFFFFFF  # SceneData::mEnvironmentMaterialAddress, Scene.cpp:631-640
        self.volumes = []  # uint8 arrays: ByteAddressBuffer gVolumes[] (NanoVDB float grids)

    @property
    def light_count(self):
        return int(self.lights.shape[0])

    @property
    def triangle_count(self):
        return int(((self.instances["packed"][:, 1] >> 12) & 0xFFFF).sum())

    @property
    def scene_flags(self):  # BDPT.cpp:486-496
        f = wire.BDPT_FLAG_HAS_EMISSIVES if self.light_count else 0
        if self.environment_address != 0xFFFFFFFF:
            f |= wire.BDPT_FLAG_HAS_ENVIRONMENT
        if self.volumes:  # BDPT.cpp:441,459-460,497-500: any Medium component in the scene
            f |= wire.BDPT_FLAG_HAS_MEDIA
        return f

    def set_instance_transform(self, index, transform):
        """Move instance `index` to a new object-to-world transform (4x4 or 3x4): the arrays Scene::update refreshes every
        frame (Scene.cpp:398-427) — transform, its inverse, and the motion transform prev_object_to_world x world_to_object
        (make_instance_motion_transform, scene.h:49)."""
        m = np.asarray(transform, dtype=np.float64)
        m32 = m[:3, :].astype(np.float32)
        prev = self.transforms["m"][index].copy()
        self.transforms["m"][index] = m32
        self.inverse_transforms["m"][index] = transform_inverse(m32)
        self.motion_transforms["m"][index] = tmul(prev, self.inverse_transforms["m"][index])

    def view_medium_instances(self, view_transforms):
        """gViewMediumInstances (BDPT.cpp:456-466): per view the volume instance whose grid world box contains the camera."""
        out = np.full(view_transforms.shape[0], wire.INVALID_INSTANCE, np.uint32)
        kinds = self.instances["packed"][:, 0] & 0xF
        for i in np.nonzero(kinds == wire.INSTANCE_TYPE_VOLUME)[0]:
            grid = self.volumes[int(self.instances["packed"][i, 2])]
            box = grid[560:608].view(np.float64)  # GridData::mWorldBBox (PNANOVDB_GRID_OFF_WORLD_BBOX)
            inv = self.inverse_transforms["m"][i].astype(np.float64)
            for v in range(view_transforms.shape[0]):
                p = inv[:, :3] @ view_transforms["m"][v][:, 3].astype(np.float64) + inv[:, 3]
                if np.all(p >= box[:3]) and np.all(p <= box[3:]):
                    out[v] = i
        return out

    def desc(self):
        d = wire.SceneDesc()
        d.gVertices = wire.ptr(self.vertices)
        d.vertex_count = self.vertices.shape[0]
        d.gIndices = wire.ptr(self.indices)
        d.indices_bytes = self.indices.nbytes
        d.gInstances = wire.ptr(self.instances)
        d.instance_count = self.instances.shape[0]
        d.gInstanceTransforms = wire.ptr(self.transforms)
        d.gInstanceInverseTransforms = wire.ptr(self.inverse_transforms)
        d.gInstanceMotionTransforms = wire.ptr(self.motion_transforms)
        d.gMaterialData = wire.ptr(self.materials)
        d.material_bytes = self.materials.nbytes
        d.gLightInstances = wire.ptr(self.lights) if self.light_count else None
        d.light_count = self.light_count
        if self.images:
            self._image_descs = (wire.ImageDesc * len(self.images))()
            for i, im in enumerate(self.images):
                self._image_descs[i].pixels = wire.ptr(im)
                self._image_descs[i].width = im.shape[1]
                self._image_descs[i].height = im.shape[0]
            d.gImages = C.cast(self._image_descs, C.c_void_p)
            d.image_count = len(self.images)
        if self.images1:
            self._image1_descs = (wire.ImageDesc * len(self.images1))()
            for i, im in enumerate(self.images1):
                self._image1_descs[i].pixels = wire.ptr(im)
                self._image1_descs[i].width = im.shape[1]
                self._image1_descs[i].height = im.shape[0]
            d.gImage1s = C.cast(self._image1_descs, C.c_void_p)
            d.image1_count = len(self.images1)
        if self.distributions.size:
            d.gDistributions = wire.ptr(self.distributions)
            d.distribution_count = self.distributions.size
        if self.volumes:
            self._volume_descs = (wire.VolumeDesc * len(self.volumes))()
            for i, g in enumerate(self.volumes):
                self._volume_descs[i].data = wire.ptr(g)
                self._volume_descs[i].bytes = g.nbytes
            d.gVolumes = C.cast(self._volume_descs, C.c_void_p)
            d.volume_count = len(self.volumes)
        return d


class SceneBuilder:
    def __init__(self, name=""):
        self.name = name
        self._verts = []  # list of PackedVertexData arrays
        self._index_chunks = []  # list of bytes
        self._index_bytes = 0
        self._vertex_count = 0
        self._meshes = []  # (first_vertex, indices_byte_offset, prim_count, stride)
        self._materials = []  # MaterialRecord entries
        self._instances = []  # (mesh, material index, 4x4 transform)
        self._images = []  # RGBA32F (H, W, 4)
        self._images1 = []  # float32 (H, W): one-channel images (alpha masks)
        self._spheres = []  # (material index, 4x4 node transform, radius): SpherePrimitive, Scene.hpp:34-37
        self._environment = None  # (value rgb, image handle or None): Environment, environment.h

    # -- Material::store, Material.hpp:32-38; conventions of load_mitsuba.cpp:330-343,454-489 --
    def add_material(
        self,
        base_color,
        emission=0.0,
        metallic=0.0,
        roughness=0.0,
        anisotropic=0.0,
        subsurface=0.0,
        clearcoat=0.0,
        clearcoat_gloss=0.0,
        transmission=0.0,
        eta=1.5,
    ):
        rec = np.zeros((), dtype=wire.MaterialRecord)
        rec["values"]["value"][0] = [base_color[0], base_color[1], base_color[2], emission]
        rec["values"]["value"][1] = [metallic, roughness, anisotropic, subsurface]
        rec["values"]["value"][2] = [clearcoat, clearcoat_gloss, transmission, eta]
        rec["values"]["image_index"][:] = 0xFFFFFFFF  # MaterialResources::get_index of a null view (image_value.h:40-41)
        rec["alpha_mask_index"] = 0xFFFFFFFF
        rec["bump_index"] = 0xFFFFFFFF
        rec["bump_strength"] = 1.0
        self._materials.append(rec)
        return len(self._materials) - 1  # a handle; byte addresses are assigned in build() in order of first use

    def add_image(self, rgba):
        """Registers a Texture2D<float4> (float32 array H x W x 4, row 0 first); returns its index in gImages."""
        im = np.ascontiguousarray(rgba, dtype=np.float32)
        assert im.ndim == 3 and im.shape[2] == 4
        self._images.append(im)
        return len(self._images) - 1

    def add_image1(self, gray):
        """Registers a Texture2D<float> (float32 array H x W, row 0 first); returns its handle (gImage1s index by first use)."""
        im = np.ascontiguousarray(gray, dtype=np.float32)
        assert im.ndim == 2
        self._images1.append(im)
        return len(self._images1) - 1

    def set_material_alpha_mask(self, material, image1):
        """Material::alpha_mask (Material.hpp:14,35): coverage image tested by eAlphaTest."""
        self._materials[material]["alpha_mask_index"] = image1

    def set_material_images(self, material, base_color_image=None, params_image=None, lobes_image=None, bump_image=None, bump_strength=1.0):
        """Binds images to the three ImageValue4 of a material (disney_data.h:1-20) and/or a normal map."""
        rec = self._materials[material]
        for k, img in enumerate((base_color_image, params_image, lobes_image)):
            if img is not None:
                rec["values"]["image_index"][k] = img
        if bump_image is not None:
            rec["bump_index"] = bump_image
            rec["bump_strength"] = bump_strength

    def add_emitter(self, radiance):
        """Mitsuba area emitter: base_color = L / lum(L), emission = lum(L), eta = 0 (load_mitsuba.cpp:480-489)."""
        L = np.asarray(radiance, dtype=np.float32)
        lum = np.float32(np.dot(L, np.array([0.2126, 0.7152, 0.0722], dtype=np.float32)))
        return self.add_material(L / lum, emission=float(lum), eta=0.0)

    # -- copy_vertices + concatenation, copy_vertices.hlsl:29-37, Scene.cpp:643-658 --
    def add_mesh(self, positions, normals, uvs, triangles, index_stride=4):
        positions = np.asarray(positions, dtype=np.float32).reshape(-1, 3)
        n = positions.shape[0]
        normals = np.asarray(normals, dtype=np.float32).reshape(-1, 3) if normals is not None else np.zeros((n, 3), np.float32)
        uvs = np.asarray(uvs, dtype=np.float32).reshape(-1, 2) if uvs is not None else np.zeros((n, 2), np.float32)
        triangles = np.asarray(triangles).reshape(-1, 3)
        assert triangles.shape[0] <= 0xFFFF, "16-bit primitive index (scene.h:23-24,37): chunk the mesh"
        assert triangles.max() < n
        v = np.zeros(n, dtype=wire.PackedVertexData)
        v["position"], v["normal"], v["u"], v["v"] = positions, normals, uvs[:, 0], uvs[:, 1]
        if index_stride == 2:
            assert n <= 0x10000
            ib = triangles.astype("<u2").tobytes()
        else:
            ib = triangles.astype("<u4").tobytes()
        mesh = (self._vertex_count, self._index_bytes, triangles.shape[0], index_stride)
        self._verts.append(v)
        self._vertex_count += n
        pad = (-len(ib)) % 4  # Scene.cpp:507 align_up(size, 4)
        self._index_chunks.append(ib + b"\0" * pad)
        self._index_bytes += len(ib) + pad
        self._meshes.append(mesh)
        return len(self._meshes) - 1

    def add_instance(self, mesh, material, transform=None):
        m = np.eye(4) if transform is None else np.asarray(transform, dtype=np.float64)
        if m.shape == (3, 4):
            m = np.vstack([m, [0, 0, 0, 1]])
        self._instances.append((mesh, material, m))
        return len(self._instances) - 1

    def add_sphere(self, material, radius, transform=None):
        """SpherePrimitive: a sphere of `radius` at the node's origin (Scene.cpp:511-553)."""
        m = np.eye(4) if transform is None else np.asarray(transform, dtype=np.float64)
        if m.shape == (3, 4):
            m = np.vstack([m, [0, 0, 0, 1]])
        self._spheres.append((material, m, float(radius)))
        return len(self._instances) + len(self._spheres) - 1

    def add_volume(self, grid_bytes):
        """A NanoVDB float grid (the bytes of a nanovdb::GridHandle) for Medium::density_buffer / albedo_buffer."""
        if not hasattr(self, "_volumes"):
            self._volumes, self._media, self._medium_instances = [], [], []
        self._volumes.append(np.ascontiguousarray(np.frombuffer(bytes(grid_bytes), dtype=np.uint8) if not isinstance(grid_bytes, np.ndarray) else grid_bytes.astype(np.uint8)))
        return len(self._volumes) - 1

    def add_medium(self, density_volume, density_scale=(1, 1, 1), albedo_scale=(1, 1, 1), anisotropy=0.0, attenuation_unit=1.0, albedo_volume=None, transform=None):
        """A Medium component (Material.hpp:72-87) on a node: one volume instance of the scene (Scene.cpp:556-590)."""
        m = np.eye(4) if transform is None else np.asarray(transform, dtype=np.float64)
        if m.shape == (3, 4):
            m = np.vstack([m, [0, 0, 0, 1]])
        rec = (tuple(float(x) for x in density_scale), float(anisotropy), tuple(float(x) for x in albedo_scale), float(attenuation_unit), int(density_volume), albedo_volume)
        self._media.append(rec)
        self._medium_instances.append((len(self._media) - 1, m))
        return len(self._medium_instances) - 1

    def set_environment(self, value, image=None):
        """Environment component: constant radiance `value`, or `value` times a lat-long image (environment.h)."""
        self._environment = (np.asarray(value, dtype=np.float32).reshape(3), image)

    def build(self):
        media = getattr(self, "_medium_instances", [])
        n = len(self._instances) + len(self._spheres) + len(media)
        inst = np.zeros(n, dtype=wire.InstanceData)
        xf = np.zeros(n, dtype=wire.TransformData)
        inv = np.zeros(n, dtype=wire.TransformData)
        mot = np.zeros(n, dtype=wire.TransformData)
        # process_material, Scene.cpp:387-396: a material is appended to the byte buffer the first time an
        # instance uses it; its address is the byte offset at that moment
        # ... and an image gets its gImages index the first time a stored material refers to it
        # (MaterialResources::get_index, image_value.h:34-48)
        address_of, used, image_index_of, image_order = {}, [], {}, []
        image1_index_of, image1_order = {}, []

        def image1_index(handle):
            handle = int(handle)
            if handle == 0xFFFFFFFF:
                return 0xFFFFFFFF
            if handle not in image1_index_of:
                image1_index_of[handle] = len(image1_order)
                image1_order.append(handle)
            return image1_index_of[handle]

        def image_index(handle):
            handle = int(handle)
            if handle == 0xFFFFFFFF:
                return 0xFFFFFFFF
            if handle not in image_index_of:
                image_index_of[handle] = len(image_order)
                image_order.append(handle)
            return image_index_of[handle]

        for mat in [m for _, m, _ in self._instances] + [m for m, _, _ in self._spheres]:
            if mat not in address_of:
                address_of[mat] = len(used) * wire.MaterialRecord.itemsize
                rec = self._materials[mat].copy()
                for k in range(3):  # Material::store's order, Material.hpp:32-37: the three values, the alpha mask, the bump map
                    rec["values"]["image_index"][k] = image_index(rec["values"]["image_index"][k])
                rec["alpha_mask_index"] = image1_index(rec["alpha_mask_index"])
                rec["bump_index"] = image_index(rec["bump_index"])
                used.append(rec)
        mats = np.array(used, dtype=wire.MaterialRecord) if used else np.zeros(0, wire.MaterialRecord)
        lights = []
        for i, (mesh, mat, m) in enumerate(self._instances):
            fv, ibo, pc, stride = self._meshes[mesh]
            # make_instance_triangles, scene.h:51-61
            p0 = 0 | (address_of[mat] << 4)  # INSTANCE_TYPE_TRIANGLES
            p1 = 0xFFF | (pc << 12) | (stride << 28)
            emission = float(self._materials[mat]["values"]["value"][0][3])
            if emission > 0:  # process_instance, Scene.cpp:403-409
                p1 = (p1 & ~0xFFF) | (len(lights) & 0xFFF)
                lights.append(i)
            inst["packed"][i] = [p0 & 0xFFFFFFFF, p1 & 0xFFFFFFFF, fv, ibo]
            m32 = m[:3, :].astype(np.float32)
            xf["m"][i] = m32
            inv["m"][i] = transform_inverse(m32)
            # make_instance_motion_transform(inv, prevObjectToWorld), scene.h:49; static scene: prev = current
            mot["m"][i] = tmul(m32, inv["m"][i])
        for k, (mat, m, radius) in enumerate(self._spheres):  # Scene.cpp:511-553, after every mesh instance
            i = len(self._instances) + k
            m32 = m[:3, :].astype(np.float32)
            # the reference scales the radius by the DETERMINANT of the node's 3x3 block (:520) and keeps only the
            # translation of the node transform (:521)
            b = m32[:, :3]
            det = np.float32(
                b[0, 0] * (b[1, 1] * b[2, 2] - b[1, 2] * b[2, 1]) - b[0, 1] * (b[1, 0] * b[2, 2] - b[1, 2] * b[2, 0]) + b[0, 2] * (b[1, 0] * b[2, 1] - b[1, 1] * b[2, 0])
            )
            r = np.float32(np.float32(radius) * det)
            t32 = np.zeros((3, 4), np.float32)
            t32[0, 0] = t32[1, 1] = t32[2, 2] = 1
            t32[:, 3] = m32[:, 3]
            p0 = wire.INSTANCE_TYPE_SPHERE | (address_of[mat] << 4)  # make_instance_sphere, scene.h:62-70
            p1 = 0xFFF
            emission = float(self._materials[mat]["values"]["value"][0][3])
            if emission * (4 * np.pi * float(r) * float(r)) > 0:
                p1 = (p1 & ~0xFFF) | (len(lights) & 0xFFF)
                lights.append(i)
            inst["packed"][i] = [p0 & 0xFFFFFFFF, p1 & 0xFFFFFFFF, int(np.array([r], np.float32).view(np.uint32)[0]), 0]
            xf["m"][i] = t32
            inv["m"][i] = transform_inverse(t32)
            mot["m"][i] = tmul(t32, inv["m"][i])
        vertices = np.concatenate(self._verts) if self._verts else np.zeros(0, wire.PackedVertexData)
        indices = np.frombuffer(b"".join(self._index_chunks), dtype=np.uint8).copy()
        mat_bytes = np.ascontiguousarray(mats).view(np.uint8).reshape(-1)
        # media, Scene.cpp:556-590, after every mesh and sphere: Medium::store appends 40 bytes (Material.hpp:80-87), the
        # volumes get their gVolumes index the first time a stored medium refers to them (density, then albedo)
        volume_index_of, volume_order = {}, []

        def volume_index(handle):
            if handle is None:
                return 0xFFFFFFFF
            if handle not in volume_index_of:
                volume_index_of[handle] = len(volume_order)
                volume_order.append(handle)
            return volume_index_of[handle]

        for k, (medium, m) in enumerate(media):
            i = len(self._instances) + len(self._spheres) + k
            density_scale, anisotropy, albedo_scale, attenuation_unit, density_volume, albedo_volume = self._media[medium]
            address = mat_bytes.size
            rec = np.zeros(10, np.uint32)
            rec[:3] = np.array(density_scale, np.float32).view(np.uint32)
            rec[3] = np.array([anisotropy], np.float32).view(np.uint32)[0]
            rec[4:7] = np.array(albedo_scale, np.float32).view(np.uint32)
            rec[7] = np.array([attenuation_unit], np.float32).view(np.uint32)[0]
            rec[8] = volume_index(density_volume)
            rec[9] = volume_index(albedo_volume)
            mat_bytes = np.concatenate([mat_bytes, rec.view(np.uint8)])
            m32 = m[:3, :].astype(np.float32)
            inst["packed"][i] = [(wire.INSTANCE_TYPE_VOLUME | (address << 4)) & 0xFFFFFFFF, 0xFFF, int(rec[8]), 0]  # make_instance_volume, scene.h:71-79
            xf["m"][i] = m32
            inv["m"][i] = transform_inverse(m32)
            mot["m"][i] = tmul(m32, inv["m"][i])
        # environment material, Scene.cpp:631-640: appended after every instance material, only if its value is not zero
        env_address, dist = 0xFFFFFFFF, np.zeros(0, np.float32)
        if self._environment is not None and np.any(self._environment[0] != 0):
            value, image = self._environment
            env_address = mat_bytes.size
            rec = np.zeros(4, np.uint32)
            rec[:3] = value.view(np.uint32)
            rec[3] = image_index(0xFFFFFFFF if image is None else image)
            parts = [rec]
            if image is not None:  # Environment::store, environment.h:17-22: offsets into gDistributions in get_index order
                tables = build_distributions(self._images[int(image)])  # marginal_pdf, row_pdf, marginal_cdf, row_cdf
                offs, off = [], 0
                for t in tables:
                    offs.append(off)
                    off += t.size
                parts.append(np.array(offs, np.uint32))
                dist = np.concatenate(tables).astype(np.float32)
            mat_bytes = np.concatenate([mat_bytes, np.concatenate(parts).view(np.uint8)])
        sd = SceneData(
            np.ascontiguousarray(vertices),
            indices,
            inst,
            xf,
            inv,
            mot,
            np.ascontiguousarray(mat_bytes),
            np.array(lights, dtype=np.uint32),
            name=self.name,
        )
        sd.builder = self  # the inputs the arrays were packed from (dump_description)
        sd.images = [self._images[h] for h in image_order]
        sd.images1 = [self._images1[h] for h in image1_order]
        sd.distributions = dist
        sd.environment_address = env_address
        sd.volumes = [self._volumes[h] for h in volume_order]
        return sd


def build_distributions(image):
    """dist2.h:80-154 build_distributions for a lat-long RGBA32F image (H, W, 4): returns (marginal_pdf[H],
    row_pdf[H*W], marginal_cdf[H+1], row_cdf[H*(W+1)]). Accumulation order and the float/double mix follow the
    reference's loops: f(x, y) is a double (float luminance times double sin), every running sum is stored as float."""
    img = np.asarray(image, dtype=np.float32)
    H, W = img.shape[0], img.shape[1]
    lum = (img[..., 0] * np.float32(0.2126) + img[..., 1] * np.float32(0.7152)) + img[..., 2] * np.float32(0.0722)
    inv_h = np.float64(np.float32(1.0) / np.float32(H))
    y = np.arange(H, dtype=np.float32) + np.float32(0.5)
    # libm's sin (math.sin), the function the C++ hosts call: numpy's vectorised sin may differ in the last bit
    sin_row = np.array([math.sin(float(v)) for v in (np.pi * y.astype(np.float64) * inv_h)], np.float64)
    f = lum.astype(np.float64) * sin_row[:, None]
    cdf_rows = np.zeros((H, W + 1), np.float32)
    for x in range(W):
        cdf_rows[:, x + 1] = (cdf_rows[:, x].astype(np.float64) + f[:, x]).astype(np.float32)
    integral = cdf_rows[:, W].copy()
    pdf_rows = np.zeros((H, W), np.float32)
    pos = integral > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        cdf_rows[pos, :W] = cdf_rows[pos, :W] / integral[pos, None]
        pdf_rows[pos] = (f[pos] / integral[pos, None].astype(np.float64)).astype(np.float32)
    if np.any(~pos):
        pdf_rows[~pos] = np.float32(1) / np.float32(W)
        cdf_rows[~pos, :W] = np.arange(W, dtype=np.float32) / np.float32(W)
        cdf_rows[~pos, W] = 1
    cdf_marg = np.zeros(H + 1, np.float32)
    for yy in range(H):
        cdf_marg[yy + 1] = cdf_marg[yy] + cdf_rows[yy, W]
    total = cdf_marg[H]
    pdf_marg = np.zeros(H, np.float32)
    if total > 0:
        weights = cdf_rows[:, W].copy()
        cdf_marg[:H] = cdf_marg[:H] / total
        cdf_marg[H] = 1
        pdf_marg[:] = weights / total
    else:
        pdf_marg[:] = np.float32(1) / np.float32(H)
        cdf_marg[:H] = np.arange(H, dtype=np.float32) / np.float32(H)
        cdf_marg[H] = 1
    cdf_rows[:, W] = 1
    return pdf_marg, pdf_rows.reshape(-1), cdf_marg, cdf_rows.reshape(-1)


def dump_description(path, scene, frame):
    """Writes the INPUT of a SceneBuilder (materials, meshes, instances with their node transforms), the view,
    and the packed OUTPUT arrays to one little-endian binary file. tests/cpp/host_test.cpp rebuilds the scene
    through the C++ Node-graph API of stratum_amd/host/stratum_hip.hpp from the input part and must reproduce the
    output part byte for byte."""
    import struct

    builder = scene.builder
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(builder._images)))
        for im in builder._images:
            f.write(struct.pack("<II", im.shape[1], im.shape[0]))
            f.write(im.tobytes())
        f.write(struct.pack("<I", len(builder._images1)))
        for im in builder._images1:
            f.write(struct.pack("<II", im.shape[1], im.shape[0]))
            f.write(im.tobytes())
        mats = np.array(builder._materials, dtype=wire.MaterialRecord)
        f.write(struct.pack("<I", mats.shape[0]))
        f.write(mats.tobytes())
        f.write(struct.pack("<I", len(builder._meshes)))
        for k, (fv, ibo, pc, stride) in enumerate(builder._meshes):
            v = builder._verts[k]
            f.write(struct.pack("<III", v.shape[0], pc, stride))
            f.write(np.ascontiguousarray(v["position"]).tobytes())
            f.write(np.ascontiguousarray(v["normal"]).tobytes())
            f.write(np.ascontiguousarray(np.stack([v["u"], v["v"]], 1)).tobytes())
            raw = builder._index_chunks[k][: pc * 3 * stride]
            idx = np.frombuffer(raw, dtype="<u2" if stride == 2 else "<u4").astype("<u4")
            f.write(idx.tobytes())
        f.write(struct.pack("<I", len(builder._instances)))
        for mesh, mat, m in builder._instances:
            f.write(struct.pack("<II", mesh, mat))
            f.write(m[:3, :].astype("<f4").tobytes())
        f.write(struct.pack("<I", len(builder._spheres)))
        for mat, m, radius in builder._spheres:
            f.write(struct.pack("<If", mat, radius))
            f.write(m[:3, :].astype("<f4").tobytes())
        if builder._environment is None:
            f.write(struct.pack("<I", 0))
        else:
            value, image = builder._environment
            f.write(struct.pack("<I", 1 if image is None else 2))
            f.write(value.astype("<f4").tobytes())
            f.write(struct.pack("<I", 0xFFFFFFFF if image is None else int(image)))
        volumes, media = getattr(builder, "_volumes", []), getattr(builder, "_medium_instances", [])
        f.write(struct.pack("<I", len(volumes)))
        for g in volumes:
            f.write(struct.pack("<Q", g.nbytes))
            f.write(g.tobytes())
        f.write(struct.pack("<I", len(media)))
        for medium, m in media:
            density_scale, anisotropy, albedo_scale, attenuation_unit, density_volume, albedo_volume = builder._media[medium]
            f.write(struct.pack("<3ff3ffII", *density_scale, anisotropy, *albedo_scale, attenuation_unit, density_volume, 0xFFFFFFFF if albedo_volume is None else albedo_volume))
            f.write(m[:3, :].astype("<f4").tobytes())
        f.write(frame.views.tobytes())
        f.write(frame.view_transforms.tobytes())
        f.write(struct.pack("<II", frame.width, frame.height))
        for a in (scene.vertices, scene.indices, scene.instances, scene.transforms, scene.inverse_transforms, scene.motion_transforms, scene.materials, scene.lights, scene.distributions):
            b = np.ascontiguousarray(a).tobytes()
            f.write(struct.pack("<Q", len(b)))
            f.write(b)
