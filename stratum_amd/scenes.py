"""Deterministic procedural scenes of BASELINE.json's configs (no assets, no textures).

cornell_box  — 32 triangles, diffuse walls + one emissive quad (configs[0], configs[1])
atrium       — ~1M-triangle "Sponza-class" hall of displaced grids, chunked into meshes of
               <= 65535 triangles because instance/primitive ids are 16-bit (scene.h:23-24,37)
forest       — ~10M triangles as ~1000 instances of a few ~10K-triangle tree meshes (configs[4])

Every generator returns (SceneData, camera dict) where camera = {eye, target, fovy}.
"""
import numpy as np

from .scene import SceneBuilder, rotate_x, rotate_y, scale, translate

MAX_TRIS = 0xFFFF


# ---------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------
def _quad(p0, p1, p2, p3, n):
    pos = np.array([p0, p1, p2, p3], dtype=np.float32)
    nrm = np.tile(np.asarray(n, dtype=np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3]])
    return pos, nrm, uv, tri


def _merge(parts):
    pos, nrm, uv, tri, base = [], [], [], [], 0
    for p, n, u, t in parts:
        pos.append(p)
        nrm.append(n)
        uv.append(u)
        tri.append(np.asarray(t) + base)
        base += p.shape[0]
    return np.concatenate(pos), np.concatenate(nrm), np.concatenate(uv), np.concatenate(tri)


def grid_surface(fn, nu, nv, flip=False):
    """Tessellate P(u,v), u,v in [0,1], into nu x nv quads. Normals from central differences."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    P = fn(U, V)
    e = 1e-4
    dU = fn(U + e, V) - fn(U - e, V)
    dV = fn(U, V + e) - fn(U, V - e)
    N = np.cross(dU, dV)
    N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-20)
    if flip:
        N = -N
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tri = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2).reshape(-1, 3)
    if flip:
        tri = tri[:, ::-1]
    uv = np.stack([U, V], -1)
    return (P.reshape(-1, 3).astype(np.float32), N.reshape(-1, 3).astype(np.float32), uv.reshape(-1, 2).astype(np.float32), tri)


def add_chunked(b, part, material, transform=None, index_stride=4, max_tris=MAX_TRIS):
    """Split a mesh into chunks of <= max_tris triangles, each its own mesh + instance."""
    pos, nrm, uv, tri = part
    ids = []
    for s in range(0, tri.shape[0], max_tris):
        t = tri[s : s + max_tris]
        used, inv = np.unique(t, return_inverse=True)
        m = b.add_mesh(pos[used], nrm[used], uv[used], inv.reshape(-1, 3), index_stride=index_stride)
        ids.append(b.add_instance(m, material, transform))
    return ids


# ---------------------------------------------------------------------------------------------
# config 1/2 — Cornell box (SURVEY.md §8d "Config 1")
# ---------------------------------------------------------------------------------------------
def cornell_box(fog=None, fog_transform=None, density=(3.0, 3.0, 3.0), albedo=(0.9, 0.9, 0.9), anisotropy=0.0):
    """fog: the bytes of a NanoVDB float grid (e.g. tests/golden/fog_sphere.npz["grid"]) placed in the box as a
    Medium component: the Cornell box with a cloud in it."""
    b = SceneBuilder("cornell_box" if fog is None else "foggy_cornell_box")
    white = b.add_material((0.73, 0.73, 0.73))
    red = b.add_material((0.65, 0.05, 0.05))
    green = b.add_material((0.12, 0.45, 0.15))
    light = b.add_emitter((17.0, 12.0, 4.0))

    def wall(p0, p1, p2, p3, n, mat, stride=4):
        pos, nrm, uv, tri = _quad(p0, p1, p2, p3, n)
        b.add_instance(b.add_mesh(pos, nrm, uv, tri, index_stride=stride), mat)

    # box spans [-1,1]^3, open towards +Z (the camera side)
    wall((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (0, 1, 0), white)  # floor
    wall((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, -1, 0), white, stride=2)  # ceiling (16-bit indices)
    wall((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (0, 0, 1), white)  # back
    wall((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (1, 0, 0), red)  # left
    wall((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0), green, stride=2)  # right

    # unit block (5 faces, no bottom): [-0.5,0.5]^2 x [0,1] in y; placed by instance transforms
    faces = [
        _quad((-0.5, 1, 0.5), (0.5, 1, 0.5), (0.5, 1, -0.5), (-0.5, 1, -0.5), (0, 1, 0)),
        _quad((-0.5, 0, 0.5), (0.5, 0, 0.5), (0.5, 1, 0.5), (-0.5, 1, 0.5), (0, 0, 1)),
        _quad((0.5, 0, -0.5), (-0.5, 0, -0.5), (-0.5, 1, -0.5), (0.5, 1, -0.5), (0, 0, -1)),
        _quad((0.5, 0, 0.5), (0.5, 0, -0.5), (0.5, 1, -0.5), (0.5, 1, 0.5), (1, 0, 0)),
        _quad((-0.5, 0, -0.5), (-0.5, 0, 0.5), (-0.5, 1, 0.5), (-0.5, 1, -0.5), (-1, 0, 0)),
    ]
    block = b.add_mesh(*_merge(faces))
    b.add_instance(block, white, translate((0.33, -1.0, 0.3)) @ rotate_y(-0.29) @ scale((0.6, 0.6, 0.6)))  # short block
    b.add_instance(block, white, translate((-0.35, -1.0, -0.35)) @ rotate_y(0.31) @ scale((0.6, 1.2, 0.6)))  # tall block

    # ceiling light, slightly below the ceiling, facing down
    wall((-0.24, 0.995, -0.2), (0.24, 0.995, -0.2), (0.24, 0.995, 0.18), (-0.24, 0.995, 0.18), (0, -1, 0), light)
    if fog is not None:
        b.add_medium(b.add_volume(fog), density_scale=density, albedo_scale=albedo, anisotropy=anisotropy, transform=fog_transform)
    sc = b.build()
    assert sc.triangle_count == 32
    return sc, {"eye": (0.0, 0.0, 3.9), "target": (0.0, 0.0, 0.0), "fovy": np.radians(39.3)}


def furnace_box(albedo=0.5, emission=1.0):
    """Closed cube whose six walls all emit `emission` and have diffuse albedo 0 (emitters do not
    scatter, disney_material.hlsli:83) — used for the furnace-style analytic check."""
    b = SceneBuilder("furnace")
    m = b.add_material((1.0, 1.0, 1.0), emission=emission, eta=0.0)
    parts = [
        _quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (0, 1, 0)),
        _quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, -1, 0)),
        _quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (0, 0, 1)),
        _quad((1, -1, 1), (-1, -1, 1), (-1, 1, 1), (1, 1, 1), (0, 0, -1)),
        _quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (1, 0, 0)),
        _quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0)),
    ]
    b.add_instance(b.add_mesh(*_merge(parts)), m)
    return b.build(), {"eye": (0.0, 0.0, 0.0), "target": (0.0, 0.0, -1.0), "fovy": np.radians(60.0)}


# ---------------------------------------------------------------------------------------------
# config 3/4 — procedural ~1M-triangle atrium (SURVEY.md §8d "Config 3")
# ---------------------------------------------------------------------------------------------
def _hash01(ix, iy, seed):
    """pcg-style integer hash -> [0,1); deterministic across platforms."""
    h = (ix.astype(np.uint64) * np.uint64(73856093)) ^ (iy.astype(np.uint64) * np.uint64(19349663)) ^ np.uint64(seed * 83492791 + 12345)
    with np.errstate(over="ignore"):  # the wrap-around is the hash
        h = (h * np.uint64(747796405) + np.uint64(2891336453)) & np.uint64(0xFFFFFFFF)
    h = (((h >> ((h >> np.uint64(28)) + np.uint64(4))) ^ h) * np.uint64(277803737)) & np.uint64(0xFFFFFFFF)
    h = ((h >> np.uint64(22)) ^ h) & np.uint64(0xFFFFFFFF)
    return h.astype(np.float64) / 4294967296.0


def _value_noise(x, y, seed):
    x0, y0 = np.floor(x), np.floor(y)
    fx, fy = x - x0, y - y0
    sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    ix, iy = x0.astype(np.int64) + 10000, y0.astype(np.int64) + 10000
    a = _hash01(ix, iy, seed)
    b = _hash01(ix + 1, iy, seed)
    c = _hash01(ix, iy + 1, seed)
    d = _hash01(ix + 1, iy + 1, seed)
    return (a * (1 - sx) + b * sx) * (1 - sy) + (c * (1 - sx) + d * sx) * sy


def atrium(target_tris=1_000_000):
    """Hall 30 x 12 x 10 (x,z,y): bumpy flagstone floor, brick walls, two rows of fluted columns
    carrying arches and an upper gallery, hanging drapes, a coffered ceiling with emissive
    skylight panels. `target_tris` scales the tessellation (1.0 -> ~1.0M triangles)."""
    k = np.sqrt(target_tris / 1_315_000.0)  # the unscaled tessellation below yields ~1.315M triangles

    def n(x):
        return max(2, int(round(x * k)))

    b = SceneBuilder("atrium")
    stone = b.add_material((0.62, 0.58, 0.52))
    brick = b.add_material((0.55, 0.33, 0.25))
    marble = b.add_material((0.78, 0.76, 0.72))
    plaster = b.add_material((0.70, 0.68, 0.62))
    cloth = [b.add_material(c) for c in ((0.65, 0.08, 0.08), (0.10, 0.20, 0.55), (0.12, 0.45, 0.15), (0.70, 0.55, 0.10))]
    wood = b.add_material((0.35, 0.22, 0.12))
    sky = b.add_emitter((22.0, 20.0, 17.0))
    lamp = b.add_emitter((30.0, 20.0, 9.0))

    LX, LZ, H = 15.0, 6.0, 10.0

    # floor: flagstones with mortar grooves + low-frequency undulation
    def floor(U, V):
        x, z = (U * 2 - 1) * LX, (V * 2 - 1) * LZ
        gx, gz = np.abs(((x * 1.0) % 1.0) - 0.5), np.abs(((z * 1.0) % 1.0) - 0.5)
        groove = np.clip((np.maximum(gx, gz) - 0.44) / 0.06, 0, 1)
        y = 0.02 * _value_noise(x * 0.7, z * 0.7, 1) - 0.025 * groove
        return np.stack([x, y, z], -1)

    add_chunked(b, grid_surface(floor, n(420), n(170)), stone)

    # long walls (z = +-LZ) and end walls (x = +-LX): brick relief
    def brick_relief(a, h, seed):
        row = np.floor(h * 4.0)
        col = a * 2.0 + 0.5 * (row % 2)
        ga, gh = np.abs((col % 1.0) - 0.5), np.abs(((h * 4.0) % 1.0) - 0.5)
        groove = np.clip((np.maximum(ga * 0.5 + 0.25, gh) - 0.43) / 0.07, 0, 1)
        return 0.03 * _value_noise(a * 3.0, h * 3.0, seed) - 0.03 * groove

    def wall_z(sign, seed):
        def f(U, V):
            x, y = (U * 2 - 1) * LX, V * H
            return np.stack([x, y, sign * (LZ + brick_relief(x, y, seed))], -1)

        return f

    def wall_x(sign, seed):
        def f(U, V):
            z, y = (U * 2 - 1) * LZ, V * H
            return np.stack([sign * (LX + brick_relief(z, y, seed)), y, z], -1)

        return f

    add_chunked(b, grid_surface(wall_z(-1, 2), n(300), n(100)), brick)
    add_chunked(b, grid_surface(wall_z(+1, 3), n(300), n(100), flip=True), brick)
    add_chunked(b, grid_surface(wall_x(-1, 4), n(120), n(100), flip=True), brick)
    add_chunked(b, grid_surface(wall_x(+1, 5), n(120), n(100)), brick)

    # ceiling with coffers
    def ceiling(U, V):
        x, z = (U * 2 - 1) * LX, (V * 2 - 1) * LZ
        cx, cz = np.abs(((x / 3.0) % 1.0) - 0.5), np.abs(((z / 3.0) % 1.0) - 0.5)
        coffer = np.clip((0.38 - np.maximum(cx, cz)) / 0.08, 0, 1)
        return np.stack([x, H + 0.25 * coffer, z], -1)

    add_chunked(b, grid_surface(ceiling, n(300), n(120), flip=True), plaster)

    # columns: two rows of fluted, slightly tapered shafts with a torus base and capital
    col_x = np.linspace(-LX + 2.5, LX - 2.5, 10)
    col_h = 6.0

    def shaft(U, V):
        a = U * 2 * np.pi
        y = V * col_h
        r = (0.42 - 0.06 * V) * (1.0 - 0.05 * np.abs(np.sin(a * 10))) + 0.10 * np.exp(-((V - 0.02) / 0.03) ** 2) + 0.14 * np.exp(-((V - 0.98) / 0.035) ** 2)
        return np.stack([r * np.cos(a), y, r * np.sin(a)], -1)

    shaft_mesh = grid_surface(shaft, n(96), n(72), flip=True)
    pos, nrm, uv, tri = shaft_mesh
    shaft_id = b.add_mesh(pos, nrm, uv, tri)  # one mesh, 20 transformed instances
    for z in (-3.2, 3.2):
        for i, x in enumerate(col_x):
            b.add_instance(shaft_id, marble, translate((x, 0.0, z)) @ rotate_y(0.37 * i))

    # arches between neighbouring columns (half tori), per row
    def arch(x0, x1, z):
        cx, R = 0.5 * (x0 + x1), 0.5 * (x1 - x0)

        def f(U, V):
            a = U * np.pi
            t = V * 2 * np.pi
            r = 0.22 * (1 + 0.08 * np.cos(t * 6))
            rr = R + r * np.cos(t)
            return np.stack([cx - rr * np.cos(a), col_h + rr * np.sin(a) * 0.75, z + r * np.sin(t)], -1)

        return f

    arch_parts = []
    for z in (-3.2, 3.2):
        for i in range(len(col_x) - 1):
            arch_parts.append(grid_surface(arch(col_x[i], col_x[i + 1], z), n(56), n(20), flip=True))
    add_chunked(b, _merge(arch_parts), marble)

    # upper gallery slabs along both long walls + balustrade rails
    def slab(z0, z1, y):
        def f(U, V):
            x, z = (U * 2 - 1) * (LX - 0.5), z0 + (z1 - z0) * V
            return np.stack([x, y + 0.01 * _value_noise(x * 2, z * 2, 7), z], -1)

        return f

    gallery = [grid_surface(slab(-LZ, -3.0, 7.6), n(200), n(24), flip=False), grid_surface(slab(3.0, LZ, 7.6), n(200), n(24), flip=False)]
    gallery += [grid_surface(slab(-LZ, -3.0, 7.3), n(100), n(12), flip=True), grid_surface(slab(3.0, LZ, 7.3), n(100), n(12), flip=True)]
    add_chunked(b, _merge(gallery), wood)

    def rail(z, y):
        def f(U, V):
            x = (U * 2 - 1) * (LX - 0.5)
            t = V * 2 * np.pi
            return np.stack([x, y + 0.06 * np.sin(t), z + 0.06 * np.cos(t)], -1)

        return f

    rails = [grid_surface(rail(z, y), n(240), n(10)) for z in (-3.0, 3.0) for y in (8.0, 8.5)]
    add_chunked(b, _merge(rails), wood)

    # drapes: hanging cloth with folds, between columns on the gallery level and across the nave
    def drape(x0, z0, width, height, axis, seed):
        def f(U, V):
            s = (U - 0.5) * width
            y = 7.2 - V * height
            fold = 0.18 * np.sin(U * 2 * np.pi * 5 + seed) * (0.3 + 0.7 * V) + 0.05 * _value_noise(U * 8, V * 8, seed)
            sag = 0.25 * np.sin(np.pi * U) * V
            if axis == 0:
                return np.stack([x0 + s, y - sag, z0 + fold], -1)
            return np.stack([x0 + fold, y - sag, z0 + s], -1)

        return f

    d = 0
    for i in range(0, len(col_x) - 1, 2):
        for z in (-4.6, 4.6):
            add_chunked(b, grid_surface(drape(0.5 * (col_x[i] + col_x[i + 1]), z, 2.6, 3.5, 0, 10 + d), n(150), n(150)), cloth[d % 4])
            d += 1
    for x in (-9.0, 0.0, 9.0):
        add_chunked(b, grid_surface(drape(x, 0.0, 5.0, 2.5, 2, 10 + d), n(170), n(120)), cloth[d % 4])
        d += 1

    # lights: two skylight panels in the ceiling and two warm lamps on the end walls
    def panel(cx, cz, sx, sz, y):
        return _quad((cx - sx, y, cz - sz), (cx + sx, y, cz - sz), (cx + sx, y, cz + sz), (cx - sx, y, cz + sz), (0, -1, 0))

    for cx in (-7.5, 7.5):
        pos, nrm, uv, tri = panel(cx, 0.0, 2.5, 1.2, H - 0.02)
        b.add_instance(b.add_mesh(pos, nrm, uv, tri), sky)
    for sx in (-1, 1):
        pos, nrm, uv, tri = _quad((sx * (LX - 0.15), 3.0, -0.8), (sx * (LX - 0.15), 3.0, 0.8), (sx * (LX - 0.15), 4.2, 0.8), (sx * (LX - 0.15), 4.2, -0.8), (-sx, 0, 0))
        b.add_instance(b.add_mesh(pos, nrm, uv, tri), lamp)

    sc = b.build()
    return sc, {"eye": (-12.5, 2.2, 1.4), "target": (6.0, 4.0, -0.6), "fovy": np.radians(60.0)}


# ---------------------------------------------------------------------------------------------
# config 5 — instanced forest (SURVEY.md §8d "Config 5")
# ---------------------------------------------------------------------------------------------
def _tree(seed, tris_target=10_000):
    rng = np.random.RandomState(seed)
    parts = []
    k = np.sqrt(tris_target / 10_000.0)

    def n(x):
        return max(3, int(round(x * k)))

    h = 3.0 + rng.rand() * 1.5

    def trunk(U, V):
        a = U * 2 * np.pi
        r = (0.22 - 0.12 * V) * (1 + 0.08 * np.sin(a * 7 + V * 9))
        return np.stack([r * np.cos(a) + 0.1 * np.sin(V * 3 + seed), V * h, r * np.sin(a)], -1)

    parts.append(grid_surface(trunk, n(20), n(24), flip=True))
    nb = 9
    for i in range(nb):
        cy = h * (0.55 + 0.12 * i / 2.0)
        ang = i * 2.399963 + seed
        off = 0.9 * (1 - i / (nb * 1.3))
        c = np.array([off * np.cos(ang), cy, off * np.sin(ang)])
        R = 0.9 - 0.05 * i + 0.2 * rng.rand()
        s = int(rng.randint(1, 1000))

        def blob(U, V, c=c, R=R, s=s):
            th, ph = V * np.pi, U * 2 * np.pi
            r = R * (1 + 0.25 * (_value_noise(U * 6, V * 6, s) - 0.5) + 0.1 * np.sin(ph * 5) * np.sin(th * 4))
            return np.stack([c[0] + r * np.sin(th) * np.cos(ph), c[1] + 0.8 * r * np.cos(th), c[2] + r * np.sin(th) * np.sin(ph)], -1)

        parts.append(grid_surface(blob, n(24), n(20), flip=True))
    return _merge(parts)


def forest(n_instances=1000, tree_tris=10_000, tree_kinds=4):
    b = SceneBuilder("forest")
    soil = b.add_material((0.30, 0.24, 0.16))
    leaf = [b.add_material(c) for c in ((0.12, 0.35, 0.08), (0.20, 0.42, 0.10), (0.30, 0.38, 0.08), (0.10, 0.28, 0.12))]
    sun = b.add_emitter((60.0, 55.0, 45.0))
    ext = 2.2 * np.sqrt(n_instances)

    def ground(U, V):
        x, z = (U * 2 - 1) * ext, (V * 2 - 1) * ext
        return np.stack([x, 0.6 * _value_noise(x * 0.15, z * 0.15, 3) + 0.05 * _value_noise(x, z, 4), z], -1)

    add_chunked(b, grid_surface(ground, 180, 180), soil)
    kinds = []
    for s in range(tree_kinds):
        pos, nrm, uv, tri = _tree(s + 1, tree_tris)
        kinds.append(b.add_mesh(pos, nrm, uv, tri[:MAX_TRIS]))
    rng = np.random.RandomState(7)
    side = int(np.ceil(np.sqrt(n_instances)))
    for i in range(n_instances):
        gx, gz = i % side, i // side
        x = (gx + 0.5 + 0.7 * (rng.rand() - 0.5)) / side * 2 * ext - ext
        z = (gz + 0.5 + 0.7 * (rng.rand() - 0.5)) / side * 2 * ext - ext
        y = 0.6 * float(_value_noise(np.array(x * 0.15), np.array(z * 0.15), 3))
        s = 0.8 + 0.7 * rng.rand()
        b.add_instance(kinds[i % tree_kinds], leaf[int(rng.randint(0, 4))], translate((x, y - 0.05, z)) @ rotate_y(rng.rand() * 6.283) @ scale((s, s * (0.9 + 0.3 * rng.rand()), s)))
    # one large emissive panel high above (an overcast "sky" light)
    pos, nrm, uv, tri = _quad((-ext, 40.0, -ext), (ext, 40.0, -ext), (ext, 40.0, ext), (-ext, 40.0, ext), (0, -1, 0))
    b.add_instance(b.add_mesh(pos, nrm, uv, tri), sun)
    return b.build(), {"eye": (-0.55 * ext, 9.0, -0.55 * ext), "target": (0.0, 1.5, 0.0), "fovy": np.radians(55.0)}


def clustered_box(n_coincident=60_000, n_cluster=40_000):
    """A Cornell box holding a pathological mesh for a Morton-code builder: `n_coincident` triangles that are copies of ONE
    triangle (equal centroids: one Morton code, told apart only by index) and `n_cluster` triangles packed into a ball of
    1e-4 units (codes that share all but their last bits), plus the box walls far away on the same grid. A radix tree over
    such keys is as deep as its longest common prefix allows; the traversal stack has to take it or the builder has to
    give in (stratum_amd/csrc/bvh_build.cpp: LBVH_MAX_HEIGHT)."""
    sc0, cam = cornell_box()
    b = SceneBuilder("clustered_box")
    white = b.add_material((0.73, 0.73, 0.73))
    red = b.add_material((0.65, 0.05, 0.05))
    light = b.add_emitter((17.0, 12.0, 4.0))

    def wall(p0, p1, p2, p3, n, mat):
        pos, nrm, uv, tri = _quad(p0, p1, p2, p3, n)
        b.add_instance(b.add_mesh(pos, nrm, uv, tri), mat)

    wall((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (0, 1, 0), white)
    wall((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, -1, 0), white)
    wall((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (0, 0, 1), white)
    wall((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (1, 0, 0), red)
    wall((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0), white)
    wall((-0.24, 0.995, -0.2), (0.24, 0.995, -0.2), (0.24, 0.995, 0.18), (-0.24, 0.995, 0.18), (0, -1, 0), light)
    rng = np.random.RandomState(11)
    # the coincident fan: every triangle has the same three corners (a visible 0.5-unit triangle facing the camera)
    base = np.array([[-0.3, -0.6, 0.1], [0.3, -0.6, 0.1], [0.0, -0.1, 0.1]], np.float32)
    pos = [np.tile(base, (n_coincident, 1))]
    # the cluster: tiny triangles inside a 1e-4 ball
    c = np.array([0.45, -0.2, -0.3], np.float32)
    pos.append((c + rng.normal(size=(n_cluster * 3, 3)) * 3e-5).astype(np.float32))
    pos = np.concatenate(pos)
    n_tri = n_coincident + n_cluster
    tri = np.arange(n_tri * 3).reshape(n_tri, 3)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (pos.shape[0], 1))
    uv = np.zeros((pos.shape[0], 2), np.float32)
    add_chunked(b, (pos, nrm, uv, tri), red)
    return b.build(), cam


SCENES = {"cornell_box": cornell_box, "atrium": atrium, "forest": forest, "furnace": furnace_box}


# ---------------------------------------------------------------------------------------------
# textured variant of the Cornell box (image values, normal maps, ray cones)
# ---------------------------------------------------------------------------------------------
def _checker(n, cells, a, b):
    y, x = np.mgrid[0:n, 0:n]
    m = ((x * cells // n) + (y * cells // n)) % 2
    img = np.where(m[..., None] == 0, np.asarray(a, np.float32), np.asarray(b, np.float32)).astype(np.float32)
    return np.concatenate([img, np.ones((n, n, 1), np.float32)], -1)


def _noise_image(n, seed, lo, hi, freq=8.0):
    y, x = np.mgrid[0:n, 0:n].astype(np.float64)
    v = _value_noise(x * freq / n, y * freq / n, seed) * 0.6 + _value_noise(x * freq * 4 / n, y * freq * 4 / n, seed + 1) * 0.4
    img = np.asarray(lo, np.float64) + (np.asarray(hi, np.float64) - np.asarray(lo, np.float64)) * v[..., None]
    return np.concatenate([img, np.ones((n, n, 1))], -1).astype(np.float32)


def _bump_image(n, cells):
    """Tangent-space normal map of a grid of rounded studs."""
    y, x = np.mgrid[0:n, 0:n].astype(np.float64)
    u, v = (x * cells / n) % 1.0 - 0.5, (y * cells / n) % 1.0 - 0.5
    r2 = u * u + v * v
    hgt = np.exp(-r2 * 18.0)
    dx = np.gradient(hgt, axis=1) * n / cells * 0.25
    dy = np.gradient(hgt, axis=0) * n / cells * 0.25
    nrm = np.stack([-dx, -dy, np.ones_like(dx)], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    return np.concatenate([nrm * 0.5 + 0.5, np.ones((n, n, 1))], -1).astype(np.float32)


def textured_box(mirror_map=False):
    """Cornell-like box: checker floor with a normal map, noise-textured walls, a textured emitter, metal block
    with a roughness map. Exercises image values, mip selection through ray cones, normal maps.
    mirror_map: the roughness map is a checker whose dark squares are 0, so the metal ball (constant roughness 0.6, not
    specular by its constants) IS specular at those texels (is_specular is evaluated on value * texel,
    disney_material.hlsli:125 after image_value.h:194-198) and paths run on past gMaxDiffuseVertices there."""
    b = SceneBuilder("textured_box")
    img_checker = b.add_image(_checker(256, 8, (0.9, 0.9, 0.9), (0.15, 0.2, 0.6)))
    img_noise = b.add_image(_noise_image(128, 5, (0.3, 0.3, 0.3), (1.0, 1.0, 1.0)))
    img_bump = b.add_image(_bump_image(128, 8))
    if mirror_map:
        img_rough = b.add_image(_checker(64, 4, (1.0, 0.0, 0.0), (1.0, 1.0, 0.0)))
    else:
        img_rough = b.add_image(_noise_image(64, 9, (1.0, 0.05, 0.0, ), (1.0, 1.0, 0.0), freq=4.0))
    img_light = b.add_image(_checker(16, 4, (1.0, 1.0, 1.0), (0.3, 0.3, 0.3)))

    floor = b.add_material((1.0, 1.0, 1.0), roughness=0.4)
    b.set_material_images(floor, base_color_image=img_checker, bump_image=img_bump, bump_strength=1.5)
    wall = b.add_material((0.8, 0.75, 0.7))
    b.set_material_images(wall, base_color_image=img_noise)
    red = b.add_material((0.65, 0.05, 0.05))
    metal = b.add_material((0.9, 0.7, 0.4), metallic=1.0, roughness=0.6)
    b.set_material_images(metal, params_image=img_rough)
    light = b.add_emitter((17.0, 12.0, 4.0))
    b.set_material_images(light, base_color_image=img_light)

    def wall_q(p0, p1, p2, p3, n, mat, uvscale=1.0):
        pos, nrm, uv, tri = _quad(p0, p1, p2, p3, n)
        b.add_instance(b.add_mesh(pos, nrm, uv * uvscale, tri), mat)

    wall_q((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (0, 1, 0), floor, 2.0)
    wall_q((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, -1, 0), wall)
    wall_q((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (0, 0, 1), wall, 3.0)
    wall_q((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (1, 0, 0), red)
    wall_q((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0), wall)
    sphere = grid_surface(lambda U, V: np.stack([np.sin(V * np.pi) * np.cos(U * 2 * np.pi), np.cos(V * np.pi), np.sin(V * np.pi) * np.sin(U * 2 * np.pi)], -1), 32, 20, flip=True)
    ball = b.add_mesh(*sphere)
    b.add_instance(ball, metal, translate((0.35, -0.55, 0.2)) @ scale(0.45))
    b.add_instance(ball, floor, translate((-0.45, -0.65, -0.3)) @ rotate_y(0.5) @ scale((0.35, 0.35, 0.35)))
    wall_q((-0.24, 0.995, -0.2), (0.24, 0.995, -0.2), (0.24, 0.995, 0.18), (-0.24, 0.995, 0.18), (0, -1, 0), light)
    return b.build(), {"eye": (0.0, 0.0, 3.9), "target": (0.0, 0.0, 0.0), "fovy": np.radians(39.3)}


SCENES["textured_box"] = textured_box


# ---------------------------------------------------------------------------------------------
# sphere instances / sphere lights and environment maps
# ---------------------------------------------------------------------------------------------
def sky_image(w=64, h=32, sun=(0.3, 0.25), sun_radiance=60.0):
    """Procedural lat-long environment (RGBA32F, row 0 = +Y pole): blue gradient, a ground tone and a soft sun."""
    v, u = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
    up = np.cos(np.pi * v)
    sky = np.stack([0.35 + 0.3 * (1 - up), 0.5 + 0.3 * (1 - up), 0.9 + 0.0 * up], -1) * np.clip(up + 0.15, 0.05, 1)[..., None] * 1.5
    ground = np.stack([0.25 + 0 * up, 0.22 + 0 * up, 0.2 + 0 * up], -1)
    img = np.where((up > 0)[..., None], sky, ground)
    d2 = ((u - sun[0] + 0.5) % 1.0 - 0.5) ** 2 + (v - sun[1]) ** 2
    img = img + sun_radiance * np.exp(-d2 / (2 * 0.02**2))[..., None] * np.array([1.0, 0.9, 0.7])
    out = np.ones((h, w, 4), np.float32)
    out[..., :3] = img
    return out


def spheres_room():
    """Closed room lit by one sphere light and one quad light, with diffuse / metal / glass sphere instances
    (SpherePrimitive) next to a triangle box: spheres in the traversal contract, sphere-light sampling."""
    b = SceneBuilder("spheres_room")
    white = b.add_material((0.73, 0.73, 0.73))
    red = b.add_material((0.65, 0.05, 0.05))
    green = b.add_material((0.12, 0.45, 0.15))
    metal = b.add_material((0.9, 0.8, 0.5), metallic=1.0, roughness=0.15)
    glass = b.add_material((1.0, 1.0, 1.0), transmission=1.0, roughness=0.05, eta=1.5)
    blue = b.add_material((0.2, 0.3, 0.8), roughness=0.6)
    lamp = b.add_emitter((18.0, 15.0, 10.0))
    panel = b.add_emitter((6.0, 7.0, 9.0))
    S = 2.0
    b.add_instance(b.add_mesh(*_quad((-S, 0, S), (S, 0, S), (S, 0, -S), (-S, 0, -S), (0, 1, 0))), white)
    b.add_instance(b.add_mesh(*_quad((-S, 3, -S), (S, 3, -S), (S, 3, S), (-S, 3, S), (0, -1, 0))), white)
    b.add_instance(b.add_mesh(*_quad((-S, 0, -S), (S, 0, -S), (S, 3, -S), (-S, 3, -S), (0, 0, 1))), white)
    b.add_instance(b.add_mesh(*_quad((-S, 0, S), (-S, 0, -S), (-S, 3, -S), (-S, 3, S), (1, 0, 0))), red)
    b.add_instance(b.add_mesh(*_quad((S, 0, -S), (S, 0, S), (S, 3, S), (S, 3, -S), (-1, 0, 0))), green)
    b.add_instance(b.add_mesh(*_quad((-0.5, 2.99, -0.3), (0.5, 2.99, -0.3), (0.5, 2.99, 0.3), (-0.5, 2.99, 0.3), (0, -1, 0))), panel, translate((1.0, 0.0, -0.8)))
    b.add_sphere(lamp, 0.25, translate((-0.9, 2.2, -0.6)))
    b.add_sphere(blue, 0.5, translate((-1.0, 0.5, -0.8)))
    b.add_sphere(metal, 0.45, translate((0.2, 0.45, -1.1)))
    b.add_sphere(glass, 0.35, translate((0.9, 0.35, 0.2)))
    b.add_sphere(white, 0.2, translate((-0.2, 0.4, 0.6)) @ scale((1.0, 2.0, 1.0)))  # radius scaled by det = 2
    return b.build(), {"eye": (0.0, 1.5, 4.6), "target": (0.0, 1.2, 0.0), "fovy": np.radians(45.0)}


def environment_scene(image=True, emitter=True):
    """Objects on a ground plane under an environment map (lat-long image with a sun, or a constant colour), with an
    optional triangle emitter so that the environment / emitter choice of light sampling is exercised."""
    b = SceneBuilder("environment")
    ground = b.add_material((0.5, 0.5, 0.5), roughness=0.8)
    metal = b.add_material((0.95, 0.95, 0.95), metallic=1.0, roughness=0.05)
    clay = b.add_material((0.7, 0.35, 0.25))
    S = 6.0
    b.add_instance(b.add_mesh(*_quad((-S, 0, S), (S, 0, S), (S, 0, -S), (-S, 0, -S), (0, 1, 0))), ground)
    parts = [
        _quad((-0.5, 0, 0.5), (0.5, 0, 0.5), (0.5, 1, 0.5), (-0.5, 1, 0.5), (0, 0, 1)),
        _quad((0.5, 0, -0.5), (-0.5, 0, -0.5), (-0.5, 1, -0.5), (0.5, 1, -0.5), (0, 0, -1)),
        _quad((-0.5, 0, -0.5), (-0.5, 0, 0.5), (-0.5, 1, 0.5), (-0.5, 1, -0.5), (-1, 0, 0)),
        _quad((0.5, 0, 0.5), (0.5, 0, -0.5), (0.5, 1, -0.5), (0.5, 1, 0.5), (1, 0, 0)),
        _quad((-0.5, 1, 0.5), (0.5, 1, 0.5), (0.5, 1, -0.5), (-0.5, 1, -0.5), (0, 1, 0)),
    ]
    b.add_instance(b.add_mesh(*_merge(parts)), clay, translate((-1.0, 0.0, 0.0)) @ rotate_y(0.5))
    b.add_sphere(metal, 0.6, translate((0.8, 0.6, 0.2)))
    if emitter:
        lamp = b.add_emitter((8.0, 4.0, 2.0))
        b.add_instance(b.add_mesh(*_quad((-0.3, 0, -0.3), (0.3, 0, -0.3), (0.3, 0, 0.3), (-0.3, 0, 0.3), (0, -1, 0))), lamp, translate((0.0, 2.0, 0.5)))
    if image:
        b.set_environment((1.0, 1.0, 1.0), b.add_image(sky_image()))
    else:
        b.set_environment((0.6, 0.8, 1.2))
    return b.build(), {"eye": (0.0, 1.3, 4.5), "target": (0.0, 0.6, 0.0), "fovy": np.radians(40.0)}

def leaf_mask(n=64, seed=3):
    """Coverage image of a leaf-like shape (1 inside, 0 outside, a soft rim in between) with a few holes."""
    v, u = np.meshgrid((np.arange(n) + 0.5) / n, (np.arange(n) + 0.5) / n, indexing="ij")
    x, y = 2 * u - 1, 2 * v - 1
    r = np.sqrt((x / 0.55) ** 2 + (y / 0.95) ** 2) + 0.15 * np.sin(6 * np.arctan2(y, x))
    a = np.clip((1.0 - r) * 6.0, 0.0, 1.0)
    rng = np.random.RandomState(seed)
    for _ in range(5):
        cx, cy = rng.uniform(-0.3, 0.3), rng.uniform(-0.6, 0.6)
        a *= np.clip((np.sqrt((x - cx) ** 2 + (y - cy) ** 2) - 0.08) * 12.0, 0.0, 1.0)
    return a.astype(np.float32)


def foliage():
    """Alpha-masked cards (leaves) over a floor, lit by a quad light: with eAlphaTest the cut-out shapes and their
    shadows appear, without it the cards are solid quads (intersection.hlsli:117-131)."""
    b = SceneBuilder("foliage")
    floor = b.add_material((0.6, 0.6, 0.6))
    leaf = b.add_material((0.15, 0.55, 0.1), roughness=0.6)
    leaf2 = b.add_material((0.6, 0.5, 0.1), roughness=0.6)
    lamp = b.add_emitter((30.0, 28.0, 25.0))
    mask = b.add_image1(leaf_mask())
    b.set_material_alpha_mask(leaf, mask)
    b.set_material_alpha_mask(leaf2, b.add_image1(leaf_mask(32, 9)))
    S = 4.0
    b.add_instance(b.add_mesh(*_quad((-S, 0, S), (S, 0, S), (S, 0, -S), (-S, 0, -S), (0, 1, 0))), floor)
    b.add_instance(b.add_mesh(*_quad((-0.6, 0, -0.6), (0.6, 0, -0.6), (0.6, 0, 0.6), (-0.6, 0, 0.6), (0, -1, 0))), lamp, translate((0.0, 4.0, 0.0)))
    p, n, uv, tri = _quad((-0.5, 0, 0.5), (0.5, 0, 0.5), (0.5, 0, -0.5), (-0.5, 0, -0.5), (0, 1, 0))
    uv = np.array([[0, 1], [1, 1], [1, 0], [0, 0]], np.float32)
    card = b.add_mesh(p, n, uv, tri)
    rng = np.random.RandomState(5)
    for k in range(24):
        m = translate((rng.uniform(-1.5, 1.5), rng.uniform(0.6, 2.2), rng.uniform(-1.5, 1.0))) @ rotate_y(rng.uniform(0, 6.28)) @ rotate_x(rng.uniform(-0.9, 0.9)) @ scale((rng.uniform(0.5, 1.0),) * 3)
        b.add_instance(card, leaf if k % 3 else leaf2, m)
    return b.build(), {"eye": (0.0, 1.6, 5.0), "target": (0.0, 1.0, 0.0), "fovy": np.radians(42.0)}


SCENES["foliage"] = foliage
SCENES["spheres_room"] = spheres_room
SCENES["environment"] = environment_scene


def fog_box(**kwargs):
    """The Cornell box with the test suite's NanoVDB fog sphere in it (tests/golden/fog_sphere.npz): media for bench.py --scene."""
    import os

    grid = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "fog_sphere.npz"))["grid"]
    return cornell_box(fog=grid, **kwargs)


SCENES["fog_box"] = fog_box
