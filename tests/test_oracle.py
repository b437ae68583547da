import os
"""Pins of the CPU oracle (no GPU). The reference ships no tests or golden vectors (SURVEY.md §4), so
the oracle is pinned by independent restatements of the integer code, by IEEE conversions that numpy
implements on its own, and by analytic properties the estimator must have."""
import numpy as np
import pytest

from oracle import oracle_py as orc
from stratum_amd import camera, scenes, wire
from stratum_amd.scene import SceneBuilder, build_distributions, rotate_y, scale, translate

U32 = np.uint32


# ---------------------------------------------------------------------------------------------
# R1/R2 integer hashes: independent numpy restatement of rng.hlsli:6-47
# ---------------------------------------------------------------------------------------------
def np_pcg4d(v):
    v = v.astype(np.uint64)
    M = np.uint64(0xFFFFFFFF)
    v = (v * np.uint64(1664525) + np.uint64(1013904223)) & M
    x, y, z, w = v[:, 0], v[:, 1], v[:, 2], v[:, 3]
    x = (x + y * w) & M
    y = (y + z * x) & M
    z = (z + x * y) & M
    w = (w + y * z) & M
    x, y, z, w = [a ^ (a >> np.uint64(16)) for a in (x, y, z, w)]
    x = (x + y * w) & M
    y = (y + z * x) & M
    z = (z + x * y) & M
    w = (w + y * z) & M
    return np.stack([x, y, z, w], 1).astype(np.uint32)


def test_pcg4d_matches_independent_restatement():
    rng = np.random.RandomState(0)
    v = rng.randint(0, 2**32, (4096, 4), dtype=np.uint64).astype(np.uint32)
    v[0] = 0
    v[1] = 0xFFFFFFFF
    assert np.array_equal(orc.pcg4d(v), np_pcg4d(v))


def test_pcg_and_xxhash_scalar():
    def pcg(v):
        state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
        word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
        return ((word >> 22) ^ word) & 0xFFFFFFFF

    def xx(p):
        P2, P3, P4, P5 = 2246822519, 3266489917, 668265263, 374761393
        h = (p + P5) & 0xFFFFFFFF
        h = (P4 * (((h << 17) | (h >> 15)) & 0xFFFFFFFF)) & 0xFFFFFFFF
        h = (P2 * (h ^ (h >> 15))) & 0xFFFFFFFF
        h = (P3 * (h ^ (h >> 13))) & 0xFFFFFFFF
        return h ^ (h >> 16)

    for v in [0, 1, 2, 12345, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 3141592653]:
        assert orc.lib().orc_pcg(v) == pcg(v)
        assert orc.lib().orc_xxhash32(v) == xx(v)


def test_xxhash32_against_the_xxhash_library():
    """A known answer from OUTSIDE this repository: rng.hlsli:6-15 is the Shadertoy shortening of XXH32 for one 32-bit
    word (cited there: github.com/Cyan4973/xxHash) — the library's 4-byte path is
        h = seed + PRIME5 + len;  h += word * PRIME3;  h = rotl(h, 17) * PRIME4;  avalanche(h)
    and the shader keeps the rotate, PRIME4 and the avalanche but starts from h = p + PRIME5. So for every p there is a
    word x with PRIME5 + 4 + x * PRIME3 == p + PRIME5 (PRIME3 is odd, hence invertible mod 2^32), and
    xxhash32(p) must equal the real XXH32 of that word with seed 0, computed by the xxhash package (C reference code)."""
    import struct

    xxhash = pytest.importorskip("xxhash")
    P3 = 3266489917
    inv = pow(P3, -1, 1 << 32)
    assert (P3 * inv) & 0xFFFFFFFF == 1
    rng = np.random.RandomState(7)
    values = [0, 1, 4, 5, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF] + [int(v) for v in rng.randint(0, 2**32, 2000, dtype=np.uint64)]
    for p in values:
        x = ((p - 4) * inv) & 0xFFFFFFFF
        assert orc.lib().orc_xxhash32(p) == xxhash.xxh32(struct.pack("<I", x), seed=0).intdigest(), hex(p)
    # and the library itself answers its published test value for the empty input (xxHash README / xxhsum self-test)
    assert xxhash.xxh32(b"", seed=0).intdigest() == 0x02CC5D05


def test_rng_stream_is_counter_based():
    """rng.hlsli:35-47: counter++ THEN hash; float = asfloat(0x3f800000 | u >> 9) - 1 in [0, 1)."""
    f = orc.rng_floats(17, 4, 9, 0, 64)
    assert (f >= 0).all() and (f < 1).all()
    state = np.array([[17, 4, 9, c] for c in range(1, 65)], np.uint32)
    u = np_pcg4d(state)[:, 0]
    expect = ((u >> 9) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1)
    assert np.array_equal(f, expect)
    # restartable from any counter (what ShadowRayData.rng_offset relies on, path.hlsli:358)
    assert np.array_equal(orc.rng_floats(17, 4, 9, 10, 8), f[10:18])


# ---------------------------------------------------------------------------------------------
# R3 f16 conversions and octahedral normals: numpy's own IEEE conversion is the independent check
# ---------------------------------------------------------------------------------------------
def test_f32tof16_is_round_to_nearest_even():
    rng = np.random.RandomState(1)
    x = np.concatenate(
        [
            rng.uniform(-2, 2, 200000).astype(np.float32),
            (rng.uniform(-1, 1, 50000) * 10.0 ** rng.uniform(-9, 5, 50000)).astype(np.float32),
            np.array([0, -0.0, 65504, 65519.99, 65520, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, np.inf, -np.inf], np.float32),
        ]
    )
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16).astype(np.uint32)
    assert np.array_equal(orc.f32tof16(x), want)


def test_f16tof32_exact_for_all_halves():
    h = np.arange(65536, dtype=np.uint32)
    want = h.astype(np.uint16).view(np.float16).astype(np.float32)
    got = orc.f16tof32(h)
    nan = np.isnan(want)
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))
    assert np.isnan(got[nan]).all()


def test_octahedral_normals_roundtrip():
    rng = np.random.RandomState(2)
    v = rng.normal(size=(100000, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    u = orc.unpack_normal(orc.pack_normal(v))
    assert np.abs(np.linalg.norm(u, axis=1) - 1).max() < 1e-6
    assert np.abs((u * v).sum(1) - 1).max() < 2e-6  # fp16 octahedral: < 0.12 degrees
    # packing the unpacked vector is a fixed point (the quantised normal is what every cosine sees, B1)
    assert np.array_equal(orc.pack_normal(u[:6]), orc.pack_normal(v[:6]))


def test_detmath_against_libm():
    rng = np.random.RandomState(3)
    x = rng.uniform(0, 2 * np.pi, 200000).astype(np.float32)
    s, c = orc.sincos(x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2.5e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 2.5e-7
    a = (10.0 ** rng.uniform(-7, 2, 100000)).astype(np.float32)
    assert np.abs(orc.log(a) / np.log(a.astype(np.float64)) - 1)[np.abs(np.log(a)) > 1e-2].max() < 1e-6
    b = rng.uniform(0, 1, 100000).astype(np.float32)
    a = rng.uniform(1e-6, 1e-2, 100000).astype(np.float32)
    assert np.abs(orc.pow(a, b) / np.power(a.astype(np.float64), b) - 1).max() < 5e-6


def test_ray_offset_properties():
    """intersection.hlsli:44-62: the offset point lies on the normal's side, |delta| tiny, exact branch at 1/32."""
    rng = np.random.RandomState(4)
    p = (rng.uniform(-1, 1, (50000, 3)) * 10.0 ** rng.uniform(-3, 3, (50000, 1))).astype(np.float32)
    n = rng.normal(size=(50000, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    q = orc.ray_offset(p, n)
    d = (q.astype(np.float64) - p) * n
    assert (d >= 0).all()  # every component moves along the normal's sign (or not at all)
    assert np.abs(q - p).max(axis=1).max() <= 1e-4 * np.abs(p).max(axis=1).max() + 2e-5
    small = np.abs(p) < 1 / 32.0
    want = (p + n * np.float32(1 / 65536.0)).astype(np.float32)
    assert np.array_equal(q[small], want[small])


# ---------------------------------------------------------------------------------------------
# M2/M3 Disney BSDF: sample/eval consistency, pdf normalisation, energy
# ---------------------------------------------------------------------------------------------
def material(**kw):
    b = SceneBuilder()
    b.add_material(kw.pop("base_color", (0.8, 0.6, 0.4)), **kw)
    return b._materials[0]


MATERIALS = {
    "diffuse": dict(),
    "rough_diffuse_subsurface": dict(roughness=0.7, subsurface=0.4),
    "metal": dict(metallic=1.0, roughness=0.4),
    "aniso_metal": dict(metallic=0.9, roughness=0.5, anisotropic=0.6),
    "glass": dict(transmission=1.0, roughness=0.3, eta=1.5),
    "clearcoat": dict(clearcoat=1.0, clearcoat_gloss=0.6, roughness=0.5),
    "mixed": dict(metallic=0.3, roughness=0.4, transmission=0.4, clearcoat=0.5, clearcoat_gloss=0.3, subsurface=0.2),
}


@pytest.mark.parametrize("name", sorted(MATERIALS))
def test_disney_sample_agrees_with_eval(name):
    rec = material(**MATERIALS[name])
    rng = np.random.RandomState(5)
    n = 20000
    di = rng.normal(size=(n, 3)).astype(np.float32)
    di /= np.linalg.norm(di, axis=1, keepdims=True)
    if "glass" not in name and name != "mixed":
        di[:, 2] = np.abs(di[:, 2])
    rnd = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    smp = orc.disney_sample(rec, di, rnd)
    do = smp[:, 0:3].copy()
    ok = smp[:, 3] > 1e-4
    ev = orc.disney_eval(rec, di, do)
    f_s, pdf_s, beta = smp[:, 7:10], smp[:, 3], smp[:, 10:13]
    # eval() of the sampled direction returns the same f and pdf that sample() reported
    good = ok & (np.abs(do[:, 2]) > 1e-3) & (np.abs(di[:, 2]) > 1e-3)
    # sample() prices a reflection that ends below the horizon (VNDF on a rough lobe) as a reflection, while
    # eval() classifies by hemisphere (disney_material.hlsli:155,237-250): compare only where the two agree
    refracted = smp[:, 5] != 0
    same_side = di[:, 2] * do[:, 2] > 0
    good &= np.where(refracted, ~same_side, same_side)
    assert good.sum() > n // 2
    rel = np.abs(ev[good, 3] - pdf_s[good]) / np.maximum(pdf_s[good], 1e-3)
    assert np.percentile(rel, 99) < 2e-2
    relf = np.abs(ev[good, 0:3] - f_s[good]) / np.maximum(np.abs(f_s[good]), 1e-2)
    assert np.percentile(relf, 99) < 2e-2
    # beta *= f / pdf (disney_material.hlsli:313)
    w = f_s[good] / pdf_s[good, None]
    assert np.allclose(beta[good], w, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["diffuse", "rough_diffuse_subsurface", "metal", "clearcoat"])
def test_disney_pdf_integrates_to_one_and_conserves_energy(name):
    rec = material(base_color=(1.0, 1.0, 1.0), **MATERIALS[name])
    rng = np.random.RandomState(6)
    n = 400000
    z = rng.uniform(-1, 1, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    r = np.sqrt(1 - z * z)
    do = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)
    th = 0.6
    di = np.tile(np.array([np.sin(th), 0, np.cos(th)], np.float32), (n, 1))
    ev = orc.disney_eval(rec, di, do).astype(np.float64)
    ipdf = ev[:, 3].mean() * 4 * np.pi
    alb = ev[:, 0].mean() * 4 * np.pi  # f already contains |cos|
    w_total = {"diffuse": 1.0, "rough_diffuse_subsurface": 1.0, "metal": 1.0, "clearcoat": 1.25}[name]
    # lobe weights are not normalised (disney_material.hlsli:230-268); clearcoat adds 0.25 on top of diffuse
    assert abs(ipdf - w_total) < 0.03 * w_total, ipdf
    assert alb < 1.02 * w_total, alb


# ---------------------------------------------------------------------------------------------
# T1/T2 traversal contract: the oracle's BVH never changes the brute-force answer
# ---------------------------------------------------------------------------------------------
def soup_scene(seed, n_meshes=5, tris=300):
    rng = np.random.RandomState(seed)
    b = SceneBuilder("soup")
    m = b.add_material((0.5, 0.5, 0.5))
    for k in range(n_meshes):
        c = rng.uniform(-1, 1, (tris, 1, 3))
        p = (c + rng.normal(scale=0.15, size=(tris, 3, 3))).reshape(-1, 3)
        tri = np.arange(tris * 3).reshape(-1, 3)
        mesh = b.add_mesh(p, None, None, tri, index_stride=2 if k % 2 else 4)
        b.add_instance(mesh, m, None if k == 0 else translate(rng.uniform(-1, 1, 3)) @ rotate_y(rng.uniform(0, 6)) @ scale(rng.uniform(0.5, 1.5, 3)))
        if k == 1:  # a second instance of the same mesh
            b.add_instance(mesh, m, translate((2.0, 0.3, -1.0)) @ rotate_y(1.0))
    return b.build()


def random_rays(n, seed, lo=-3.0, hi=3.0):
    rng = np.random.RandomState(seed)
    rays = np.zeros(n, wire.Ray)
    rays["origin"] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmax"] = np.inf
    return rays


def test_bvh_equals_brute_force():
    sc = soup_scene(7)
    o = orc.OracleScene(sc)
    rays = random_rays(20000, 8)
    a, _ = o.trace(rays)
    b, _ = o.trace(rays, brute=True)
    for f in ("instance_primitive_index", "t", "b1", "b2"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
    assert (a["instance_primitive_index"] != wire.MISS).mean() > 0.15
    rays["tmax"] = 1.5
    a, _ = o.trace(rays, any_hit=True)
    b, _ = o.trace(rays, any_hit=True, brute=True)
    assert np.array_equal(a["instance_primitive_index"], b["instance_primitive_index"])


def test_hits_reconstruct_the_surface_point():
    """P = v0 + b1 (v1-v0) + b2 (v2-v0) (shading_data.hlsli:69-72) lies on the ray at t."""
    sc, _ = scenes.cornell_box()
    o = orc.OracleScene(sc)
    rays = random_rays(5000, 9, -0.9, 0.9)
    h, _ = o.trace(rays)
    hit = h["instance_primitive_index"] != wire.MISS
    assert hit.mean() > 0.7  # the box is open towards +z
    sd = o.shading_data(h["instance_primitive_index"][hit], np.stack([h["b1"], h["b2"]], 1)[hit])
    p = rays["origin"][hit] + rays["direction"][hit] * h["t"][hit, None]
    assert np.abs(sd["position"] - p).max() < 2e-5


def test_watertight_shared_edges():
    """Rays aimed exactly at the shared diagonal and at shared vertices of a tessellated wall never leak."""
    b = SceneBuilder()
    m = b.add_material((0.5, 0.5, 0.5))
    part = scenes.grid_surface(lambda U, V: np.stack([U * 2 - 1, V * 2 - 1, 0 * U], -1), 16, 16)
    b.add_instance(b.add_mesh(*part), m)
    sc = b.build()
    o = orc.OracleScene(sc)
    rng = np.random.RandomState(10)
    n = 20000
    # targets on grid vertices, on horizontal/vertical/diagonal edges
    gx = rng.randint(0, 17, n) / 8.0 - 1
    gy = rng.randint(0, 17, n) / 8.0 - 1
    t = rng.uniform(0, 1, n)
    kind = rng.randint(0, 3, n)
    tx = np.where(kind == 0, gx, np.clip(gx + t / 8.0, -1, 1))
    ty = np.where(kind == 1, gy, np.where(kind == 2, np.clip(gy + t / 8.0, -1, 1), gy))
    target = np.stack([tx, ty, np.zeros(n)], 1)
    origin = target + np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(0.5, 3, n)], 1)
    rays = np.zeros(n, wire.Ray)
    rays["origin"] = origin
    d = target - origin
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmax"] = np.inf
    inside = (np.abs(tx) < 0.999) & (np.abs(ty) < 0.999)
    h, _ = o.trace(rays)
    assert (h["instance_primitive_index"][inside] != wire.MISS).all()


# ---------------------------------------------------------------------------------------------
# estimator: direct illumination against the closed-form form factor; N-sample semantics
# ---------------------------------------------------------------------------------------------
def plane_and_light(rho=0.6, Le=5.0, a=0.8, h=1.0):
    b = SceneBuilder("plane_light")
    white = b.add_material((rho, rho, rho))
    light = b.add_emitter((Le, Le, Le))
    S = 50.0
    floor = scenes._quad((-S, 0, S), (S, 0, S), (S, 0, -S), (-S, 0, -S), (0, 1, 0))
    b.add_instance(b.add_mesh(*floor), white)
    lq = scenes._quad((-a, h, -a), (a, h, -a), (a, h, a), (-a, h, a), (0, -1, 0))
    b.add_instance(b.add_mesh(*lq), light)
    return b.build()


def form_factor_point_to_coaxial_square(a, h):
    A = a / h

    def corner(A, B):
        return (A / np.sqrt(1 + A * A) * np.arctan(B / np.sqrt(1 + A * A)) + B / np.sqrt(1 + B * B) * np.arctan(A / np.sqrt(1 + B * B))) / (2 * np.pi)

    return 4 * corner(A, A)


@pytest.mark.parametrize("flags", ["default", "~nee", "~samplebsdfs", "~mis", "~defershadowrays", "presamplelights", "neereservoirs", "neereservoirs ~samplebsdfs"])
def test_direct_light_matches_form_factor(flags):
    """L = rho * Le * F at the floor point under the light, for every estimator combination
    (NEE + BSDF sampling with MIS, either alone, MIS off = 0.5/0.5, shadow rays not deferred)."""
    rho, Le, a, h = 0.6, 5.0, 0.8, 1.0
    sc = plane_and_light(rho, Le, a, h)
    o = orc.OracleScene(sc)
    fr = camera.Frame(8, 8, np.radians(0.05), (0.3, 0.5, 0.0), (0.0, 0.0, 0.0), up=(0, 0, 1))  # looks at the origin, below the light
    pc = wire.default_push_constants(8, 8, sc.light_count)
    pc.gMaxDiffuseVertices = 1  # direct light only
    f = wire.DEFAULT_SAMPLING_FLAGS
    if flags == "presamplelights":  # NEE from a small presampled tile per seed (bdpt.hlsl:84-99): unbiased over the seeds
        f |= wire.flag_mask("ePresampleLights")
        pc.gLightPresampleTileSize, pc.gLightPresampleTileCount = 64, 4
    elif flags.startswith("neereservoirs"):  # RIS over gReservoirM candidates (path.hlsli:368-486); 0.5 / 0.5 with BSDF sampling
        f |= wire.flag_mask("eNEEReservoirs")
        if flags.endswith("~samplebsdfs"):
            f &= ~wire.flag_mask("eSampleBSDFs")
    elif flags != "default":
        f &= ~wire.flag_mask({"~nee": "eNEE", "~samplebsdfs": "eSampleBSDFs", "~mis": "eMIS", "~defershadowrays": "eDeferShadowRays"}[flags])
    out = o.render(fr, pc, f, 0, 2048, aovs=False)
    got = out["radiance"][..., 0].astype(np.float64).mean()
    want = rho * Le * form_factor_point_to_coaxial_square(a, h)
    assert out["radiance"][..., 3].min() == 2048
    assert abs(got / want - 1) < 0.02, (got, want)


def test_directly_visible_emitter_has_weight_one():
    """path.hlsli:869-870: emission at the first hit is added unweighted; emitters do not scatter."""
    sc, cam = scenes.furnace_box(emission=3.5)
    o = orc.OracleScene(sc)
    fr = camera.Frame(16, 16, cam["fovy"], cam["eye"], cam["target"])
    out = o.render(fr, wire.default_push_constants(16, 16, sc.light_count), wire.DEFAULT_SAMPLING_FLAGS, 0, 3)
    assert np.array_equal(out["radiance"][..., :3], np.full((16, 16, 3), 3.5, np.float32))
    assert out["ray_count"][0] == 16 * 16 * 3  # one ray per pixel-sample, no NEE from an emitter


def test_n_samples_is_the_running_mean_of_seeds():
    """A1: sample i <=> seed i; result = temporal_accumulation's running mean over per-seed frames."""
    sc, cam = scenes.cornell_box()
    o = orc.OracleScene(sc)
    fr = camera.Frame(32, 32, cam["fovy"], cam["eye"], cam["target"])
    pc = wire.default_push_constants(32, 32, sc.light_count)
    per_seed = [o.render(fr, pc, seed_begin=s, seed_count=1, aovs=False)["radiance"] for s in range(3, 8)]
    acc = per_seed[0][..., :3].copy()
    for n, cur in enumerate(per_seed[1:], start=2):
        alpha = np.float32(1.0) / np.float32(n)
        acc = acc + alpha * (cur[..., :3] - acc)
    got = o.render(fr, pc, seed_begin=3, seed_count=5, aovs=False)["radiance"]
    assert np.array_equal(got[..., :3], acc)
    assert (got[..., 3] == 5).all()
    # different seeds give different noise, the same seed the same frame
    assert not np.array_equal(per_seed[0], per_seed[1])
    assert np.array_equal(per_seed[0], o.render(fr, pc, seed_begin=3, seed_count=1, aovs=False)["radiance"])


def test_ray_budget_at_default_flags():
    """<= 3 closest-hit and <= 2 shadow rays per pixel-sample with the default limits."""
    sc, cam = scenes.cornell_box()
    o = orc.OracleScene(sc)
    fr = camera.Frame(64, 64, cam["fovy"], cam["eye"], cam["target"])
    out = o.render(fr, wire.default_push_constants(64, 64, sc.light_count), aovs=False)
    total, path = int(out["ray_count"][0]), int(out["ray_count"][1])
    assert 64 * 64 <= path <= 3 * 64 * 64
    assert total - path <= 2 * 64 * 64


def test_cornell_golden_fixture():
    """configs[0]: Cornell box 256x256, seed 0 — the committed fixture (tests/golden/make_golden.py)."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cornell_256_seed0.npz"))
    sc, cam = scenes.cornell_box()
    o = orc.OracleScene(sc)
    fr = camera.Frame(256, 256, cam["fovy"], cam["eye"], cam["target"])
    out = o.render(fr, wire.default_push_constants(256, 256, sc.light_count))
    assert np.array_equal(out["radiance"].view(np.uint32), g["radiance"].view(np.uint32))
    assert np.array_equal(out["visibility"]["instance_primitive_index"], g["instance_primitive_index"])
    assert np.array_equal(out["ray_count"], g["ray_count"])


def test_estimator_golden_fixture():
    """tests/golden/estimators.npz (make_estimator_golden.py): the estimators whose upstream result depends on scheduling,
    under the orders DESIGN.md 5 defines, frozen as committed frames — the oracle must keep reproducing them bit for bit."""
    import importlib.util
    import os

    here = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_estimator_golden", os.path.join(here, "make_estimator_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    g = np.load(os.path.join(here, "estimators.npz"))
    sc, cam = scenes.cornell_box()
    o = orc.OracleScene(sc)
    fr = camera.Frame(mk.W, mk.H, cam["fovy"], cam["eye"], cam["target"])
    for name, (names, overrides) in mk.CASES.items():
        out = o.render(fr, mk.push_constants(sc, overrides), mk.flags_of(names), 3, mk.SEEDS)
        assert np.array_equal(out["radiance"].view(np.uint32), g[name + "_radiance"].view(np.uint32)), name
        assert np.array_equal(out["ray_count"], g[name + "_ray_count"]), name


# ---------------------------------------------------------------------------------------------
# N2: software texture sampler (repeat + trilinear over a box-filtered mip chain), ray cones, normal maps
# ---------------------------------------------------------------------------------------------
def test_texture_sampler_properties():
    b = SceneBuilder()
    rng = np.random.RandomState(11)
    img = rng.uniform(0, 1, (8, 16, 4)).astype(np.float32)  # H = 8, W = 16
    i0 = b.add_image(img)
    flat = b.add_image(np.full((32, 32, 4), 0.37, np.float32))
    m = b.add_material((1, 1, 1))
    b.set_material_images(m, base_color_image=i0, params_image=flat)  # indices follow first use: i0 -> 0, flat -> 1
    b.add_instance(b.add_mesh(*scenes._quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1))), m)
    sc = b.build()
    assert len(sc.images) == 2
    i0, flat = 0, 1
    o = orc.OracleScene(sc)
    # texel centres return the texel (level 0: uv_screen_size = 0 means "no footprint", image_value.h:85)
    ys, xs = np.mgrid[0:8, 0:16]
    q = np.stack([(xs + 0.5) / 16, (ys + 0.5) / 8, np.zeros_like(xs, float)], -1).reshape(-1, 3)
    assert np.array_equal(o.sample_image(i0, q), img.reshape(-1, 4))
    # repeat addressing
    assert np.array_equal(o.sample_image(i0, q + [1.0, -2.0, 0.0]), o.sample_image(i0, q))
    # halfway between two texels = their mean
    mid = o.sample_image(i0, [[1.0 / 16, 0.5 / 8, 0.0]])[0]
    assert np.allclose(mid, 0.5 * (img[0, 0] + img[0, 1]), atol=1e-7)
    # lod = log2(uv_screen_size * max(w, h)): size 2/16 -> level 1 = 2x2 box filter of level 0
    lvl1 = img.reshape(4, 2, 8, 2, 4).mean(axis=(1, 3))
    q1 = np.stack([(np.arange(8) + 0.5) / 8, np.full(8, 0.5 / 4), np.full(8, 2.0 / 16)], -1)
    assert np.allclose(o.sample_image(i0, q1), lvl1[0], atol=1e-6)
    # without ray cones the footprint is ignored
    assert np.array_equal(o.sample_image(i0, q1, ray_cones=False), o.sample_image(i0, q1 * [1, 1, 0]))
    # a constant image is constant at every level and position
    qq = np.concatenate([rng.uniform(-3, 3, (200, 2)), rng.uniform(0, 2, (200, 1))], 1)
    assert np.allclose(o.sample_image(flat, qq), 0.37, atol=1e-6)


def test_ray_cones_and_normal_maps_change_the_image():
    sc, cam = scenes.textured_box()
    o = orc.OracleScene(sc)
    fr = camera.Frame(64, 48, cam["fovy"], cam["eye"], cam["target"])
    pc = wire.default_push_constants(64, 48, sc.light_count)
    base = o.render(fr, pc, wire.DEFAULT_SAMPLING_FLAGS, 0, 1)
    no_cones = o.render(fr, pc, wire.DEFAULT_SAMPLING_FLAGS & ~wire.flag_mask("eRayCones"), 0, 1)
    no_bump = o.render(fr, pc, wire.DEFAULT_SAMPLING_FLAGS & ~wire.flag_mask("eNormalMaps"), 0, 1)
    assert not np.array_equal(base["albedo"], no_cones["albedo"])  # mip level 0 instead of the footprint's level
    assert not np.array_equal(base["visibility"]["packed_normal"], no_bump["visibility"]["packed_normal"])
    assert np.array_equal(base["visibility"]["instance_primitive_index"], no_cones["visibility"]["instance_primitive_index"])


# ---------------------------------------------------------------------------------------------
# sphere instances, sphere lights, environment maps
# ---------------------------------------------------------------------------------------------
def test_detmath_inverse_trig_against_libm():
    rng = np.random.RandomState(11)
    y = rng.uniform(-3, 3, 200000).astype(np.float32)
    x = rng.uniform(-3, 3, 200000).astype(np.float32)
    assert np.abs(orc.atan2(y, x) - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 6e-7
    # stable_atan2's x == 0 rule (common.h:134-136) and the axes
    assert orc.atan2(np.float32([0, 1, -1, 0, 0]), np.float32([0, 0, 0, 1, -1])).tolist() == [0.0, np.float32(np.pi / 2), -np.float32(np.pi / 2), 0.0, np.float32(np.pi)]
    c = np.concatenate([rng.uniform(-1, 1, 200000), [-1, 1, 0, 0.5, -0.5]]).astype(np.float32)
    assert np.abs(orc.acos(c) - np.arccos(c.astype(np.float64))).max() < 5e-7
    assert np.abs(orc.asin(c) - np.arcsin(c.astype(np.float64))).max() < 5e-7


def sphere_soup(seed, n_spheres=12):
    sc0 = soup_scene(seed, n_meshes=3, tris=150)
    b = sc0.builder
    rng = np.random.RandomState(seed + 100)
    m = b.add_material((0.5, 0.5, 0.5))
    for _ in range(n_spheres):
        b.add_sphere(m, rng.uniform(0.1, 0.8), translate(rng.uniform(-2, 2, 3)))
    return b.build()


def test_spheres_in_the_trace_contract():
    sc = sphere_soup(3)
    o = orc.OracleScene(sc)
    rays = random_rays(6000, 5)
    h_bvh, _ = o.trace(rays, any_hit=False, brute=False)
    h_brute, _ = o.trace(rays, any_hit=False, brute=True)
    for f in ("instance_primitive_index", "t", "b1", "b2"):
        assert np.array_equal(h_bvh[f].view(np.uint32), h_brute[f].view(np.uint32)), f
    a_bvh, _ = o.trace(rays, any_hit=True, brute=False)
    a_brute, _ = o.trace(rays, any_hit=True, brute=True)
    assert np.array_equal(a_bvh["instance_primitive_index"], a_brute["instance_primitive_index"])
    ip = h_bvh["instance_primitive_index"]
    on_sphere = (ip != wire.MISS) & ((ip >> 16) == 0xFFFF)
    assert on_sphere.sum() > 300
    # a sphere hit lies on its sphere: |o + t d - c| = r
    inst = ip[on_sphere] & 0xFFFF
    centre = sc.transforms["m"][inst][:, :, 3]
    radius = sc.instances["packed"][inst, 2].astype(np.uint32).view(np.float32)
    p = rays["origin"][on_sphere] + rays["direction"][on_sphere] * h_bvh["t"][on_sphere, None]
    assert np.abs(np.linalg.norm(p - centre, axis=1) - radius).max() < 2e-5
    # rays that start inside a sphere leave through the far root
    inside = np.zeros(1, wire.Ray)
    inside["origin"], inside["direction"], inside["tmax"] = centre[0], (0, 0, 1), np.inf
    only = SceneBuilder("one")
    only.add_sphere(only.add_material((1, 1, 1)), float(radius[0]), translate(centre[0]))
    h, _ = orc.OracleScene(only.build()).trace(inside)
    assert abs(h["t"][0] - radius[0]) < 1e-6


def plane_and_sphere_light(rho, Le, r, d):
    b = SceneBuilder("plane_sphere_light")
    white = b.add_material((rho, rho, rho))
    light = b.add_emitter((Le, Le, Le))
    S = 50.0
    b.add_instance(b.add_mesh(*scenes._quad((-S, 0, S), (S, 0, S), (S, 0, -S), (-S, 0, -S), (0, 1, 0))), white)
    b.add_sphere(light, r, translate((0.0, d, 0.0)))
    return b.build()


@pytest.mark.parametrize("flags", ["default", "~nee", "~samplebsdfs", "uniformspheresampling", "~defershadowrays"])
def test_sphere_light_matches_the_closed_form(flags):
    """A sphere light of radius r whose centre is d above a diffuse floor point: E = pi Le (r/d)^2, L = rho Le (r/d)^2,
    with cone sampling (default), uniform area sampling, and either estimator alone."""
    rho, Le, r, d = 0.7, 9.0, 0.4, 1.5
    sc = plane_and_sphere_light(rho, Le, r, d)
    o = orc.OracleScene(sc)
    # seen from the same steep direction as the form-factor test: the Disney diffuse lobe is Lambertian only to ~1 % there
    fr = camera.Frame(8, 8, np.radians(0.05), (0.3, 0.5, 0.0), (0.0, 0.0, 0.0), up=(0, 0, 1))
    pc = wire.default_push_constants(8, 8, sc.light_count)
    pc.gMaxDiffuseVertices = 1
    f = wire.DEFAULT_SAMPLING_FLAGS
    if flags == "uniformspheresampling":
        f |= wire.flag_mask("eUniformSphereSampling")
    elif flags != "default":
        f &= ~wire.flag_mask({"~nee": "eNEE", "~samplebsdfs": "eSampleBSDFs", "~defershadowrays": "eDeferShadowRays"}[flags])
    out = o.render(fr, pc, f, 0, 2048, aovs=False)
    got = out["radiance"][..., 0].astype(np.float64).mean()
    want = rho * Le * (r / d) ** 2
    assert abs(got / want - 1) < 0.02, (got, want)


def sphere_under_sky(rho, value, image):
    b = SceneBuilder("furnace_env")
    b.add_sphere(b.add_material((rho, rho, rho)), 1.0, translate((0, 0, 0)))
    b.set_environment(value, None if image is None else b.add_image(image))
    return b.build()


def test_environment_estimators_agree():
    """A diffuse sphere under a uniform environment L: the background shows L exactly, and the three estimators of the
    lat-long image path (dist2d light sampling + MIS, light sampling alone, BSDF sampling alone) agree on the sphere,
    as does BSDF sampling under the image-less environment of the same radiance. (The Disney diffuse lobe is not
    Lambertian, so the value itself is not rho L.) The image-less environment is not checked with light sampling: its
    direction mapping is biased upstream (sample_uniform_sphere's angles are fed to spherical_uv_to_cartesian as if
    they were uv, environment.h:58-62), and that bias is restated, not fixed."""
    rho, L = 0.6, 2.5
    fr = camera.Frame(24, 24, np.radians(35.0), (0.0, 0.0, 5.0), (0.0, 0.0, 0.0))
    D = wire.DEFAULT_SAMPLING_FLAGS
    means = {}
    for kind, image, f in (
        ("image mis", True, D),
        ("image light sampling", True, D & ~wire.flag_mask("eSampleBSDFs")),
        ("image bsdf sampling", True, D & ~wire.flag_mask("eNEE")),
        ("constant bsdf sampling", False, D & ~wire.flag_mask("eNEE")),
        # eSampleEnvironmentMapDirectly: sample_texel's descent through the mip chain instead of the dist2d tables
        ("image direct mis", True, D | wire.flag_mask("eSampleEnvironmentMapDirectly")),
        ("image direct light sampling", True, (D | wire.flag_mask("eSampleEnvironmentMapDirectly")) & ~wire.flag_mask("eSampleBSDFs")),
    ):
        sc = sphere_under_sky(rho, (L, L, L), np.ones((8, 16, 4), np.float32) if image else None)
        pc = wire.default_push_constants(24, 24, sc.light_count)
        pc.gEnvironmentMaterialAddress = sc.environment_address
        pc.gMaxDiffuseVertices = 1
        out = orc.OracleScene(sc).render(fr, pc, f, 0, 512)
        on = out["visibility"]["instance_primitive_index"] != wire.MISS
        assert 100 < on.sum() < 400
        rad = out["radiance"][..., 0].astype(np.float64)
        assert np.allclose(rad[~on], L, rtol=1e-6)
        means[kind] = rad[on].mean()
    ref = means["image bsdf sampling"]
    assert 0.85 * rho * L < ref < 1.05 * rho * L
    for kind, m in means.items():
        assert abs(m / ref - 1) < 0.01, means


def test_build_distributions_follows_dist2():
    img = scenes.sky_image(32, 16)
    pdf_m, pdf_r, cdf_m, cdf_r = build_distributions(img)
    H, W = 16, 32
    assert pdf_m.shape == (H,) and pdf_r.shape == (H * W,) and cdf_m.shape == (H + 1,) and cdf_r.shape == (H * (W + 1),)
    cr = cdf_r.reshape(H, W + 1)
    assert cdf_m[0] == 0 and cdf_m[-1] == 1 and np.all(np.diff(cdf_m) >= 0)
    assert np.all(cr[:, 0] == 0) and np.all(cr[:, -1] == 1) and np.all(np.diff(cr, axis=1) >= -1e-7)
    assert abs(pdf_m.sum() - 1) < 1e-5 and np.abs(pdf_r.reshape(H, W).sum(1) - 1).max() < 1e-5
    # the sun's row carries most of the probability
    assert pdf_m.argmax() == int(0.25 * H)
    # an all-black image falls back to the uniform tables (dist2.h:112-119,142-149)
    z = build_distributions(np.zeros((4, 8, 4), np.float32))
    assert np.allclose(z[0], 0.25) and np.allclose(z[1], 0.125) and np.allclose(z[2], [0, 0.25, 0.5, 0.75, 1])


def test_alpha_masks_in_the_trace_contract():
    """gAlphaTest (intersection.hlsli:117-131): a masked triangle is hit only where its coverage image is >= 0.75 at the
    hit's uv; the acceleration structure does not change that; without the flag the card is solid."""
    b = SceneBuilder("card")
    m = b.add_material((1, 1, 1))
    mask = np.zeros((8, 8), np.float32)
    mask[:, :4] = 1.0  # left half of uv space covered
    b.set_material_alpha_mask(m, b.add_image1(mask))
    p, n, _, tri = scenes._quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), (0, 0, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    b.add_instance(b.add_mesh(p, n, uv, tri), m)
    o = orc.OracleScene(b.build())
    rays = np.zeros(2, wire.Ray)
    rays["origin"] = [(-0.5, 0.1, 1), (0.5, 0.1, 1)]
    rays["direction"] = (0, 0, -1)
    rays["tmax"] = np.inf
    h, _ = o.trace(rays)
    assert (h["instance_primitive_index"] != wire.MISS).all()  # flag off: solid
    h, _ = o.trace(rays, alpha_test=True)
    assert h["instance_primitive_index"][0] != wire.MISS and h["instance_primitive_index"][1] == wire.MISS
    # bilinear coverage: the edge between covered and empty texels passes 0.75 a quarter texel into the covered side
    xs = np.linspace(-0.2, 0.2, 81).astype(np.float32)
    edge = np.zeros(81, wire.Ray)
    edge["origin"] = np.stack([xs, np.full(81, 0.1, np.float32), np.ones(81, np.float32)], 1)
    edge["direction"] = (0, 0, -1)
    edge["tmax"] = np.inf
    h, _ = o.trace(edge, alpha_test=True)
    hit = h["instance_primitive_index"] != wire.MISS
    u_edge = (xs[hit].max() + 1) / 2  # uv.x of the last hit
    assert abs(u_edge - (0.5 - 0.25 / 8)) < 0.01
    # foliage: BVH == brute force, closest and any hit, with flipped uvs too
    sc, _ = scenes.foliage()
    o = orc.OracleScene(sc)
    rays = random_rays(20000, 2, -2.0, 2.5)
    for flip in (False, True):
        a, _ = o.trace(rays, alpha_test=True, flip_uvs=flip)
        bb, _ = o.trace(rays, brute=True, alpha_test=True, flip_uvs=flip)
        for f in ("instance_primitive_index", "t", "b1", "b2"):
            assert np.array_equal(a[f].view(np.uint32), bb[f].view(np.uint32)), f
    solid, _ = o.trace(rays)
    assert (solid["instance_primitive_index"] != a["instance_primitive_index"]).mean() > 0.01
    a, _ = o.trace(rays, any_hit=True, alpha_test=True)
    bb, _ = o.trace(rays, any_hit=True, brute=True, alpha_test=True)
    assert np.array_equal(a["instance_primitive_index"], bb["instance_primitive_index"])


def test_shading_normal_shadow_fix_only_touches_bumped_shading():
    """eShadingNormalShadowFix (path.hlsli:84-86): G = min(1, |ngdotout / (ndotout ngdotns)|) -> -G^3 + G^2 + G is 1 where the
    shading normal equals the geometry normal (flat-shaded Cornell box), and darkens terminators where a normal map
    bends it (textured box)."""
    f_on = wire.DEFAULT_SAMPLING_FLAGS | wire.flag_mask("eShadingNormalShadowFix")
    sc, cam = scenes.cornell_box()
    fr = camera.Frame(48, 48, cam["fovy"], cam["eye"], cam["target"])
    pc = wire.default_push_constants(48, 48, sc.light_count)
    o = orc.OracleScene(sc)
    a = o.render(fr, pc, wire.DEFAULT_SAMPLING_FLAGS, 0, 2)["radiance"]
    b = o.render(fr, pc, f_on, 0, 2)["radiance"]
    assert np.abs(a - b).max() < 2e-3 * a.max()  # fp16-packed normals make ngdotns differ from 1 by ~1e-3
    sc, cam = scenes.textured_box()
    fr = camera.Frame(64, 48, cam["fovy"], cam["eye"], cam["target"])
    pc = wire.default_push_constants(64, 48, sc.light_count)
    o = orc.OracleScene(sc)
    a = o.render(fr, pc, wire.DEFAULT_SAMPLING_FLAGS, 0, 4)["radiance"][..., :3]
    b = o.render(fr, pc, f_on, 0, 4)["radiance"][..., :3]
    assert np.isfinite(b).all() and np.abs(a - b).mean() > 1e-3 * a.mean()



def test_light_tracing_is_unbiased_against_path_tracing():
    """eConnectToViews with BSDF sampling (NEE off): light tracing and the view paths share every transport path through
    the MIS weights of path.hlsli:600-605,870-880, so the image converges to the path tracer's; without MIS the uniform
    path_weight does the same with NEE on. (With MIS AND NEE on the three strategies' weights do not sum to one upstream —
    eval_emission and connect_view only weigh against each other — and that is restated, not fixed.)"""
    sc, cam = scenes.cornell_box()
    o = orc.OracleScene(sc)
    W = H = 40
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    pc = wire.default_push_constants(W, H, sc.light_count)
    D = wire.DEFAULT_SAMPLING_FLAGS
    ref = o.render(fr, pc, D, 0, 512, aovs=False)["radiance"][..., :3].astype(np.float64)
    for f in ((D | wire.flag_mask("eConnectToViews")) & ~wire.flag_mask("eNEE"), (D | wire.flag_mask("eConnectToViews")) & ~wire.flag_mask("eMIS")):
        out = o.render(fr, pc, f, 0, 512, aovs=False)
        img = out["radiance"][..., :3].astype(np.float64)
        assert abs(img.mean() / ref.mean() - 1) < 0.02  # 40 x 40 x 512 samples: ~1 % noise on the mean
        assert np.sqrt(((img - ref) ** 2).sum() / (ref**2).sum()) < 0.04
        assert out["ray_count"][0] > 1.5 * W * H * 512  # the light paths' rays are counted too


def test_light_path_connections_against_path_tracing():
    """eConnectToLightPaths (no vertex cache) with NEE and MIS: the dVC recursion of path.hlsli:29-36 weighs the view
    paths, NEE and the subpath connections against each other. Upstream's recursion also counts the strategy of a light
    path hitting the pinhole camera (bsdf_pdf = 1 at the camera, p0_fwd = 1), which is never sampled, so about 1 % of
    the energy is lost; that is restated. The image stays within 3 % of the path tracer's and is less noisy."""
    sc, cam = scenes.cornell_box()
    o = orc.OracleScene(sc)
    W = H = 40
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    pc = wire.default_push_constants(W, H, sc.light_count)
    pc.gMaxDiffuseVertices = 3
    pc.gMaxPathVertices = 6
    D = wire.DEFAULT_SAMPLING_FLAGS
    L = wire.flag_mask("eConnectToLightPaths")
    ref = o.render(fr, pc, D, 0, 1024, aovs=False)["radiance"][..., :3].astype(np.float64)
    for f in (D | L, D | L | wire.flag_mask("eConnectToViews")):
        out = o.render(fr, pc, f, 0, 256, aovs=False)
        img = out["radiance"][..., :3].astype(np.float64)
        assert 0.97 < img.mean() / ref.mean() < 1.005
        assert np.sqrt(((img - ref) ** 2).sum() / (ref**2).sum()) < 0.02
    # the stored vertices matter: without the light pass's vertices (gMaxPathVertices = 2 stores none) nothing connects
    few = o.render(fr, pc, D | L, 0, 4, aovs=False)
    base = o.render(fr, pc, D, 0, 4, aovs=False)
    assert few["ray_count"][0] > 1.5 * base["ray_count"][0]


def test_sample_texel_agrees_with_the_distribution_tables():
    """A sky with a sun (most of the energy in a few texels): light sampling alone through sample_texel / sample_texel_pdf
    (bdpt_util.hlsli:85-180) and through the dist2d tables are two unbiased estimators of the same integral."""
    sc = sphere_under_sky(0.6, (1.0, 1.0, 1.0), scenes.sky_image(64, 32))
    fr = camera.Frame(24, 24, np.radians(35.0), (0.0, 0.0, 5.0), (0.0, 0.0, 0.0))
    pc = wire.default_push_constants(24, 24, sc.light_count)
    pc.gEnvironmentMaterialAddress = sc.environment_address
    pc.gMaxDiffuseVertices = 1
    f = wire.DEFAULT_SAMPLING_FLAGS & ~wire.flag_mask("eSampleBSDFs")
    means = []
    for flags in (f, f | wire.flag_mask("eSampleEnvironmentMapDirectly")):
        out = orc.OracleScene(sc).render(fr, pc, flags, 0, 1024)
        on = out["visibility"]["instance_primitive_index"] != wire.MISS
        means.append(out["radiance"][..., :3].astype(np.float64)[on].mean())
    assert means[0] > 0.05 and abs(means[1] / means[0] - 1) < 0.02, means


def test_nanovdb_reader_against_the_reference_accessor():
    """tests/golden/fog_sphere.npz holds a NanoVDB 32.3 float grid made with the reference's vendored NanoVDB and 6000 values,
    the root bounding box / maximum and the four map functions read from it by the reference's own PNanoVDB.h
    (oracle/nvdb_ref.cpp). The oracle's restated reader returns the same bits."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fog_sphere.npz"))
    bmin, bmax, root_max, values, maps = orc.nvdb_probe(g["grid"], g["coords"], g["map"][:, 0, :])
    assert np.array_equal(bmin, g["bbox_min"]) and np.array_equal(bmax, g["bbox_max"]) and root_max == float(g["root_max"])
    assert np.array_equal(values.view(np.uint32), g["values"].view(np.uint32))
    assert (values > 0).sum() > 500 and (values == 0).sum() > 1000  # inside the fog, and background / inactive voxels
    assert np.array_equal(maps.view(np.uint32), g["map"][:, 1:, :].view(np.uint32))


def _fog_grid():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "fog_sphere.npz"))["grid"]


def test_media_furnace():
    """A cloud with albedo 1 inside a closed cube whose walls all emit L: with BSDF (phase-function) sampling every pixel
    is exactly L whatever the density and anisotropy — delta tracking, the spectral pdf bookkeeping (T_dir_pdf) and the
    phase-function sampling conserve energy. An absorbing cloud is darker. With NEE the reference loses energy: a visibility
    ray from a vertex inside the medium gets no distance epsilon (DirectLightSample::setup, path.hlsli:207-212) and hits
    the emitter it aims at about every other time; that, and delta tracking ending at its first null collision
    (medium.hlsli:104-110), are restated, not fixed."""
    for density, albedo, aniso in ((1.0, 1.0, 0.0), (2.0, 1.0, 0.6), (2.0, 0.5, -0.4)):
        sc0, _ = scenes.furnace_box(albedo=0.5, emission=1.0)
        b = sc0.builder
        b.add_medium(b.add_volume(_fog_grid()), density_scale=(density,) * 3, albedo_scale=(albedo,) * 3, anisotropy=aniso)
        sc = b.build()
        o = orc.OracleScene(sc)
        fr = camera.Frame(24, 24, np.radians(40), (0, 0, 0.95), (0, 0, -1))
        pc = wire.default_push_constants(24, 24, sc.light_count)
        pc.gMaxNullCollisions = 64
        pc.gMaxDiffuseVertices, pc.gMaxPathVertices, pc.gMinPathVertices = 40, 50, 60
        out = o.render(fr, pc, wire.DEFAULT_SAMPLING_FLAGS & ~wire.flag_mask("eNEE"), 0, 16)
        r = out["radiance"][..., :3]
        in_fog = (out["visibility"]["instance_primitive_index"] & 0xFFFF) == sc.instances.shape[0] - 1
        assert in_fog.mean() > 0.5
        if albedo == 1.0:
            assert np.abs(r - 1).max() < 1e-5
        else:
            assert r[in_fog].mean() < 0.9
        nee = o.render(fr, pc, wire.DEFAULT_SAMPLING_FLAGS, 0, 16)["radiance"][..., :3]
        if albedo == 1.0:
            assert 0.7 < nee.mean() < 0.95  # the reference's NEE from inside a medium, see above


def test_nanovdb_fixture_is_what_the_reference_build_makes(tmp_path):
    """Where oracle/_ref/nvdb_ref exists (built from the reference's vendored NanoVDB by `make -C oracle ref`), it
    reproduces the committed fixture byte for byte."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "nvdb_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/nvdb_ref is only built where /root/reference exists")
    grid, probe = str(tmp_path / "fog.nvdb"), str(tmp_path / "probe.bin")
    subprocess.check_call([exe, "10", "0.08", "3", grid, probe], stdout=subprocess.DEVNULL)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fog_sphere.npz"))
    assert np.array_equal(np.fromfile(grid, dtype=np.uint8), g["grid"])
    raw = np.fromfile(probe, dtype=np.int32)
    ni, nf, n, m = raw[:4]
    floats = raw[4 + ni : 4 + ni + nf].view(np.float32)
    assert np.array_equal(floats[1 : 1 + n].view(np.uint32), g["values"].view(np.uint32))


def test_oracle_bvh_never_loses_a_hit_on_awkward_rays():
    """The oracle's own acceleration structure against its brute force on rays a renderer never makes but a caller of
    sthip_trace_rays may: tiny / huge unnormalised directions, origins on flat meshes, tmin > 0 (tools/fuzz_parity.py rays
    found that unpadded boxes lose hits whose rounded t lies a few ulp off a flat, axis-aligned mesh)."""
    for make in (scenes.foliage, scenes.cornell_box, scenes.spheres_room):
        sc, _ = make()
        o = orc.OracleScene(sc)
        rng = np.random.default_rng(3)
        n = 60000
        lo, hi = sc.vertices["position"].min(0), sc.vertices["position"].max(0)
        rays = np.zeros(n, wire.Ray)
        rays["origin"] = rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), (n, 3)).astype(np.float32)
        rays["origin"][::7, 1] = lo[1]  # on the ground plane
        rays["direction"] = (rng.normal(size=(n, 3)) * rng.choice([1e-6, 1.0, 1e4], n)[:, None]).astype(np.float32)
        rays["tmin"] = (rng.uniform(0, 1, n) * rng.integers(0, 2, n)).astype(np.float32)
        rays["tmax"] = np.inf
        a, _ = o.trace(rays, brute=True)
        b, _ = o.trace(rays)
        for f in ("t", "b1", "b2", "instance_primitive_index"):
            assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), (sc.name, f)
        ha = o.trace(rays, any_hit=True, brute=True)[0]["instance_primitive_index"] != wire.MISS
        hb = o.trace(rays, any_hit=True)[0]["instance_primitive_index"] != wire.MISS
        assert np.array_equal(ha, hb), sc.name


def test_far_origin_hits_of_the_contract_are_pinned():
    """A known hole of the hit CONTRACT ("spurious contract hits from far origins"), pinned as it is: the contract's
    triangle test is Woop, Benthin and Wald's in single precision; from an origin hundreds of scene sizes away the
    translated and sheared vertices collapse, edge functions come out as exact zeros, and the test admits a "hit" whose
    point o + t d lies nowhere near the triangle. No acceleration structure can find such a hit (its boxes are nowhere near
    the ray), so brute force and every traversal — the oracle's own BVH and the HIP kernels, which agree with each other
    bit for bit (tools/fuzz_parity.py rays) — differ there. What this test fixes: (1) for origins within ten scene sizes
    brute force and the BVH traversal agree on every ray; (2) from farther away every disagreement is of that one kind —
    brute force reports a hit the traversal does not, and its hit point lies outside the scene's bounds by more than
    the scene's size. A guard in the contract (hit point against the triangle's padded bounds) would remove the hole; it
    changes both sides and every golden frame, and is not made."""
    sc, _ = scenes.forest(n_instances=12, tree_tris=400, tree_kinds=2)
    o = orc.OracleScene(sc)
    lo, hi = np.array([-12.0, -1.0, -12.0]), np.array([12.0, 9.0, 12.0])  # (generous bounds of the instanced scene)
    size = float((hi - lo).max())
    rng = np.random.default_rng(9)
    for scale, must_agree in ((1.0, True), (10.0, True), (300.0, False), (3000.0, False)):
        n = 12000
        origin = rng.uniform(lo, hi, (n, 3)) * scale
        target = rng.uniform(lo, hi, (n, 3))
        rays = np.zeros(n, wire.Ray)
        rays["origin"] = origin.astype(np.float32)
        rays["direction"] = (target - origin).astype(np.float32)  # unnormalised, aimed at the scene
        rays["tmin"], rays["tmax"] = 0.0, np.inf
        brute, _ = o.trace(rays, brute=True)
        bvh, _ = o.trace(rays)
        differ = np.nonzero(brute["instance_primitive_index"] != bvh["instance_primitive_index"])[0]
        same_where_equal = np.ones(n, bool)
        same_where_equal[differ] = False
        for f in ("t", "b1", "b2"):
            assert np.array_equal(brute[f][same_where_equal].view(np.uint32), bvh[f][same_where_equal].view(np.uint32)), (scale, f)
        if must_agree:
            assert differ.size == 0, (scale, differ.size)
            continue
        assert differ.size < n // 200, (scale, differ.size)  # rare
        if differ.size:
            assert (bvh["instance_primitive_index"][differ] == wire.MISS).all() or (brute["t"][differ] < bvh["t"][differ]).all(), scale
            hp = rays["origin"][differ].astype(np.float64) + brute["t"][differ].astype(np.float64)[:, None] * rays["direction"][differ].astype(np.float64)
            outside = ((hp < lo - size) | (hp > hi + size)).any(axis=1) | ~np.isfinite(hp).all(axis=1)
            assert outside.all(), (scale, hp[~outside])
