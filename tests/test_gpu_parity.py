"""Parity of the HIP path against the CPU oracle, through the C ABI (needs an MI355X)."""
import numpy as np
import pytest

from stratum_amd import camera, scenes, wire

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b**2).sum()), 1e-300))


@pytest.fixture(scope="module")
def renderer(built):
    from stratum_amd.bdpt import BDPT

    r = BDPT(device=0)
    yield r
    r.close()


def random_rays(n, seed, lo, hi, tmax=np.inf):
    rng = np.random.RandomState(seed)
    rays = np.zeros(n, wire.Ray)
    rays["origin"] = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["direction"] = d.astype(np.float32)
    rays["tmin"] = 0
    rays["tmax"] = tmax
    return rays


def test_trace_contract_cornell(renderer, cornell):
    from oracle import oracle_py

    sc, _ = cornell
    renderer.update(sc)
    o = oracle_py.OracleScene(sc)
    rays = random_rays(50000, 1, -0.99, 0.99)
    got = renderer.trace(rays)
    ref, _ = o.trace(rays, brute=True)
    assert np.array_equal(got["instance_primitive_index"], ref["instance_primitive_index"])
    assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(got["b1"].view(np.uint32), ref["b1"].view(np.uint32))
    assert np.array_equal(got["b2"].view(np.uint32), ref["b2"].view(np.uint32))
    # occlusion
    rays["tmax"] = 0.7
    got = renderer.trace(rays, any_hit=True)
    ref, _ = o.trace(rays, any_hit=True, brute=True)
    assert np.array_equal(got["instance_primitive_index"], ref["instance_primitive_index"])


def test_cornell_256_one_sample(renderer, cornell):
    """configs[0]: Cornell box, 256x256, 1 sample, default flags."""
    from oracle import oracle_py

    sc, cam = cornell
    renderer.update(sc)
    frame = camera.Frame(256, 256, cam["fovy"], cam["eye"], cam["target"])
    got = renderer.render(frame, 0, 1)
    ref = oracle_py.OracleScene(sc).render(frame, renderer.push_constants(frame), renderer.mSamplingFlags, 0, 1)
    assert np.array_equal(got["visibility"]["instance_primitive_index"], ref["visibility"]["instance_primitive_index"])
    assert np.array_equal(got["visibility"]["packed_normal"], ref["visibility"]["packed_normal"])
    assert np.array_equal(got["albedo"], ref["albedo"])
    assert np.array_equal(got["ray_count"], ref["ray_count"])
    d = rel_l2(got["radiance"], ref["radiance"])
    nd = int((got["radiance"].view(np.uint32) != ref["radiance"].view(np.uint32)).any(axis=-1).sum())
    print("rel-L2 %.3e, pixels that differ in any bit: %d" % (d, nd))
    assert d <= 1e-4  # north_star tolerance on the HDR framebuffer
    assert np.array_equal(got["depth"]["z"].view(np.uint32), ref["depth"]["z"].view(np.uint32))
    assert np.allclose(got["prev_uv"], ref["prev_uv"], rtol=0, atol=0)
