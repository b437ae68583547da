"""Full-size runs of the non-default paths (GPU box): they complete, stay finite, and report their rates."""
import os, sys, time
os.environ.setdefault("STHIP_STRICT_FLAGS", "1")  # a misspelt --bdptFlag name is an error in a tool that measures (the mirror ignores it, as upstream does)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

fog = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fog_sphere.npz"))["grid"]
W, H = 1920, 1080
for name, (sc, cam), args in (
    ("foggy cornell, default flags", scenes.cornell_box(fog=fog), {"maxDiffuseVertices": 4}),
    ("foggy cornell, forward scattering, dense", scenes.cornell_box(fog=fog, anisotropy=0.7, density=(8, 8, 8)), {"maxDiffuseVertices": 8, "maxPathVertices": 12}),
    ("cornell, bidirectional", scenes.cornell_box(), {"bdptFlag": ["connecttolightpaths", "connecttoviews"], "maxDiffuseVertices": 4}),
):
    r = BDPT(0, args=args)
    r.update(sc)
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    r.render(fr, 0, 1, aovs=False)
    t = time.perf_counter()
    out = r.render(fr, 1, 8, aovs=False)
    dt = time.perf_counter() - t
    rad = out["radiance"][..., :3]
    print("%-45s %6.1f ms / 8 spp, %7.1f Mray/s, mean %.4f, finite %s" % (name, dt * 1e3, out["ray_count"][0] / dt / 1e6, rad.mean(), bool(np.isfinite(rad).all())), flush=True)
    r.close()

# round 4: media with the estimators that walk their visibility rays inline, at 1920x1080 against the oracle itself (one sample
# per pixel: a few seconds of the box's cores per row), bit for bit — frame and ray counts
from oracle import oracle_py

for name, flags, args in (
    ("fog box, inline NEE", ["~defershadowrays"], {"maxDiffuseVertices": 3}),
    ("fog box, NEE reservoirs + reuse", ["neereservoirs", "neereservoirreuse", "~defershadowrays"], {"maxDiffuseVertices": 3, "reservoirM": 4}),
    ("fog box, light tracing", ["connecttoviews"], {"maxDiffuseVertices": 3}),
    ("fog box, light-subpath connections", ["connecttolightpaths", "connecttoviews"], {"maxDiffuseVertices": 3}),
    ("fog box, light vertex cache + reservoirs", ["connecttolightpaths", "lightvertexcache", "lvcreservoirs", "~defershadowrays"], {"maxDiffuseVertices": 3, "lightPathCount": 500000, "reservoirM": 4}),
):
    sc, cam = scenes.fog_box()
    r = BDPT(0, args=dict(args, bdptFlag=flags))
    r.update(sc)
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    r.render(fr, 0, 1, aovs=False)
    t = time.perf_counter()
    got = r.render(fr, 5, 1, aovs=False)
    dt = time.perf_counter() - t
    t = time.perf_counter()
    ref = oracle_py.OracleScene(sc).render(fr, r.push_constants(fr), r.mSamplingFlags, 5, 1, threads=os.cpu_count(), aovs=False)
    odt = time.perf_counter() - t
    same = np.array_equal(got["radiance"].view(np.uint32), ref["radiance"].view(np.uint32)) and np.array_equal(got["ray_count"], ref["ray_count"])
    print("%-45s %6.1f ms, %7.1f Mray/s (host outputs), oracle %.1f s, identical %s" % (name, dt * 1e3, got["ray_count"][0] / dt / 1e6, odt, same), flush=True)
    r.close()
    if not same:
        sys.exit(1)
