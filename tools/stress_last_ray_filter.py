"""GPU box: "answer_last_rays" on and off at sample counts the test suite cannot afford (48 samples of 512 x 512: ~50 M rays per
scene) — the Cornell box, two emissive instances with transforms (one of them 2 cm wide: grazing rays), a deeper budget, the
atrium. Frames and ray counts must be bit-identical. usage: python tools/stress_last_ray_filter.py"""
import sys, numpy as np
sys.path.insert(0, ".")
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT
from stratum_amd.scene import rotate_y, scale, translate
def scene():
    sc0, cam = scenes.cornell_box()
    b = sc0.builder
    glow = b.add_emitter((3.0, 6.0, 9.0))
    pos, nrm, uv, tri = scenes._quad((-0.5, 0.0, 0.5), (0.5, 0.0, 0.5), (0.5, 0.0, -0.5), (-0.5, 0.0, -0.5), (0, 1, 0))
    panel = b.add_mesh(pos, nrm, uv, tri)
    b.add_instance(panel, glow, translate((0.55, -0.2, 0.1)) @ rotate_y(0.7) @ scale((0.3, 1.0, 0.5)))
    b.add_instance(panel, glow, translate((-0.6, 0.4, -0.5)) @ rotate_y(-1.1) @ scale((0.02, 1.0, 0.02)))  # a tiny one: grazing rays
    return b.build(), cam
for name, (sc, cam), args in (("cornell", scenes.cornell_box(), {}), ("emitters", scene(), {}), ("emitters deep", scene(), {"maxDiffuseVertices": 4, "maxPathVertices": 6}), ("atrium", scenes.atrium(target_tris=300000), {})):
    frame = camera.Frame(512, 512, cam["fovy"], cam["eye"], cam["target"])
    out = []
    for opt in (1, 0):
        r = BDPT(0, args=args)
        r.set_option("answer_last_rays", opt)
        r.update(sc)
        o = r.render(frame, 0, 48, aovs=False)
        out.append((o["radiance"].copy(), o["ray_count"].copy(), r.stats()["rays_answered"]))
        r.close()
    same = np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) and np.array_equal(out[0][1], out[1][1])
    print(name, "identical:", same, "rays", int(out[0][1][0]), "answered", out[0][2], flush=True)
    assert same
