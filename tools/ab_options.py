"""A/B of context options on the bench workload (GPU box): ms per step and Mray/s for each setting of one option.
usage: python tools/ab_options.py treetop=0,1 [refill_idle=16 ...] [--scene atrium] [--steps 20]
Options given with several values are swept (the first one listed is swept, the others fixed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

args = [a for a in sys.argv[1:] if "=" in a]
scene = "atrium"
steps = 20
for i, a in enumerate(sys.argv):
    if a == "--scene":
        scene = sys.argv[i + 1]
    if a == "--steps":
        steps = int(sys.argv[i + 1])
sweep = []
for a in args:
    k, v = a.split("=")
    sweep.append((k, [int(x) for x in v.split(",")]))
sc, cam = scenes.SCENES[scene]()
fr = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
buf = torch.zeros((1080, 1920, 4), device="cuda")
out = {"radiance": buf.data_ptr()}


def run(r):
    for i in range(3):
        r.render(fr, i, 1, device_outputs=out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for i in range(steps):
            r.render(fr, 3 + i, 1, device_outputs=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / steps * 1e3)
    r.render(fr, 3, 1, device_outputs=out)
    rays = r.stats()["rays_total"]
    r.set_option("time_kernels", 1)
    r.render(fr, 3, 1, device_outputs=out)
    s = r.stats()
    r.set_option("time_kernels", 0)
    return best, rays / best / 1e3, s["ms_trace"], s["ms_trace_primary"], s["ms_shade"]


key, values = sweep[0]
ref = None
for v in values:
    r = BDPT(0)
    for k, vs in sweep[1:]:
        r.set_option(k, vs[0])
    r.set_option(key, v)
    r.update(sc)
    ms, mray, mt, mp, msh = run(r)
    img = buf.clone()
    same = "" if ref is None else ("  image identical to first: %s" % bool(torch.equal(img, ref)))
    if ref is None:
        ref = img
    print("%s=%d: %.3f ms/step, %.0f Mray/s | k_trace %.3f ms, primary %.3f, shade %.3f%s" % (key, v, ms, mray, mt, mp, msh, same), flush=True)
    r.close()
