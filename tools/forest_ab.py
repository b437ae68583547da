#!/usr/bin/env python3
"""Config 5 (forest, 3840x2160 x 16 seeds, 8 / 10 / 4 vertices, ~coherentrr) under context options, e.g. wide_bvh=1,3:
python tools/forest_ab.py wide_bvh=1,3 [other=value ...]  (GPU box)"""
import os, sys, time
os.environ.setdefault("STHIP_STRICT_FLAGS", "1")  # a misspelt --bdptFlag name is an error in a tool that measures (the mirror ignores it, as upstream does)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

args = [a for a in sys.argv[1:] if "=" in a]
sweep = [(a.split("=")[0], [int(x) for x in a.split("=")[1].split(",")]) for a in args]
sc, cam = scenes.forest()
W, H, seeds = 3840, 2160, 16
fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
rad = torch.zeros((H, W, 4), device="cuda")
rc = torch.zeros(2, dtype=torch.int64, device="cuda")
out = {"radiance": rad.data_ptr(), "ray_count": rc.data_ptr()}
key, values = sweep[0]
ref = None
for v in values:
    r = BDPT(0, args={"maxDiffuseVertices": 8, "maxPathVertices": 10, "minPathVertices": 4, "bdptFlag": ["~coherentrr"]})
    for k, vs in sweep[1:]:
        r.set_option(k, vs[0])
    r.set_option(key, v)
    r.set_option("answer_last_rays", 0)
    r.update(sc)
    r.render(fr, 0, 2, device_outputs=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        r.render(fr, 0, seeds, device_outputs=out)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    img = rad.clone()
    same = "" if ref is None else "  image identical to first: %s" % bool(torch.equal(img, ref))
    ref = img if ref is None else ref
    print("%s=%d: %.2f ms, %.0f Mray/s (node bytes %d)%s" % (key, v, np.median(ts) * 1e3, int(rc[0].item()) / np.median(ts) / 1e6, r.stats()["bvh_node_bytes"], same), flush=True)
    r.close()
