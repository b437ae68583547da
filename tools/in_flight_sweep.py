"""GPU box: "max_paths_in_flight" (how many seeds of the owned pixels are traced together) on the multi-seed configurations:
atrium 1080p x 8 seeds and the forest at 4K x 4 seeds with config 5's budgets. usage: python tools/in_flight_sweep.py"""
import os
os.environ.setdefault("STHIP_STRICT_FLAGS", "1")  # a misspelt --bdptFlag name is an error in a tool that measures (the mirror ignores it, as upstream does)
import sys, time

import numpy as np
import torch

sys.path.insert(0, ".")
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

for name, make, W, H, seeds, args in (
    ("atrium 1080p x 8", scenes.atrium, 1920, 1080, 8, {}),
    ("forest 4K x 4 (8 diffuse / 10 path vertices)", scenes.forest, 3840, 2160, 4, {"maxDiffuseVertices": 8, "maxPathVertices": 10, "minPathVertices": 4, "bdptFlag": ["~coherentrr"]}),
):
    sc, cam = make()
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    rad = torch.zeros((H, W, 4), device="cuda")
    rc = torch.zeros(2, dtype=torch.int64, device="cuda")
    out = {"radiance": rad.data_ptr(), "ray_count": rc.data_ptr()}
    ref = None
    for shift in (22, 23, 24, 25, 26, 27):
        r = BDPT(0, args=args)
        r.set_option("answer_last_rays", 0)
        r.set_option("max_paths_in_flight", 1 << shift)
        r.update(sc)
        r.render(fr, 0, seeds, device_outputs=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            r.render(fr, 0, seeds, device_outputs=out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        img = rad.cpu().numpy().copy()
        if ref is None:
            ref = img
        print("%-46s max_paths_in_flight = 2^%d: %8.3f ms, %6.0f Mray/s, identical to the first: %s, device memory %.1f GB" % (name, shift, np.median(ts) * 1e3, int(rc[0].item()) / np.median(ts) / 1e6, np.array_equal(ref.view(np.uint32), img.view(np.uint32)), torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9), flush=True)
        r.close()
