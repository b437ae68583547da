"""How much of a k_trace launch is ramp-up and drain (GPU box): kernel times of the bench frame with 1, 2, 4, 8 seeds per
render call (the seeds of a call share the launches, so a launch traces n times the rays) and the fit t = T0 + n * t1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

sc, cam = scenes.atrium()
fr = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
r = BDPT(0)
r.set_option("max_paths_in_flight", 1 << 25)
r.update(sc)
buf = torch.zeros((1080, 1920, 4), device="cuda")
out = {"radiance": buf.data_ptr()}
r.set_option("time_kernels", 1)
rows = []
for n in (1, 2, 4, 8):
    acc = np.zeros(4)
    reps = 6
    for k in range(reps + 1):
        r.render(fr, 16 * k, n, device_outputs=out)
        s = r.stats()
        if k:
            acc += (s["ms_trace"], s["ms_trace_primary"], s["ms_shade"], s["ms_total"])
    acc /= reps
    rows.append((n, acc))
    print("seeds per call %d: k_trace %.3f ms, primary %.3f, shade %.3f, all kernels %.3f | per seed: %.3f %.3f %.3f %.3f" % ((n,) + tuple(acc) + tuple(acc / n)))
ns = np.array([x[0] for x in rows], float)
for j, name in enumerate(("k_trace", "k_trace_primary", "k_shade", "all kernels")):
    t = np.array([x[1][j] for x in rows])
    t1, t0 = np.polyfit(ns, t, 1)
    print("%s: t(n) = %.3f ms + n * %.3f ms  (fixed part = %.1f %% of the one-seed time)" % (name, t0, t1, 100 * t0 / t[0]))
