"""Sweep of the persistent trace kernel's scheduler knobs on the bench workload (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

sc, cam = scenes.atrium()
fr = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
r = BDPT(0)
r.update(sc)
buf = torch.zeros((1080, 1920, 4), device="cuda")
out = {"radiance": buf.data_ptr()}
def run(steps=15):
    for i in range(3):
        r.render(fr, i, 1, device_outputs=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        r.render(fr, 3 + i, 1, device_outputs=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
for refill in (8, 12, 16, 24, 32):
    row = []
    for lanes in (12, 16, 24, 32, 40):
        r.set_option("refill_idle", refill)
        r.set_option("inner_min_lanes", lanes)
        row.append("%.3f" % run())
    print("refill_idle %2d | inner_min_lanes 12/16/24/32/40: %s" % (refill, " ".join(row)))
