#!/usr/bin/env python3
"""Sweeps the scheduler knobs of the persistent trace kernels on the bench workload (GPU box)."""
import itertools
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from stratum_amd import camera, scenes  # noqa: E402
from stratum_amd.bdpt import BDPT  # noqa: E402

sc, cam = scenes.atrium()
frame = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
r = BDPT(0)
r.update(sc)
rad = torch.zeros((1080, 1920, 4), device="cuda")
out = {"radiance": rad.data_ptr()}


def run(n=8):
    for i in range(2):
        r.render(frame, i, 1, device_outputs=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        r.render(frame, 2 + i, 1, device_outputs=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


knobs = {"refill_idle": [4, 8, 12, 16, 20, 28, 40], "inner_min_lanes": [1, 4, 8, 12, 16, 24, 32]}
if len(sys.argv) > 1:
    knobs = eval(sys.argv[1])
names = list(knobs)
for vals in itertools.product(*[knobs[k] for k in names]):
    for k, v in zip(names, vals):
        r.set_option(k, v)
    print(dict(zip(names, vals)), "%.3f ms/step" % run(), flush=True)
