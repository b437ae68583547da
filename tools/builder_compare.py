#!/usr/bin/env python3
"""Traversal speed of the two BVH builders on the bench workload (GPU box)."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from stratum_amd import camera, scenes  # noqa: E402
from stratum_amd.bdpt import BDPT  # noqa: E402

sc, cam = scenes.atrium()
frame = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
rad = torch.zeros((1080, 1920, 4), device="cuda")
out = {"radiance": rad.data_ptr()}
for kind, name in ((0, "sah/host"), (1, "lbvh/gpu")):
    r = BDPT(0)
    r.set_option("bvh_builder", kind)
    r.update(sc)
    for i in range(3):
        r.render(frame, i, 1, device_outputs=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(10):
        r.render(frame, 3 + i, 1, device_outputs=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    s = r.stats()
    print("%s: build %.1f ms (gpu kernels %.2f ms), %.3f ms/frame, %.0f Mray/s" % (name, s["bvh_build_ms"], s["bvh_build_gpu_ms"], dt * 1e3, s["rays_total"] / dt / 1e6))
    r.close()
