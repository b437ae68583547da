"""Experiment: consecutive frames on alternating contexts / HIP streams (frames in flight, as the reference's per-frame
resource pool allows): does the next frame's work fill the drain phase of the previous frame's persistent kernels?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

sc, cam = scenes.atrium()
W, H = 1920, 1080
frame = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
steps = int(os.environ.get("STEPS", "40"))
for n in (1, 2, 3):
    rs, bufs = [], []
    for k in range(n):
        r = BDPT(0)
        r.update(sc)
        s = torch.cuda.Stream()
        r.set_stream(s.cuda_stream)
        rs.append((r, s))
        bufs.append(torch.zeros((H, W, 4), device="cuda"))
    def step(i):
        r, s = rs[i % n]
        r.render(frame, seed_begin=i, seed_count=1, device_outputs={"radiance": bufs[i % n].data_ptr()})
    for i in range(2 * n):
        step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        step(2 * n + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    rays = sum(r.stats()["rays_total"] for r, _ in rs) / n
    print("%d frame(s) in flight: %.3f ms/step, %.0f Mray/s" % (n, dt * 1e3, rays / dt / 1e6))
    for r, _ in rs:
        r.close()
