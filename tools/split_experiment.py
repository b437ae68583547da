"""Experiment: does running the frame as two independent half-frame pipelines on two HIP streams (tile-interleaved
shards of one GPU) hide the drain phase of the persistent trace kernels?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratum_amd import camera, scenes, shard
from stratum_amd.bdpt import BDPT

sc, cam = scenes.atrium()
W, H = 1920, 1080
frame = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
steps = 20

def run(nsplit, tile=(64, 32)):
    rs, streams, bufs = [], [], []
    for h in range(nsplit):
        r = BDPT(0)
        r.update(sc)
        r.set_shard(h, nsplit, *tile)
        s = torch.cuda.Stream()
        r.set_stream(s.cuda_stream)
        rs.append(r); streams.append(s)
        bufs.append(torch.zeros((shard.slot_count(W, H, 0, nsplit, *tile), 4), device="cuda"))
    def step(i):
        for h in range(nsplit):
            rs[h].render(frame, seed_begin=i, seed_count=1, device_outputs={"radiance": bufs[h].data_ptr()}, packed_tiles=True)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        step(3 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    rays = sum(r.stats()["rays_total"] for r in rs)
    for r in rs:
        r.close()
    return dt * 1e3, rays / dt / 1e6

for n in (1, 2, 3, 4):
    ms, rate = run(n)
    print("split %d: %.3f ms/step, %.0f Mray/s" % (n, ms, rate))
ms, rate = run(2, (128, 64))
print("split 2 with 128x64 tiles: %.3f ms/step, %.0f Mray/s" % (ms, rate))
