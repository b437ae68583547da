"""Experiment: the frame as two (or more) independent pixel-tile shards of ONE GPU, each rendered by its own host thread on
its own context and HIP stream, free-running (the threads are started half a step apart): do the kernels of one shard fill the
ramp-up and drain of the other's persistent trace launches?  Prints Mray/s of the whole frame against one context."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratum_amd import camera, scenes, shard
from stratum_amd.bdpt import BDPT

sc, cam = scenes.atrium()
W, H = 1920, 1080
frame = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
steps = int(os.environ.get("STEPS", "30"))


def run(nsplit, stagger_ms):
    rs, bufs = [], []
    for h in range(nsplit):
        r = BDPT(0)
        r.update(sc)
        r.set_shard(h, nsplit, 64, 32)
        s = torch.cuda.Stream()
        r.set_stream(s.cuda_stream)
        rs.append((r, s))
        bufs.append(torch.zeros((max(1, shard.slot_count(W, H, 0, nsplit, 64, 32)), 4), device="cuda"))
    for h, (r, s) in enumerate(rs):
        for i in range(3):
            r.render(frame, seed_begin=i, seed_count=1, device_outputs={"radiance": bufs[h].data_ptr()}, packed_tiles=True)
    torch.cuda.synchronize()
    go = threading.Event()

    def worker(h):
        r, s = rs[h]
        go.wait()
        time.sleep(h * stagger_ms * 1e-3)
        for i in range(steps):
            r.render(frame, seed_begin=3 + i, seed_count=1, device_outputs={"radiance": bufs[h].data_ptr()}, packed_tiles=True)
        s.synchronize()

    th = [threading.Thread(target=worker, args=(h,)) for h in range(nsplit)]
    for t in th:
        t.start()
    t0 = time.perf_counter()
    go.set()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rays = sum(r.stats()["rays_total"] for r, _ in rs)
    for r, _ in rs:
        r.close()
    return dt * 1e3, rays / dt / 1e6


for n, st in ((1, 0.0), (2, 0.0), (2, 0.7), (2, 1.4), (3, 0.9), (4, 0.7)):
    ms, rate = run(n, st)
    print("%d shard(s) on own threads / streams, started %.1f ms apart: %.3f ms per frame, %.0f Mray/s" % (n, st, ms, rate), flush=True)
