"""The reservoir-reuse estimators with their hash grids built between the seeds of a call (GPU box): ms per seed of an 8-seed
call on the bench scene at 1080p, with the grids built by the parallel device path and by its one-thread serial path."""
import os, sys, time
os.environ.setdefault("STHIP_STRICT_FLAGS", "1")  # a misspelt --bdptFlag name is an error in a tool that measures (the mirror ignores it, as upstream does)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

sc, cam = scenes.atrium()
fr = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
buf = torch.zeros((1080, 1920, 4), device="cuda")
out = {"radiance": buf.data_ptr()}
for flags in (["neereservoirs"], ["neereservoirs", "neereservoirreuse"], ["connecttolightpaths", "lightvertexcache", "lvcreservoirs"], ["connecttolightpaths", "lightvertexcache", "lvcreservoirs", "lvcreservoirreuse"]):
    for serial in ((0, 1) if any("reuse" in f for f in flags) else (0,)):
        r = BDPT(0, args={"bdptFlag": flags, "lightPathCount": 1920 * 1080})  # (the cache as large as the light paths of a frame without it)
        r.set_option("hashgrid_serial", serial)
        r.update(sc)
        r.render(fr, 0, 2, device_outputs=out)
        torch.cuda.synchronize()
        t = time.perf_counter()
        r.render(fr, 8, 8, device_outputs=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 8
        rays = r.stats()["rays_total"] / 8
        print("%-70s %s: %.3f ms per seed, %.0f Mray/s" % (" ".join(flags), "serial grid build" if serial else "device grid build", dt * 1e3, rays / dt / 1e6), flush=True)
        r.close()
