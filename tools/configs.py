"""SURVEY §8d's other configurations on one GPU (GPU box): config 2 (Cornell 1080p, seeds 0..63), config 3 x 256 seeds
(config 4's per-frame work on one GPU), config 5 (forest, 4K, 16 seeds, its flags) — complete, finite, rates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

def run(name, sc, cam, W, H, seeds, args):
    r = BDPT(0, args=args)
    r.update(sc)
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    buf = torch.zeros((H, W, 4), device="cuda")
    rc = torch.zeros(2, dtype=torch.int64, device="cuda")
    out = {"radiance": buf.data_ptr(), "ray_count": rc.data_ptr()}
    r.render(fr, 0, 1, device_outputs=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    r.render(fr, 0, seeds, device_outputs=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    rays = int(rc[0].item())
    img = buf.cpu().numpy()
    print("%-34s %dx%d x %3d seeds: %8.1f ms, %6.0f Mray/s, mean %.4f, samples/pixel %.0f, finite %s" % (name, W, H, seeds, dt * 1e3, rays / dt / 1e6, img[..., :3].mean(), img[..., 3].max(), bool(np.isfinite(img).all())))
    r.close()

sc, cam = scenes.cornell_box()
run("config 2: Cornell box", sc, cam, 1920, 1080, 64, {})
sc, cam = scenes.atrium()
run("config 3 x 256 seeds: atrium", sc, cam, 1920, 1080, 256, {})
sc, cam = scenes.forest()
run("config 5: forest 10M", sc, cam, 3840, 2160, 16, {"maxDiffuseVertices": 8, "maxPathVertices": 10, "minPathVertices": 4, "bdptFlag": ["~coherentrr"]})
