"""one row per configuration x backend, measured on the box this runs on (one MI355X + its host
cores). GPU: median of 5 timed sthip_render calls with device outputs after a warm-up; CPU: the oracle built -O3
-march=native on this box, on a BOUNDED window / seed count of the same frame (the whole frames of configs 2-5 would take
hours), median of 3. usage (GPU box): python tools/baseline_table.py [--no-cpu] [--answer-last-rays] > gpurun_out/baseline_table.md"""
import os, sys, time
os.environ.setdefault("STHIP_STRICT_FLAGS", "1")  # a misspelt --bdptFlag name is an error in a tool that measures (the mirror ignores it, as upstream does)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stratum_amd import camera, scenes, shard
from stratum_amd.bdpt import BDPT

NO_CPU = "--no-cpu" in sys.argv
ANSWER = 1 if "--answer-last-rays" in sys.argv else 0  # the table traces every ray; with the flag: the library's default (sthip.h "answer_last_rays"), rays in gRayCount semantics
if not NO_CPU:
    from oracle import oracle_py

    oracle_py.build_native()  # -O3 -march=native for THIS box's cores, before anything touches the GPU


def gpu_leg(sc, cam, W, H, seeds, args, shard_of=None):
    r = BDPT(0, args=args)
    try:
        if shard_of:
            r.set_shard(0, shard_of, 64, 32)
        r.update(sc)
        r.set_option("answer_last_rays", ANSWER)
        fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
        packed = shard_of is not None
        rad = torch.zeros((r.shard_slot_count(fr), 4) if packed else (H, W, 4), device="cuda")
        rc = torch.zeros(2, dtype=torch.int64, device="cuda")
        out = {"radiance": rad.data_ptr(), "ray_count": rc.data_ptr()}
        r.render(fr, 0, min(seeds, 2), device_outputs=out, packed_tiles=packed)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            r.render(fr, 0, seeds, device_outputs=out, packed_tiles=packed)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        rays = int(rc[0].item())
        image = rad.cpu().numpy() if not packed else None  # of the timed call (the counting pass below renders fewer seeds)
        r.set_option("count_traversal", 1)
        r.render(fr, 0, min(seeds, 2), device_outputs=out, packed_tiles=packed)
        torch.cuda.synchronize()
        s = r.stats()
        nrays = max(s["rays_path"] + s["rays_shadow"], 1)
        n_node = (s["nodes_visited"] + s["nodes_visited_shadow"]) / nrays
        n_tri = (s["tris_tested"] + s["tris_tested_shadow"]) / nrays
        return dict(rays=rays, sec=float(np.median(ts)), n_node=n_node, n_tri=n_tri, node_bytes=s["bvh_node_bytes"], pc=r.push_constants(fr), flags=r.mSamplingFlags, frame=fr,
                    image=image)
    finally:
        r.close()


def host_threads():
    """CPU threads this process may actually use: the cgroup quota if there is one, else the affinity mask (as bench.py)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_leg(sc, g, window, seeds):
    from oracle import oracle_py

    o = oracle_py.OracleScene(sc, native=True)
    ts, ref = [], None
    for _ in range(3):
        t = time.perf_counter()
        ref = o.render(g["frame"], g["pc"], g["flags"], 0, seeds, threads=host_threads(), window=window)
        ts.append(time.perf_counter() - t)
    return dict(rays=int(ref["ray_count"][0]), sec=float(np.median(ts)), ref=ref, nodes=ref.get("nodes_per_ray"), tris=ref.get("tris_per_ray"))


def rel_l2(a, b):
    a, b = a[..., :3].astype(np.float64), b[..., :3].astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b**2).sum()), 1e-300))


CONFIGS = [
    ("1", "Cornell 256x256 x 1", lambda: scenes.cornell_box(), 256, 256, 1, {}, None, (0, 0, 256, 256), 1),
    ("2", "Cornell 1920x1080 x 64", lambda: scenes.cornell_box(), 1920, 1080, 64, {}, None, (640, 360, 1280, 720), 16),
    ("3", "atrium 1M tris, 1920x1080 x 1", lambda: scenes.atrium(), 1920, 1080, 1, {}, None, (0, 0, 1920, 1080), 1),
    ("3b", "atrium 1M tris, 1920x1080 x 8 (bench.py's parity sample)", lambda: scenes.atrium(), 1920, 1080, 8, {}, None, (0, 0, 1920, 1080), 8),
    ("4", "atrium, 1920x1080 x 256, the work of ONE of 8 ranks (tiles 64x32, t % 8 == 0)", lambda: scenes.atrium(), 1920, 1080, 256, {}, 8, None, 0),
    ("5", "forest 10M tris instanced, 3840x2160 x 16, 8 diffuse / 10 path vertices", lambda: scenes.forest(), 3840, 2160, 16,
     {"maxDiffuseVertices": 8, "maxPathVertices": 10, "minPathVertices": 4, "bdptFlag": ["~coherentrr"]}, None, (1600, 900, 2240, 1260), 4),
]

print("| # | workload | backend | rays | seconds | Mray/s | n_node | n_tri | B_ray | frac of 8 TB/s | rel-L2 vs oracle | host cores |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
cores = host_threads()
for tag, name, make, W, H, seeds, args, shard_of, window, cpu_seeds in CONFIGS:
    sc, cam = make()
    g = gpu_leg(sc, cam, W, H, seeds, args, shard_of)
    b_ray = 48 + g["node_bytes"] * g["n_node"] + 48 * g["n_tri"]
    rate = g["rays"] / g["sec"]
    line = "| %s | %s | 1x MI355X | %d | %.4f | %.0f | %.1f | %.2f | %.0f | %.2f | %s | - |"
    rel = "-"
    c = None
    if window and not NO_CPU:
        c = cpu_leg(sc, g, window, cpu_seeds)
        if cpu_seeds == seeds and g["image"] is not None:
            x0, y0, x1, y1 = window
            rel = "%.1e" % rel_l2(g["image"][y0:y1, x0:x1], c["ref"]["radiance"][y0:y1, x0:x1])
    print(line % (tag, name, g["rays"], g["sec"], rate / 1e6, g["n_node"], g["n_tri"], b_ray, rate * b_ray / 8e12, rel), flush=True)
    if c:
        x0, y0, x1, y1 = window
        print("| %s | window %dx%d x %d seed(s) of it | CPU oracle, -O3 -march=native | %d | %.3f | %.2f | - | - | - | - | (checker) | %d |" % (tag, x1 - x0, y1 - y0, cpu_seeds, c["rays"], c["sec"], c["rays"] / c["sec"] / 1e6, cores), flush=True)
