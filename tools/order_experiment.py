"""Experiment: how much does the order of incoherent secondary rays matter to the traversal kernel?
Run under `rocprofv3 --kernel-trace --stats`; the k_trace_batch dispatches appear in the order printed here."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratum_amd import scenes, wire
from stratum_amd.bdpt import BDPT

sc, cam = scenes.atrium()
r = BDPT(device=0)
r.update(sc)
W, H = 1920, 1080
eye = np.array(cam["eye"], np.float64); tgt = np.array(cam["target"], np.float64)
f = tgt - eye; f /= np.linalg.norm(f)
rt = np.cross(f, [0, 1, 0]); rt /= np.linalg.norm(rt)
up = np.cross(rt, f)
th = np.tan(cam["fovy"] / 2)
# pixels in 8x8 block order (the renderer's slot order)
ys, xs = np.mgrid[0:H, 0:W]
key = ((ys // 8) * (W // 8) + xs // 8) * 64 + (ys % 8) * 8 + xs % 8
order = np.argsort(key.ravel(), kind="stable")
px = xs.ravel()[order]; py = ys.ravel()[order]
cx = (2 * (px + 0.5) / W - 1) * th * W / H; cy = -(2 * (py + 0.5) / H - 1) * th
d = f[None] + cx[:, None] * rt[None] + cy[:, None] * up[None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros(W * H, wire.Ray)
rays["origin"] = eye; rays["direction"] = d; rays["tmin"] = 0; rays["tmax"] = 1e30
hits = r.trace(rays)
ok = hits["instance_primitive_index"] != wire.MISS
print("primary hit fraction", ok.mean())
pos = eye[None] + d * (hits["t"][:, None].astype(np.float64) * 0.999)
rng = np.random.default_rng(1)
d2 = rng.normal(size=(W * H, 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
flip = (d2 * -d).sum(1) < 0
d2[flip] = -d2[flip]
sec = np.zeros(int(ok.sum()), wire.Ray)
sec["origin"] = pos[ok]; sec["direction"] = d2[ok]; sec["tmin"] = 0; sec["tmax"] = 1e30
octant = (sec["direction"][:, 0] < 0) * 1 + (sec["direction"][:, 1] < 0) * 2 + (sec["direction"][:, 2] < 0) * 4
n = sec.shape[0]
def windowed(win):
    w = np.arange(n) // win
    return np.lexsort((np.arange(n), octant, w))
orders = [("block order", np.arange(n)), ("octant in 256-windows", windowed(256)), ("octant in 4096-windows", windowed(4096)), ("octant in 65536-windows", windowed(65536)),
          ("octant global", np.argsort(octant, kind="stable")), ("random", rng.permutation(n))]
ref = None
for name, o in orders:
    for rep in range(2):
        h = r.trace(sec[o])
    inv = np.empty(n, np.int64); inv[o] = np.arange(n)
    t = h["t"][inv]
    if ref is None: ref = t
    assert np.array_equal(ref.view(np.uint32), t.view(np.uint32))
    print("dispatch pair:", name)
    for any_hit in (True,):
        r.trace(sec[o], any_hit=True)
