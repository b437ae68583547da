"""Acceleration-structure build and update times on the bench scene (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT

for name, make in (("atrium 1M", scenes.atrium), ("forest 10M instanced", scenes.forest)):
    sc, cam = make()
    for builder in (0, 1):
        r = BDPT(0)
        r.set_option("bvh_builder", builder)
        t = time.perf_counter()
        try:
            r.update(sc)
        except Exception as e:
            print(name, "builder", builder, "->", str(e)[:80])
            r.close()
            continue
        up = (time.perf_counter() - t) * 1e3
        s = r.stats()
        line = "%s, %s: upload %.0f ms (build %.0f ms, GPU kernels %.1f ms), %d nodes" % (name, "LBVH/GPU" if builder else "SAH/host", up, s["bvh_build_ms"], s["bvh_build_gpu_ms"], s["bvh_nodes"])
        ident = np.array([np.array_equal(m, np.eye(4, dtype=np.float32)[:3]) for m in sc.transforms["m"]])
        movers = np.nonzero(~ident)[0]
        if movers.size:
            for i in movers:
                m = sc.transforms["m"][i].copy()
                m[:, 3] += np.float32(0.01)
                sc.set_instance_transform(int(i), m)
            t = time.perf_counter()
            r.update_transforms(sc)
            line += "; transforms-only update of %d instances %.2f ms" % (movers.size, (time.perf_counter() - t) * 1e3)
        print(line)
        r.close()
