import os
os.environ.setdefault("STHIP_STRICT_FLAGS", "1")  # a misspelt --bdptFlag name is an error in a tool that measures (the mirror ignores it, as upstream does)
import sys, numpy as np, torch
sys.path.insert(0, ".")
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT
sc, cam = scenes.forest()
args = {"maxDiffuseVertices": 8, "maxPathVertices": 10, "minPathVertices": 4, "bdptFlag": ["~coherentrr"]}
r = BDPT(0, args=args)
r.set_option("answer_last_rays", 0)
r.update(sc)
fr = camera.Frame(3840, 2160, cam["fovy"], cam["eye"], cam["target"])
rad = torch.zeros((2160, 3840, 4), device="cuda"); rc = torch.zeros(2, dtype=torch.int64, device="cuda")
out = {"radiance": rad.data_ptr(), "ray_count": rc.data_ptr()}
r.render(fr, 0, 2, device_outputs=out); torch.cuda.synchronize()
r.set_option("time_kernels", 1)
r.render(fr, 0, 4, device_outputs=out); torch.cuda.synchronize()
s = r.stats()
print({k: s[k] for k in ("ms_trace", "ms_trace_primary", "ms_shade", "ms_total", "launches_trace", "launches_primary", "rays_total", "rays_path", "rays_shadow", "rays_primary_packets")})
print("Mray/s (kernels only):", s["rays_total"] / s["ms_total"] / 1e3)
