import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["STHIP_VERBOSE"] = "1"
import torch
from stratum_amd import camera, scenes
from stratum_amd.bdpt import BDPT
sc, cam = scenes.SCENES[sys.argv[1] if len(sys.argv) > 1 else "atrium"]()
fr = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
buf = torch.zeros((1080, 1920, 4), device="cuda")
out = {"radiance": buf.data_ptr()}
for opts in ({"wide_bvh": 0}, {"wide_bvh": 1}, {"wide_bvh": 1, "lds_stack_levels": 24}, {"wide_bvh": 1, "lds_stack_levels": 38}):
    r = BDPT(0)
    for k, v in opts.items():
        r.set_option(k, v)
    r.update(sc)
    for i in range(3):
        r.render(fr, i, 1, device_outputs=out)
    torch.cuda.synchronize()
    r.set_option("time_kernels", 1)
    r.render(fr, 3, 1, device_outputs=out)
    s = r.stats()
    r.set_option("time_kernels", 0)
    r.set_option("count_traversal", 1)
    r.render(fr, 3, 1, device_outputs=out)
    c = r.stats()
    print(opts, "k_trace %.3f ms | nodes %d (primary %d) tris %d | inner_slots %s tri_slots %s" % (s["ms_trace"], c["nodes_visited"], c["nodes_visited_primary"], c["tris_tested"], c.get("inner_slots"), c.get("tri_slots")), flush=True)
    r.close()
