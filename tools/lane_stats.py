#!/usr/bin/env python3
"""Where the lanes of the trace kernels go (GPU box): per-loop lane utilisation from the count_traversal diagnostics."""
import sys

sys.path.insert(0, ".")
from stratum_amd import camera, scenes  # noqa: E402
from stratum_amd.bdpt import BDPT  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
sc, cam = scenes.SCENES[name]()
frame = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
r = BDPT(0)
for a in sys.argv[2:]:
    k, v = a.split("=")
    r.set_option(k, int(v))
r.update(sc)
r.set_option("count_traversal", 1)
for label, max_vertices in (("primary rays only", 2), ("default (3 closest + 2 shadow rounds)", 8)):
    r.mPushConstants.gMaxPathVertices = max_vertices
    r.render(frame, 0, 1, aovs=False)
    s = r.stats()
    print(label)
    ls = s["lane_states"]
    it = max(sum(ls[0:4]), 1)
    print(
        "  node-loop iterations: %.0f%% of lanes step, %.0f%% wait with a triangle leaf, %.0f%% wait with a sentinel / instance entry, %.0f%% have no ray | leaf phases: %.2f per 10 iterations, %.0f%% of lanes test a triangle, %.0f%% handle a sentinel / entry | refill stops: %.2f per 10 iterations"
        % (100 * ls[0] / it, 100 * ls[1] / it, 100 * ls[2] / it, 100 * ls[3] / it, 10 * ls[4] / it, 100 * ls[5] / max(ls[4], 1), 100 * ls[6] / max(ls[4], 1), 10 * ls[7] / it)
    )
    for k, kind in enumerate(("closest", "shadow")):
        nodes = s["nodes_visited" if k == 0 else "nodes_visited_shadow"]
        tris = s["tris_tested" if k == 0 else "tris_tested_shadow"]
        rays = s["rays_path" if k == 0 else "rays_shadow"]
        if not rays:
            continue
        print(
            "  %-8s rays %9d  nodes/ray %.1f tris/ray %.2f | node loop: %.0f%% of lanes, tri loop: %.0f%%, lanes holding a ray per round: %.0f%% | wave iterations per ray: node %.2f tri %.2f rounds %.2f"
            % (
                kind,
                rays,
                nodes / rays,
                tris / rays,
                100 * nodes / max(s["inner_slots"][k], 1),
                100 * tris / max(s["tri_slots"][k], 1),
                100 * s["busy_rounds"][k] / max(s["round_slots"][k], 1),
                s["inner_slots"][k] / 64 / rays * 64,
                s["tri_slots"][k] / 64 / rays * 64,
                s["round_slots"][k] / 64 / rays * 64,
            )
        )
