"""Generates the committed golden fixtures from the CPU oracle.

The reference has no tests, golden vectors or runnable CPU path, so these
fixtures pin the *restatement*: they freeze the oracle's output so that a later change of the
oracle (or of the arithmetic contract) shows up as a diff. Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as orc  # noqa: E402
from stratum_amd import camera, scenes, wire  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sc, cam = scenes.cornell_box()
    o = orc.OracleScene(sc)
    fr = camera.Frame(256, 256, cam["fovy"], cam["eye"], cam["target"])
    out = o.render(fr, wire.default_push_constants(256, 256, sc.light_count))
    np.savez_compressed(
        os.path.join(HERE, "cornell_256_seed0.npz"),
        radiance=out["radiance"],
        albedo=out["albedo"],
        instance_primitive_index=out["visibility"]["instance_primitive_index"],
        packed_normal=out["visibility"]["packed_normal"],
        depth_z=out["depth"]["z"],
        ray_count=out["ray_count"],
    )
    # a small ray/hit batch of the traversal contract on the Cornell box
    rng = np.random.RandomState(1234)
    rays = np.zeros(4096, wire.Ray)
    rays["origin"] = rng.uniform(-0.95, 0.95, (4096, 3))
    d = rng.normal(size=(4096, 3))
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmax"] = np.inf
    hits, _ = o.trace(rays, brute=True)
    np.savez_compressed(os.path.join(HERE, "cornell_rays.npz"), rays=rays, hits=hits)


if __name__ == "__main__":
    main()
