"""Generates tests/golden/estimators.npz from the CPU oracle: small Cornell frames of the estimators whose upstream result
depends on thread scheduling and that this project pins to ONE defined order each (light vertex cache,
reservoir reuse through the hash grids, coherent Russian roulette, coherent sampling). The fixture freezes those orders:
a later change of the oracle's definition (or of the HIP path's) shows up as a diff against committed data, without the
oracle in the loop on the GPU side. Run from the repo root:
    python tests/golden/make_estimator_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as orc  # noqa: E402
from stratum_amd import camera, scenes, wire  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, SEEDS = 64, 48, 2

# name -> (flags switched on / off, push-constant overrides)
CASES = {
    "lvc": (["eConnectToLightPaths", "eLVC", "~eDeferShadowRays"], dict(gLightPathCount=2000, gMaxDiffuseVertices=3)),
    "lvc_reservoirs_reuse": (["eConnectToLightPaths", "eLVC", "eLVCReservoirs", "eLVCReservoirReuse", "~eDeferShadowRays"], dict(gLightPathCount=2000, gMaxDiffuseVertices=3, gReservoirM=3, gHashGridBucketCount=4096)),
    "nee_reservoirs_reuse": (["eNEEReservoirs", "eNEEReservoirReuse"], dict(gReservoirM=4, gHashGridBucketCount=4096)),
    "coherent_rr": (["eCoherentRR"], dict(gMinPathVertices=2, gMaxPathVertices=7, gMaxDiffuseVertices=5)),
    "coherent_sampling": (["eCoherentSampling", "ePresampleLights", "eNEEReservoirs", "eConnectToLightPaths", "eLVC", "eLVCReservoirs", "~eDeferShadowRays"],
                          dict(gLightPathCount=2000, gMaxDiffuseVertices=3, gReservoirM=3, gLightPresampleTileSize=64, gLightPresampleTileCount=8)),
}


def flags_of(names):
    bit = {n: i for i, n in enumerate(wire.FLAG_NAMES)}
    f = wire.DEFAULT_SAMPLING_FLAGS
    for n in names:
        if n.startswith("~"):
            f &= ~(1 << bit[n[1:]])
        else:
            f |= 1 << bit[n]
    return f


def push_constants(sc, overrides):
    pc = wire.default_push_constants(W, H, sc.light_count)
    for k, v in overrides.items():
        setattr(pc, k, v)
    return pc


def main():
    sc, cam = scenes.cornell_box()
    o = orc.OracleScene(sc)
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    out = {}
    for name, (names, overrides) in CASES.items():
        res = o.render(fr, push_constants(sc, overrides), flags_of(names), 3, SEEDS)
        out[name + "_radiance"] = res["radiance"]
        out[name + "_ray_count"] = res["ray_count"]
    np.savez_compressed(os.path.join(HERE, "estimators.npz"), **out)


if __name__ == "__main__":
    main()
