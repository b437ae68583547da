"""Makes tests/golden/fog_sphere.npz: a NanoVDB float grid (fog-volume sphere) written by the NanoVDB 32.3.3 the reference
vendors, with known answers read through the reference's own PNanoVDB accessors (oracle/nvdb_ref.cpp -> oracle/_ref/nvdb_ref,
built by `make -C oracle ref` in the container that has /root/reference). The fixture is data: grid bytes + answers."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    with tempfile.TemporaryDirectory() as d:
        grid, probe = os.path.join(d, "fog.nvdb"), os.path.join(d, "probe.bin")
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "nvdb_ref"), "10", "0.08", "3", grid, probe])
        g = np.fromfile(grid, dtype=np.uint8)
        raw = np.fromfile(probe, dtype=np.int32)
    ni, nf, n, m = raw[:4]
    ints = raw[4 : 4 + ni]
    floats = raw[4 + ni : 4 + ni + nf].view(np.float32)
    out = {
        "grid": g,
        "bbox_min": ints[0:3],
        "bbox_max": ints[3:6],
        "grid_type": ints[6],
        "root_max": floats[0],
        "coords": ints[8 : 8 + 3 * n].reshape(n, 3),
        "values": floats[1 : 1 + n],
        "map": floats[1 + n : 1 + n + 15 * m].reshape(m, 5, 3),  # point, world_to_indexf, world_to_index_dirf, index_to_worldf(of 2nd), index_to_world_dirf(of 3rd)
    }
    path = os.path.join(ROOT, "tests", "golden", "fog_sphere.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", int((out["values"] > 0).sum()), "probes inside the fog")


if __name__ == "__main__":
    sys.exit(main())
