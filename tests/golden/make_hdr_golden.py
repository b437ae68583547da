"""Generates tests/golden/hdr_writer.npz: input images and the bytes the REFERENCE's HDR writer produces for them.

The reference exports HDR frames with the stb_image_write v1.16 it vendors (src/Node/BDPT.cpp:335 ->
src/extern/stb_image_write.h). `make -C oracle ref` compiles that header as it lies under /root/reference into
oracle/_ref/libstbiw_ref.so; this script calls its stbi_write_hdr on seeded images and stores inputs + output bytes.
Run in the build container only (the reference does not travel):  python tests/golden/make_hdr_golden.py
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def cases():
    rng = np.random.default_rng(20240607)
    out = {}
    # narrow rows are stored flat (width < 8)
    out["flat_5x3"] = (rng.random((3, 5, 4)) * 4).astype(np.float32)
    # smallest run-length coded width; noise -> literals only
    out["noise_8x4"] = (rng.random((4, 8, 4)) * np.float32(100)).astype(np.float32)
    # long constant stretches (> 127 -> split run packets), black pixels (< 1e-32 -> 0,0,0,0), big dynamic range
    img = np.zeros((6, 300, 4), np.float32)
    img[0, :, :3] = 0.5
    img[1, :150, :3] = [1.0, 2.0, 3.0]
    img[2, ::7, :3] = 1000.0
    img[3, :, 0] = np.linspace(0, 1e4, 300, dtype=np.float32)
    img[4, :, :3] = (rng.random((300, 3)) * 1e-3).astype(np.float32)
    img[5, 10:290, 1] = 1e-35
    out["runs_300x6"] = img
    # > 128 distinct values in a row -> split literal packets; runs of exactly 2 and 3
    img = np.zeros((2, 200, 4), np.float32)
    img[0, :, :3] = (np.arange(200, dtype=np.float32)[:, None] % 251 + 1) / 256
    img[1, :, :3] = np.repeat(np.arange(100, dtype=np.float32) + 1, 2)[:, None] / 128
    img[1, 50:53, :3] = 0.75
    out["literals_200x2"] = img
    return out


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstbiw_ref.so"))
    ref.stbi_write_hdr.restype = C.c_int
    ref.stbi_write_hdr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    blob = {}
    with tempfile.TemporaryDirectory() as d:
        for name, img in cases().items():
            path = os.path.join(d, name + ".hdr")
            assert ref.stbi_write_hdr(path.encode(), img.shape[1], img.shape[0], 4, img.ctypes.data) == 1
            blob[name + "_in"] = img
            blob[name + "_hdr"] = np.frombuffer(open(path, "rb").read(), np.uint8)
            print(name, img.shape, len(blob[name + "_hdr"]), "bytes")
    np.savez_compressed(os.path.join(HERE, "hdr_writer.npz"), **blob)


if __name__ == "__main__":
    main()
