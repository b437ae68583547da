#!/usr/bin/env python3
"""GPU box: one sthip option swept on the bench frame — per-kernel ms of a step (time_kernels) and the frame's bytes against the first value.
usage: [STHIP_SWEEP_FLAGS=flag,flag] tools/option_sweep.py <scene> <option> <value> [value ...] [-- other=value ...]"""
import os
os.environ.setdefault("STHIP_STRICT_FLAGS", "1")  # a misspelt --bdptFlag name is an error in a tool that measures (the mirror ignores it, as upstream does)
import sys

import numpy as np

sys.path.insert(0, ".")
from stratum_amd import camera, scenes  # noqa: E402
from stratum_amd.bdpt import BDPT  # noqa: E402

args = sys.argv[1:]
fixed = []
if "--" in args:
    fixed = args[args.index("--") + 1 :]
    args = args[: args.index("--")]
name, option, values = args[0], args[1], [int(v, 0) for v in args[2:]]
sc, cam = scenes.SCENES[name]()
frame = camera.Frame(1920, 1080, cam["fovy"], cam["eye"], cam["target"])
r = BDPT(0, args={"bdptFlag": os.environ["STHIP_SWEEP_FLAGS"].split(",")} if os.environ.get("STHIP_SWEEP_FLAGS") else None)  # e.g. STHIP_SWEEP_FLAGS=neereservoirs
for a in fixed:
    k, v = a.split("=")
    r.set_option(k, int(v, 0))
ref = None
uploaded = False
for v in values:
    r.set_option(option, v)
    if not uploaded or option in ("wide_bvh", "embed_leaves", "bvh_builder", "lds_stack_levels", "treetop"):
        r.update(sc)
        uploaded = True
    for _ in range(3):
        out = r.render(frame, 0, 1, aovs=False)
    img = np.array(out["radiance"], copy=True)
    r.set_option("time_kernels", 1)
    acc = {}
    n = 10
    for s in range(n):
        r.render(frame, s, 1, aovs=False)
        st = r.stats()
        for k in ("ms_trace", "ms_trace_primary", "ms_shade", "ms_total"):
            acc[k] = acc.get(k, 0.0) + st[k]
    r.set_option("time_kernels", 0)
    if ref is None:
        ref = img
    same = bool(np.array_equal(ref.view(np.uint32), img.view(np.uint32)))
    print("%s=%d  trace %.3f primary %.3f shade %.3f all %.3f ms/step | identical to the first: %s" % (option, v, acc["ms_trace"] / n, acc["ms_trace_primary"] / n, acc["ms_shade"] / n, acc["ms_total"] / n, same), flush=True)
