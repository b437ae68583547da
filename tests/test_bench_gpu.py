"""bench.py on the GPU box: a two-rank rehearsal (both ranks on device 0, gloo in place of RCCL) must deliver the frame the one-rank
run delivers — radiance AND the G-buffer, which travels with it (sthip_pack_tiles -> gather -> sthip_assemble_tiles_bytes) — and
the JSON line must carry what the driver and the judge read (sustained, exchange bytes)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--scene", "cornell_box", "--width", "256", "--height", "128", "--steps", "2", "--warmup", "1", "--reps", "1", "--no-cpu-baseline", "--no-ceilings", "--no-last-ray-filter",
         "--no-other-workloads", "--scaling", "strong", "--strong-seeds", "2"]


def _run(tmp_path, gpus, extra_env=None, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env or {})
    dump = str(tmp_path / ("frame%d" % gpus))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dump-frame", dump] + SMALL + list(extra), env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), np.load(dump + ".npz")


@pytest.mark.gpu
def test_two_rank_rehearsal_delivers_the_one_rank_frame(tmp_path):
    one, f1 = _run(tmp_path, 1, extra=["--sustained-seconds", "0.2"])
    two, f2 = _run(tmp_path, 2, {"STHIP_BENCH_ONE_DEVICE": "1", "STHIP_BENCH_BACKEND": "gloo"}, extra=["--sustained-seconds", "0.2"])
    for k in ("radiance", "albedo", "visibility", "depth", "prev_uv"):
        assert np.array_equal(f1[k].view(np.uint32), f2[k].view(np.uint32)), k
    assert f1["albedo"].any() and f1["visibility"].any()
    assert one["config"]["rays_per_step"] == two["config"]["rays_per_step"]
    assert one["config"]["exchange_bytes_per_step"] == 0
    tiles = ((256 + 63) // 64) * ((128 + 31) // 32)
    assert two["config"]["exchange_bytes_per_step"] == 2 * ((tiles + 1) // 2) * 64 * 32 * 64  # 64 B per slot: radiance 16 + albedo 16 + depth 16 + visibility 8 + prev-uv 8
    for line in (one, two):
        s = line["sustained"]
        assert s["seconds"] >= 0.2 and s["steps"] >= 2 and s["value"] > 0
        assert line["value"] > 0 and line["scaling"] == "strong"


@pytest.mark.gpu
def test_two_rank_seed_split_rehearsal(tmp_path):
    """--shard seeds: every rank the whole frame for its share of the step's seeds, one sum-reduce of the accumulation buffers
    (SURVEY 8e 'replicas + sum-reduce'): the one-rank frame up to the order of the floating-point sum, the same rays."""
    one, f1 = _run(tmp_path, 1, extra=["--sustained-seconds", "0"])
    two, f2 = _run(tmp_path, 2, {"STHIP_BENCH_ONE_DEVICE": "1", "STHIP_BENCH_BACKEND": "gloo"}, extra=["--sustained-seconds", "0", "--shard", "seeds"])
    assert np.array_equal(f1["radiance"][..., 3], f2["radiance"][..., 3]) and f1["radiance"][..., 3].max() == 2
    np.testing.assert_allclose(f2["radiance"][..., :3], f1["radiance"][..., :3], rtol=1e-6, atol=1e-7)
    for k in ("albedo", "visibility", "depth", "prev_uv"):
        assert np.array_equal(f1[k].view(np.uint32), f2[k].view(np.uint32)), k
    assert one["config"]["rays_per_step"] == two["config"]["rays_per_step"]
    assert two["config"]["exchange_bytes_per_step"] == 2 * 256 * 128 * 16 and "seed-split" in two["config"]["parallelism"] and one["sustained"] is None
