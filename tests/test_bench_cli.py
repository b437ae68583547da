"""bench.py's own rank launcher (`python bench.py --gpus N` without torchrun), exercised without a GPU: the parent must
start N children before touching the GPU itself, and a failing rank must fail the whole run (here every rank fails,
because the product path has no CPU fallback)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch

    return not torch.cuda.is_available()


def test_gpus_n_spawns_ranks_and_propagates_failure():
    if not _no_gpu():
        import pytest

        pytest.skip("needs a box without a GPU: with one the children would run the whole benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    # both ranks were started as children (each says why it cannot run) and the parent reports the failing rank
    assert p.stderr.count("bench.py needs a GPU") >= 1
    assert "a rank exited with code" in p.stderr
    assert '"metric"' not in p.stdout


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE (2) != --gpus (4)" in p.stderr


def test_unknown_flag_names_are_refused_by_the_tools_and_ignored_by_the_mirror():
    """`--bdptFlag` names are upstream's (`lightvertexcache`, not `lvc`); BDPT.cpp:94-127 ignores a name it does not know and
    so does the Python mirror — which is how two timing scripts once measured plain connections under a cache label. bench.py
    refuses such a name before anything else happens, and no script under tools/ passes one."""
    import glob
    import re

    sys.path.insert(0, ROOT)
    from stratum_amd.bdpt import known_flag

    assert known_flag("lightvertexcache") and known_flag("~nee") and known_flag("!MIS") and not known_flag("lvc") and not known_flag("")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bdpt-flag", "lvc"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "unknown --bdpt-flag ['lvc']" in p.stderr and '"metric"' not in p.stdout
    seen = 0
    for path in glob.glob(os.path.join(ROOT, "tools", "**", "*"), recursive=True):
        if not os.path.isfile(path) or not path.endswith((".sh", ".py", ".md")):
            continue
        for name in re.findall(r"--bdpt-flag[ =]([~!]?[A-Za-z]+)", open(path).read()):
            seen += 1
            assert known_flag(name), (path, name)
    assert seen > 10
