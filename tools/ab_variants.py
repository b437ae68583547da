"""Runs tools/ab_options.py once per library variant (GPU box): python tools/ab_variants.py v0,v1 treetop=1 ..."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in sys.argv[1].split(","):
    env = dict(os.environ)
    if v != "product":
        env["STHIP_LIB"] = os.path.join(root, "_variants", v + ".so")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "ab_options.py")] + sys.argv[2:], env=env, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        print("[%s] %s" % (v, line), flush=True)
    if out.returncode != 0:
        print("[%s] FAILED: %s" % (v, out.stderr[-800:]))
