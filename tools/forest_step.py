#!/usr/bin/env python3
"""Config 5 alone (bench.py's other_workloads.forest record), for a rocprofv3 --kernel-trace --stats run of its own:
profiles/rNN/forest_kernel_stats.csv. usage (GPU box): python tools/forest_step.py"""
import json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.forest_record(0)))
