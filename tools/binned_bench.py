#!/usr/bin/env python3
"""The four-level-quality line of bench.py on its own (bench.measure_binned): tools/binned_bench.py [blocks per step] [instances] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import bench  # noqa: E402
from dsrc_amd.config import Config  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1800
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
print(json.dumps(bench.measure_binned(Config.from_levels(3, 2), 0, blocks // P, P, steps)))
