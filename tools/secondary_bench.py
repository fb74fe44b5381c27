#!/usr/bin/env python3
"""bench.py's secondary workloads alone: secondary_bench.py [p2|ld] [reps] -- one line per repetition (A/B runs under env switches)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import torch  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "p2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
for _ in range(reps):
    r = b.secondary_p2_1280(dev) if which == "p2" else b.secondary_ld(dev)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "workload"}, flush=True)
