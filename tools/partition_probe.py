#!/usr/bin/env python3
"""Trace-kernel time of one rank's share of the benchmark frame for 1, 2, 4 and 8 ranks (run on one GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = crt.Scene.load_file(os.path.join(ROOT, "tests/golden/scenes/final_frame0042.crtscene")); s.set_resolution(1920, 1080)
ctx = crt.Context(0); ctx.upload(s)
base = None
for parts in [int(x) for x in os.environ.get("PROBE_PARTS", "1,2,4,8").split(",")]:
    worst = 0.0
    for part in range(parts):
        ctx.set_partition(part, parts, 8)
        ctx.clear(); ctx.render_pass(0, spp); ctx.clear(); ctx.render_pass(0, spp)
        worst = max(worst, ctx.last_kernel_ms())
    base = base or worst
    print(f"{parts} ranks: slowest rank's kernel {worst:8.2f} ms, efficiency {base / (worst * parts):.2f}", flush=True)
