#!/usr/bin/env python3
"""Counted pass over the benchmark frame at full size: every scene-node box the full scan finds hit is checked
against the candidate mask the plain kernels scan by (violations must be 0), and both passes must agree bit for bit."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, w, h in (("final_frame0042", 1920, 1080), ("final_frame0001", 1920, 1080), ("proj7_checkerboard", 1280, 720), ("proj5_box", 1280, 720)):
    s = crt.Scene.load_file(os.path.join(ROOT, "tests/golden/scenes", name + ".crtscene")); s.set_resolution(w, h)
    ctx = crt.Context(0); ctx.upload(s)
    ctx.render_pass(0, spp); a = [x.copy() for x in ctx.read_accumulators()]
    ctx.clear(); ctx.render_pass_counted(0, spp); b = ctx.read_accumulators()
    same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))
    print(name, f"{w}x{h} spp {spp}:", ctx.scan_check(), "plain == counted:", same, flush=True)
    del ctx
