#!/usr/bin/env python3
"""Trace-kernel time of consecutive passes: the first hands tiles out in storage order, the later ones costliest first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = crt.Scene.load_file(os.path.join(ROOT, "tests/golden/scenes/final_frame0042.crtscene")); s.set_resolution(1920, 1080)
ctx = crt.Context(0); ctx.upload(s)
for k in range(4):
    ctx.clear(); ctx.render_pass(0, spp)
    print(f"frame {k}: kernel ms {ctx.last_kernel_ms():.2f}", flush=True)
