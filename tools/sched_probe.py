#!/usr/bin/env python3
"""Scheduler statistics of the counted trace kernel on the benchmark scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = crt.Scene.load_file(os.path.join(ROOT, "tests/golden/scenes/final_frame0042.crtscene")); s.set_resolution(1920, 1080)
ctx = crt.Context(0); ctx.upload(s)
ctx.render_pass_counted(0, spp)
print("counted kernel ms", ctx.last_kernel_ms())
st = ctx.stats(); print(st)
rays = st["closest_rays"] + st["shadow_rays"]
tot = 0
for k, (turns, lanes) in ctx.schedule_stats().items():
    tot += turns
    print(f"{k:9s} turns {turns:12d} lanes/turn {lanes / max(turns, 1):6.2f}  turns per 64 rays {turns * 64 / rays:8.2f}  lanes per ray {lanes / rays:8.2f}")
print("total turns per 64 rays", tot * 64 / rays)
ctx.clear(); ctx.render_pass(0, spp); print("plain kernel ms", ctx.last_kernel_ms())
