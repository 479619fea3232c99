#!/usr/bin/env python3
"""Time the trace kernel on the benchmark scene for a grid of scheduler thresholds."""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = crt.Scene.load_file(os.path.join(ROOT, "tests/golden/scenes/final_frame0042.crtscene")); s.set_resolution(1920, 1080)
boxes = [int(v) for v in os.environ.get("SWEEP_BOX", "16,24,32").split(",")]
specials = [int(v) for v in os.environ.get("SWEEP_SPECIAL", "6,8,12").split(",")]
heavies = [int(v) for v in os.environ.get("SWEEP_HEAVY", "1,24,32,40,48").split(",")]
for b, sp, h in itertools.product(boxes, specials, heavies):
    os.environ["CRT_SCHED_BOX_MIN"] = str(b); os.environ["CRT_SCHED_SPECIAL_MIN"] = str(sp); os.environ["CRT_SCHED_HEAVY_MIN"] = str(h)
    ctx = crt.Context(0); ctx.upload(s)
    ctx.render_pass(0, spp); ctx.clear(); ctx.render_pass(0, spp)
    print(f"box {b:3d} special {sp:3d} heavy {h:3d}  kernel ms {ctx.last_kernel_ms():8.2f}", flush=True)
    del ctx
