#!/usr/bin/env python3
"""Time the trace kernel on the benchmark scene for a list of knob settings.

usage: knob_sweep.py SPP "NAME=V,NAME=V" "NAME=V" ...   (names without the CRT_ prefix; "" = defaults)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = os.environ.get("SWEEP_SCENE", "final_frame0042")
w, h = (int(v) for v in os.environ.get("SWEEP_SIZE", "1920x1080").split("x"))
s = crt.Scene.load_file(os.path.join(ROOT, f"tests/golden/scenes/{scene}.crtscene"))
if os.environ.get("SWEEP_COPIES"):      # the synthetic instanced scene: SWEEP_SCENE=proj5_box SWEEP_COPIES=1582 SWEEP_SIZE=3840x2160
    s = s.make_instanced(0, int(os.environ["SWEEP_COPIES"]), 1, w, h)
else:
    s.set_resolution(w, h)
KNOBS = ("SCHED_BOX_MIN", "SCHED_SPECIAL_MIN", "SCHED_EXCHANGE_MIN", "TRAV_WAVES", "SLOTS", "SERVE_BATCH_MIN", "SERVE_LOW_WATER", "SERVE_PATIENCE", "SERVE_PRIORITY", "TILE_INTERLEAVE", "NO_BALANCE", "PAIR_BELOW", "PAIR_LIVE", "STACK_ENTRIES", "FIRST_STEP",
         "REGROUP", "CONTEXTS", "KEEP_BOX", "KEEP_LEAF", "KEEP_NODE", "KEEP_EXCHANGE", "EXCHANGE_FIRST", "BURST_MAX", "DIRECT_START", "FIRST_STEP_IDLE", "SWING_WAVES", "SWING_FILL", "TAIL_WINDOW", "TAIL_LIVE")
for spec in sys.argv[2:] or [""]:
    for k in KNOBS:
        os.environ.pop("CRT_" + k, None)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        assert k in KNOBS, k
        os.environ["CRT_" + k] = v
    ctx = crt.Context(0); ctx.upload(s)
    ctx.render_pass(0, spp); ctx.clear(); ctx.render_pass(0, spp)
    a = ctx.last_kernel_ms()
    ctx.clear(); ctx.render_pass(0, spp)
    print(f"{spec or 'defaults':60s} kernel ms {a:8.2f} {ctx.last_kernel_ms():8.2f}", flush=True)
    ctx.stats()          # raises if the kernel reported a protocol error
    ctx.close()
