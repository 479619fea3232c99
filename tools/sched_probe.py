#!/usr/bin/env python3
"""Scheduler statistics of the counted trace kernel on the benchmark scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = crt.Scene.load_file(os.path.join(ROOT, "tests/golden/scenes/final_frame0042.crtscene")); s.set_resolution(1920, 1080)
ctx = crt.Context(0); ctx.upload(s)
if os.environ.get("PROBE_PART"):      # "part,parts": one rank's share of the frame
    part, parts = (int(v) for v in os.environ["PROBE_PART"].split(","))
    ctx.set_partition(part, parts, 8)
ctx.render_pass(0, spp); ctx.clear()      # first pass measures the tile costs the counted pass is scheduled by
ctx.render_pass_counted(0, spp)
print("counted kernel ms", ctx.last_kernel_ms())
st = ctx.stats(); print(st); print("scan", ctx.scan_check())
rays = st["closest_rays"] + st["shadow_rays"]
tot = 0
sched = ctx.schedule_stats()
cyc_total = sum(c for _, _, c in sched.values())  # operations 0..5
for k, (turns, lanes, cyc) in sched.items():
    tot += turns
    print(f"{k:9s} turns {turns:12d} lanes/turn {lanes / max(turns, 1):6.2f}  turns per 64 rays {turns * 64 / rays:8.2f}  lanes per ray {lanes / rays:6.2f}  cycles/turn {cyc / max(turns, 1):8.0f}  share {100 * cyc / max(cyc_total, 1):5.1f}%")
print("total turns per 64 rays", tot * 64 / rays)
raw = ctx.stats  # noqa
import ctypes
from cudaraytracing_amd import _capi as capi
st2 = capi.Stats(); capi.lib().crt_get_stats(ctx._h, ctypes.byref(st2))
pt = sched["path"][0]
print("shading sections cycles/turn: lights %.0f  attrs %.0f  textures %.0f  light-sample %.0f  bsdf %.0f  next-ray %.0f" % tuple(st2.path_cycles[k] / max(pt, 1) for k in range(6)))
votes = max(st2.sched_turns[6], 1)
print("traversal lanes at a vote: idle %.1f  waiting to post %.1f" % (st2.sched_lanes[6] / votes, st2.sched_lanes[7] / votes))
print("queue fill at a vote: ready queries %.1f  slots waiting for shading %.1f" % (st2.sched_cycles[6] / votes, st2.sched_cycles[7] / votes))
print("regrouping: votes %d (%.2f per 64 rays), cycles per vote %.0f; parks %d with %.1f queries each; take-overs %d with %.1f queries each" % (
    st2.sched_turns[6], st2.sched_turns[6] * 64 / rays, st2.sched_cycles[8] / votes, st2.sched_turns[8], st2.sched_lanes[8] / max(st2.sched_turns[8], 1),
    st2.sched_turns[9], st2.sched_lanes[9] / max(st2.sched_turns[9], 1)))
print("exchange turns: lanes that wanted a query %d, got one %d" % (st2.sched_lanes[4], st2.sched_cycles[5]))
print("wave lifetime after the pixel queue ran dry (shading waves): %.1f %%" % (100.0 * st2.path_cycles[7] / max(st2.path_cycles[6], 1)))
ctx.clear(); ctx.render_pass(0, spp); print("plain kernel ms", ctx.last_kernel_ms())
