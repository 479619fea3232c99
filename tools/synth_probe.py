#!/usr/bin/env python3
"""The synthetic ten-million-triangle scene (1582 teapot copies, one BVH) on one GPU.

usage: synth_probe.py [spp] [copies] [width] [height]
Prints scene build / upload times, the trace kernel's time and ray rate, and the traversal counters with the
algorithmic bytes they imply (the bandwidth-bound workload of the benchmark plan, at a reduced sample count)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
from bench import algorithmic_work

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 1582
W = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
H = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
src = crt.Scene.load_file(os.path.join(ROOT, "tests/golden/scenes/proj5_box.crtscene"))
t0 = time.perf_counter(); syn = src.make_instanced(0, copies, 1, W, H); t1 = time.perf_counter()
d = syn.desc()
print(f"scene: {d.meshes[0].nf} triangles, generated in {t1 - t0:.2f} s", flush=True)
ctx = crt.Context(0)
t0 = time.perf_counter(); ctx.upload(syn); t1 = time.perf_counter()
print(f"upload (BVH build + re-layout + H2D): {t1 - t0:.1f} s", flush=True)
ctx.render_pass(0, 1); ctx.clear()
ctx.render_pass(0, spp)
ms = ctx.last_kernel_ms(); st = ctx.stats()
rays = st["closest_rays"] + st["shadow_rays"]
print(f"trace kernel {ms:.1f} ms for {spp} spp at {W}x{H}: {rays / ms * 1e-3:.1f} Mrays/s ({rays} rays)", flush=True)
if os.environ.get("SYNTH_COUNTED", "1") != "0":
    ctx.clear(); ctx.render_pass_counted(0, spp)
    c = ctx.stats()
    nbytes, flops = algorithmic_work(c, W * H)
    print("counters", c)
    print(f"algorithmic bytes {nbytes / 1e9:.2f} GB -> {nbytes / ms * 1e-6:.0f} GB/s at the plain kernel's time; flops {flops / 1e9:.1f} G")
    for k, (turns, lanes, cyc) in ctx.schedule_stats().items():
        print(f"{k:9s} turns {turns:12d} lanes/turn {lanes / max(turns, 1):6.2f} cycles/turn {cyc / max(turns, 1):8.0f}")
