#!/usr/bin/env python3
"""First-contact GPU probe: self tests, primary hits and 1-spp renders against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cudaraytracing_amd as crt
import oracle_lib as orc

G = os.path.join(ROOT, "tests", "golden", "scenes")
ctx = crt.Context(0)
rs = np.random.RandomState(1)
x = (rs.rand(100000).astype(np.float32) * 8 - 1); y = rs.rand(100000).astype(np.float32) * 40
for fn, name in enumerate(["sin", "cos", "exp", "pow", "atan2", "asin", "acos", "tan"]):
    xx = x if fn not in (5, 6) else np.clip(x / 4, -1, 1)
    a = ctx.math_selftest(fn, xx, y); b = orc.math(fn, xx, y)
    print("math", name, "mismatch", int(np.sum(a.view(np.uint32) != b.view(np.uint32)) - np.sum(np.isnan(a) & np.isnan(b) & (a.view(np.uint32) != b.view(np.uint32)))))
seeds = np.array([0, 1, 921599, 2073599, 12345], np.uint32)
for sub in (0, 1, 7, 100):
    a = ctx.rng_selftest(seeds, np.full(5, sub, np.uint32), 8); b = orc.xorwow(seeds, np.full(5, sub, np.uint32), 8)
    print("rng subseq", sub, "equal", bool(np.array_equal(a, b)))

def cmp(name, w, h, spp):
    path = os.path.join(G, name + ".crtscene")
    s = crt.Scene.load_file(path); s.set_resolution(w, h)
    o = orc.OracleScene(path); o.set_resolution(w, h)
    ctx.upload(s)
    t0 = time.time(); n1, f1, t1 = ctx.primary_hits(0); tg = time.time() - t0
    n2, f2, t2 = o.primary_hits(0)
    print(f"{name} {w}x{h}: primary node mismatch {int(np.sum(n1 != n2))} face {int(np.sum(f1 != f2))} t {int(np.sum(t1.view(np.uint32) != t2.view(np.uint32)))} ({tg*1e3:.1f} ms)")
    ctx.clear()
    t0 = time.time(); ctx.render_pass(0, spp); tg = time.time() - t0
    g = ctx.read_accumulators()
    t0 = time.time(); c, cnt = o.render(0, spp, count=True); tc = time.time() - t0
    st = ctx.stats()
    for k, nm in enumerate(["results", "normals", "albedos"]):
        eq = (g[k] == c[k]) | (np.isnan(g[k]) & np.isnan(c[k]))
        print(f"   {nm}: pixels differing {int(np.sum(~eq.all(axis=2)))} / {w*h}")
    print(f"   rays gpu {st['closest_rays']}+{st['shadow_rays']} cpu {cnt['closest_rays']}+{cnt['shadow_rays']}; kernel {ctx.last_kernel_ms():.2f} ms wall {tg*1e3:.1f} ms; cpu {tc:.2f}s")

for name in sys.argv[1:] or ["proj1_ball", "proj4_box", "teapot_low_box", "proj5_box", "proj7_checkerboard", "proj13_box", "final_frame0042"]:
    cmp(name, 320, 180, 2)
