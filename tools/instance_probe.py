#!/usr/bin/env python3
"""Scene-node scan on a scene of many instances of one mesh (n x n nodes sharing a BVH): kernel time with the scan
going by groups of nodes, with the node-by-node scan of the reference, and with the opt-in walk over a BVH of the
nodes (not bit-exact).  usage: instance_probe.py [n_side] [spp]"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cudaraytracing_amd as crt
from cudaraytracing_amd.synthetic import instance_grid_xml as _instance_grid_xml
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 40
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
with tempfile.TemporaryDirectory() as d:
    s = crt.Scene.load_xml(_instance_grid_xml(pathlib.Path(d), n_side), d)
s.set_resolution(1280, 720)
for label, env in (("scan by groups of nodes", {}), ("node-by-node scan", {"CRT_NO_SCAN_MASK": "1"}), ("BVH over the nodes (opt-in)", {"CRT_TLAS": "1"})):
    os.environ.pop("CRT_NO_SCAN_MASK", None); os.environ.pop("CRT_TLAS", None); os.environ.update(env)
    ctx = crt.Context(0); ctx.upload(s)
    ctx.render_pass(0, spp); ctx.clear(); ctx.render_pass(0, spp)
    st = ctx.stats(); rays = st["closest_rays"] + st["shadow_rays"]
    ms = ctx.last_kernel_ms()
    print(f"{s.desc().node_count} nodes, 1280x720, {spp} spp, {label:26s}: {ms:8.2f} ms  {rays / ms * 1e-3:8.1f} Mrays/s", flush=True)
    ctx.close()
