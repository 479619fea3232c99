#!/usr/bin/env python3
"""The walk over a BVH of the scene nodes (Context.set_tlas) beside the node scan: primary hits, image difference and
kernel time per scene.  usage: tlas_probe.py [spp]"""
import os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cudaraytracing_amd as crt
from conftest import scene_path
from cudaraytracing_amd.synthetic import instance_grid_xml as _instance_grid_xml
from test_gpu_parity import psnr
import oracle_lib as orc

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def run(scene, tlas, srgb):
    ctx = crt.Context(0)
    ctx.set_tlas(tlas)
    ctx.upload(scene)
    hits = ctx.primary_hits(0)
    ctx.render_pass(0, spp); ctx.clear(); ctx.render_pass(0, spp)
    ms = ctx.last_kernel_ms()
    st = ctx.stats()
    ctx.convert(1.0, srgb)
    img = ctx.read_image()
    ctx.close()
    return hits, img, ms, st["closest_rays"] + st["shadow_rays"]


cases = [(n, 640, 360) for n in ("proj1_ball", "proj1_snowman", "proj4_box", "proj5_box", "proj5_glasspot", "proj7_checkerboard", "proj13_box", "final_frame0042")]
for name, w, h in cases + [("grid14", 640, 360), ("grid40", 1280, 720)]:
    with tempfile.TemporaryDirectory() as d:
        if name.startswith("grid"):
            s = crt.Scene.load_xml(_instance_grid_xml(pathlib.Path(d), int(name[4:])), d)
        else:
            s = crt.Scene.load_file(scene_path(name))
        s.set_resolution(w, h)
        flat = os.path.join(d, "s.crtscene")
        s.save_file(flat)
        o = orc.OracleScene(flat)
    orc.set_node_cull(True)
    relaxed = orc.convert(o.render(0, spp)[0], o.srgb, 1.0)
    orc.set_node_cull(False)
    (n0, f0, t0), img0, ms0, rays0 = run(s, False, s.srgb)
    (n1, f1, t1), img1, ms1, rays1 = run(s, True, s.srgb)
    both = (n0 >= 0) & (n1 >= 0)
    rel = np.abs(t0[both] - t1[both]) / np.maximum(np.abs(t0[both]), 1e-30)
    print(f"{name:20s} {s.desc().node_count:5d} nodes  scan {ms0:8.2f} ms  tlas {ms1:8.2f} ms  rays {rays0} / {rays1}  "
          f"node!= {int((n0 != n1).sum())}  face!= {int((f0 != f1).sum())}  max rel dt {rel.max() if rel.size else 0:.2e}  "
          f"psnr {psnr(img0, img1):.2f} dB  pixels!= {int((img0 != img1).any(axis=-1).sum())} of {w * h}  "
          f"| against the oracle culling by entry: psnr {psnr(relaxed, img1):.2f} dB  pixels!= {int((relaxed != img1).any(axis=-1).sum())}", flush=True)
