#!/usr/bin/env python3
"""Bake the scene fixtures under tests/golden/scenes/ from the reference's scene,
model and texture DATA files (XML / OBJ / PNG under /root/reference) with this
project's own loader.  Run in the build container only; the GPU box reads the
baked .crtscene files and never the reference tree."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "scenes")
SCENES = {
    "proj1_ball": f"{REF}/scenes/proj1/ball.xml",
    "proj1_snowman": f"{REF}/scenes/proj1/snowman.xml",
    "proj4_box": f"{REF}/scenes/proj4/box.xml",
    "teapot_low_box": f"{ROOT}/tests/golden/scenes_src/teapot_low_box.xml",
    "proj5_box": f"{REF}/scenes/proj5/box.xml",
    "proj5_plane": f"{REF}/scenes/proj5/plane.xml",
    "proj5_glasspot": f"{REF}/scenes/proj5/glasspot.xml",
    "proj7_checkerboard": f"{REF}/scenes/proj7/checkerboard.xml",
    "proj13_box": f"{REF}/scenes/proj13/box.xml",
    "final_frame0001": f"{REF}/scenes/final/frame0001.xml",
    "final_frame0042": f"{REF}/scenes/final/frame0042.xml",
}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, xml in SCENES.items():
        s = crt.Scene.load_xml(xml, REF)
        path = os.path.join(OUT, name + ".crtscene")
        s.save_file(path)
        d = s.desc()
        print(f"{name}: {d.node_count} nodes, {d.mesh_count} meshes, {d.light_count} lights, "
              f"{d.texture_count} textures -> {os.path.getsize(path) / 1024:.0f} KiB")
