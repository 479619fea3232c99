#!/usr/bin/env python3
"""Every scene file of the reference through the product's front end and through the independent Python restatement
of the reference's loader (tests/loader_oracle.py); prints what differs.  Needs the reference's data files; CPU only.

usage: check_loaders.py [reference root]"""
import functools, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cudaraytracing_amd as crt
import loader_oracle as lo
ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
lo.load_obj = functools.lru_cache(maxsize=None)(lo.load_obj)
files = sorted(glob.glob(ref_root + "/scenes/*/*.xml"))
bad, t0 = 0, time.time()
nodes = meshes = texels = 0
for f in files:
    try:
        ref = lo.load_scene(f, ref_root)
        diffs = lo.compare(crt.Scene.load_xml(f, ref_root), ref)
        nodes += len(ref["nodes"]); meshes += len(ref["meshes"])
        texels += sum(m[k][0] * m[k][1] for m in ref["materials"] for k in ("diffuse", "specular", "refraction", "emission"))
    except Exception as e:
        diffs = [f"{type(e).__name__}: {e}"]
    if diffs:
        bad += 1
        print(os.path.relpath(f, ref_root), f"{len(diffs)} difference(s)")
        for d in diffs[:5]:
            print("    " + d[:200])
print(f"{len(files)} scene files, {nodes} nodes, {meshes} mesh loads, {texels} texels compared bit for bit: "
      f"{bad} file(s) differ ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
