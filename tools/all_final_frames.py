#!/usr/bin/env python3
"""Every frame of the reference's scenes/final sequence on the GPU: accumulators and ray counts against the CPU oracle
(bit for bit), and the counted kernel's check of the candidate mask (every scene-node box the full scan finds hit must
be a candidate of the mask the plain kernels scan by).

Bake first, in the build container (the GPU box has no reference tree):
    python tools/all_final_frames.py bake  build_variants/final_frames
then on the GPU box:
    python tools/all_final_frames.py check build_variants/final_frames [spp] [width] [height]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cudaraytracing_amd as crt

mode, folder = sys.argv[1], sys.argv[2]
if mode == "bake":
    REF = "/root/reference"
    os.makedirs(folder, exist_ok=True)
    for xml in sorted(glob.glob(f"{REF}/scenes/final/frame*.xml")):
        s = crt.Scene.load_xml(xml, REF)
        s.save_file(os.path.join(folder, os.path.basename(xml)[:-4] + ".crtscene"))
    print("baked", len(glob.glob(os.path.join(folder, "*.crtscene"))), "frames")
    sys.exit(0)

import oracle_lib as orc
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1280
H = int(sys.argv[5]) if len(sys.argv) > 5 else 720
ctx = crt.Context(0)
bad = 0; checks = 0; t0 = time.time()
files = sorted(glob.glob(os.path.join(folder, "*.crtscene")))
for f in files:
    s = crt.Scene.load_file(f); s.set_resolution(W, H)
    o = orc.OracleScene(f); o.set_resolution(W, H)
    ctx.upload(s)
    ctx.clear(); ctx.render_pass(0, spp)
    acc = ctx.read_accumulators(); st = ctx.stats()
    ctx.clear(); ctx.render_pass_counted(0, spp)
    acc_c = ctx.read_accumulators(); chk = ctx.scan_check()
    ref, cnt = o.render(0, spp, count=True, threads=int(os.environ.get("OMP_NUM_THREADS", "8")))
    ok = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(acc, ref)) and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(acc, acc_c))
    ok = ok and (st["closest_rays"], st["shadow_rays"]) == (cnt["closest_rays"], cnt["shadow_rays"]) and chk["violations"] == 0
    checks += chk["checks"]; bad += 0 if ok else 1
    print(f"{os.path.basename(f)}: {'ok' if ok else 'MISMATCH'}  rays {st['closest_rays']}+{st['shadow_rays']}  mask checks {chk['checks']} violations {chk['violations']}  ({time.time() - t0:.0f} s)", flush=True)
print(f"{len(files)} frames, {bad} mismatching, {checks} box hits checked against the candidate mask")
sys.exit(1 if bad else 0)
