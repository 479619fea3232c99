#!/usr/bin/env python3
"""Every frame of the reference's scenes/final sequence on the GPU: accumulators and ray counts against the CPU oracle
(bit for bit), and the counted kernel's check of the candidate mask (every scene-node box the full scan finds hit must
be a candidate of the mask the plain kernels scan by).

Bake first, in the build container (the GPU box has no reference tree):
    python tools/all_final_frames.py bake  build_variants/final_frames
then on the GPU box:
    python tools/all_final_frames.py check build_variants/final_frames [spp] [width] [height]
    python tools/all_final_frames.py tlas  build_variants/final_frames [spp] [width] [height]
(tlas: the opt-in walk over a BVH of the scene nodes, which is not bit-exact - per frame its PSNR against the oracle as
the reference has it and against the oracle that culls boxes by entry distance, and the share of identical pixels)"""
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
files = sorted(glob.glob(os.path.join(folder, "*.crtscene")))
if mode == "tlas":
    def psnr(a, b):
        mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
        return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    ctx = crt.Context(0); ctx.set_tlas(True)
    threads = int(os.environ.get("OMP_NUM_THREADS", "8"))
    t0 = time.time(); worst = (99.0, 99.0, 1.0); sums = np.zeros(3)
    for f in files:
        s = crt.Scene.load_file(f); s.set_resolution(W, H)
        o = orc.OracleScene(f); o.set_resolution(W, H)
        ctx.upload(s); ctx.clear(); ctx.render_pass(0, spp); ctx.stats(); ctx.convert(1.0, s.srgb)
        got = ctx.read_image()
        strict = orc.convert(o.render(0, spp, threads=threads)[0], o.srgb, 1.0)
        orc.set_node_cull(True)
        try:
            entry = orc.convert(o.render(0, spp, threads=threads)[0], o.srgb, 1.0)
        finally:
            orc.set_node_cull(False)
        row = (psnr(got, strict), psnr(got, entry), float((got == entry).all(axis=-1).mean()))
        worst = tuple(min(a, b) for a, b in zip(worst, row)); sums += row
        print(f"{os.path.basename(f)}: psnr against the oracle {row[0]:6.2f} dB, against the oracle culling by entry {row[1]:6.2f} dB, identical pixels {100 * row[2]:6.2f} %  ({time.time() - t0:.0f} s)", flush=True)
    n = len(files)
    print(f"{n} frames at {W}x{H}, {spp} spp: mean {sums[0] / n:.2f} dB / {sums[1] / n:.2f} dB / {100 * sums[2] / n:.2f} %, worst {worst[0]:.2f} dB / {worst[1]:.2f} dB / {100 * worst[2]:.2f} %")
    sys.exit(0)
ctx = crt.Context(0)
bad = 0; checks = 0; t0 = time.time()
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
