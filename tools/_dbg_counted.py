import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
s = crt.Scene.load_file(os.path.join(ROOT, "tests/golden/scenes/final_frame0042.crtscene")); s.set_resolution(192, 108)
ctx = crt.Context(0); ctx.upload(s)
ctx.render_pass(0, 2); a = [x.copy() for x in ctx.read_accumulators()]
ctx.clear(); ctx.render_pass_counted(0, 2); b = [x.copy() for x in ctx.read_accumulators()]
ctx.clear(); ctx.render_pass_counted(0, 1); c = [x.copy() for x in ctx.read_accumulators()]
ctx.clear(); ctx.render_pass(0, 1); d = [x.copy() for x in ctx.read_accumulators()]
np.savez(os.path.join(ROOT, "gpurun_out/dbg_counted.npz"), a0=a[0], a1=a[1], a2=a[2], b0=b[0], b1=b[1], b2=b[2], c0=c[0], c1=c[1], c2=c[2], d0=d[0], d1=d[1], d2=d[2])
print("saved")
