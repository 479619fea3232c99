#!/usr/bin/env python3
"""Randomised parity sweep: every fixture scene at random sizes, sample counts, pass numbers and partitions,
GPU accumulators against the CPU oracle, bit for bit.  (A wider net than the test suite; run on a GPU box.)"""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc
import cudaraytracing_amd as crt
from cudaraytracing_amd.distributed import owned_rows

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = crt.Context(0)
bad = 0
for path in sorted(glob.glob(os.path.join(ROOT, "tests/golden/scenes/*.crtscene"))):
    name = os.path.basename(path)
    for _ in range(rounds):
        w, h = int(rs.randint(9, 420)), int(rs.randint(9, 260))
        spp, p0, passes = int(rs.randint(1, 6)), int(rs.randint(0, 40)), int(rs.randint(1, 3))
        parts = int(rs.choice([1, 1, 2, 3, 5])); part = int(rs.randint(0, parts)); strip = int(rs.choice([8, 16, 24]))
        s = crt.Scene.load_file(path); s.set_resolution(w, h)
        o = orc.OracleScene(path); o.set_resolution(w, h)
        ctx.upload(s); ctx.set_partition(part, parts, strip); ctx.clear()
        ref = o.new_accumulators()
        rows = owned_rows(part, parts, h, strip)
        for p in range(p0, p0 + passes):
            ctx.render_pass(p, spp)
            for y0 in sorted(set(int(r) // strip * strip for r in rows)):
                o.render(p, spp, acc=ref, rows=(y0, min(y0 + strip, h)))
        acc = ctx.read_accumulators()
        ok = all(np.array_equal(acc[k][rows], ref[k][rows], equal_nan=True) for k in range(3))
        # rows of other ranks stay untouched
        other = np.setdiff1d(np.arange(h), rows)
        ok = ok and all(not acc[k][other].any() for k in range(3))
        # the counted pass scans every node and checks its box hits against the candidate mask of the plain pass
        ctx.clear(); ctx.render_pass_counted(p0, spp)
        chk = ctx.scan_check()
        ok = ok and chk["violations"] == 0
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {name:28s} {w}x{h} spp {spp} passes {p0}..{p0 + passes - 1} part {part}/{parts} strip {strip}", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
