#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the CPU oracle (det flavour).

The reference ships no vectors of its own and cannot be built in this image, so these
files pin the oracle against regressions rather than against the reference; the only
reference-derived numbers are the survey anchors in tests/golden/survey_anchors.json."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
S = os.path.join(G, "scenes")


def bvh_digest(b):
    h = hashlib.sha256()
    for k in ("leaf", "start", "count", "left", "right", "next", "elements"):
        h.update(np.ascontiguousarray(b[k]).tobytes())
    h.update((np.ascontiguousarray(b["box"]) + np.float32(0)).tobytes())   # +0 folds -0 into +0
    return h.hexdigest()


if __name__ == "__main__":
    seeds = np.repeat(np.array([0, 1, 921599, 2073599], np.uint32), 3)
    subs = np.tile(np.array([0, 1, 7], np.uint32), 4)
    np.savez_compressed(os.path.join(G, "xorwow_kat.npz"), seeds=seeds, subsequences=subs, words=orc.xorwow(seeds, subs, 8))

    o = orc.OracleScene(os.path.join(S, "teapot_low_box.crtscene"))
    np.savez_compressed(os.path.join(G, "bvh_teapot_low.npz"), **o.bvh(0))
    digests = {}
    for name in ("teapot_low_box", "proj5_box", "proj5_glasspot", "final_frame0001", "final_frame0042"):
        o = orc.OracleScene(os.path.join(S, name + ".crtscene"))
        digests[name] = [{"nodes": int(o.bvh(m)["node_count"]), "depth": int(o.bvh(m)["depth"]), "sha256": bvh_digest(o.bvh(m))} for m in range(o.meshes)]
    json.dump(digests, open(os.path.join(G, "bvh_digests.json"), "w"), indent=1)

    for name in ("proj1_ball", "proj1_snowman", "proj5_box", "final_frame0042"):
        o = orc.OracleScene(os.path.join(S, name + ".crtscene"))
        o.set_resolution(160, 90)
        node, face, t = o.primary_hits(0)
        np.savez_compressed(os.path.join(G, f"primary_{name}_160x90.npz"), node=node.astype(np.int16), face=face, t_bits=t.view(np.uint32))

    for name, spp in (("proj13_box", 1), ("final_frame0042", 2), ("proj7_checkerboard", 2)):
        o = orc.OracleScene(os.path.join(S, name + ".crtscene"))
        o.set_resolution(64, 36)
        acc = o.render(0, spp)
        acc = o.render(1, spp, acc=acc)
        rgb = orc.convert(acc[0], o.srgb, 0.5)
        np.savez_compressed(os.path.join(G, f"accum_{name}_64x36.npz"), spp=spp, passes=2, results=acc[0], normals=acc[1], albedos=acc[2], rgb=rgb)
    print("golden vectors written to", G)
