"""Pins of the CPU oracle: the survey-recorded outputs of the reference (hit counts, BVH sizes),
the committed golden vectors, and rocRAND's XORWOW jump table (same recurrence as cuRAND's)."""
import hashlib
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, scene_path
import oracle_lib as orc

ANCHORS = json.load(open(os.path.join(GOLDEN, "survey_anchors.json")))


@pytest.mark.parametrize("name", sorted(ANCHORS["center_hits_1280x720"]))
def test_center_hit_counts_match_the_reference(name):
    o = orc.OracleScene(scene_path(name))
    assert (o.width, o.height) == (1280, 720)
    assert o.count_center_hits() == ANCHORS["center_hits_1280x720"][name]


@pytest.mark.parametrize("name", sorted(ANCHORS["rays_1spp_pass0_1280x720"]))
def test_ray_counts_match_the_reference(name):
    """The number of closest-hit and shadow queries of a whole pass depends on everything a path touches: geometry,
    materials, textures, lights, the random sequence and Russian roulette.  The glibc flavour of the oracle (what a
    host build of the reference computes) reproduces the counts recorded from the reference exactly."""
    want = ANCHORS["rays_1spp_pass0_1280x720"][name]
    o = orc.OracleScene(scene_path(name), "glibc")
    assert (o.width, o.height) == (1280, 720)
    _, cnt = o.render(0, 1, count=True, threads=os.cpu_count() or 1)
    assert cnt["closest_rays"] == want["closest"]
    if "shadow" in want:
        assert cnt["shadow_rays"] == want["shadow"]


def test_bvh_sizes_match_the_reference():
    want = ANCHORS["bvh_nodes"]
    got = {
        "teapot-low.obj": orc.OracleScene(scene_path("teapot_low_box")).bvh(0),
        "teapot.obj": orc.OracleScene(scene_path("proj5_box")).bvh(0),
        "teapot-hollow.obj": orc.OracleScene(scene_path("proj5_glasspot")).bvh(0),
        "final/teapot_body.obj": orc.OracleScene(scene_path("final_frame0001")).bvh(0),
    }
    for k, b in got.items():
        assert b["node_count"] == want[k], k
        assert b["depth"] <= ANCHORS["bvh_max_depth"]


def _digest(b):
    h = hashlib.sha256()
    for k in ("leaf", "start", "count", "left", "right", "next", "elements"):
        h.update(np.ascontiguousarray(b[k]).tobytes())
    h.update((np.ascontiguousarray(b["box"]) + np.float32(0)).tobytes())
    return h.hexdigest()


def test_bvh_golden_vectors():
    g = np.load(os.path.join(GOLDEN, "bvh_teapot_low.npz"))
    b = orc.OracleScene(scene_path("teapot_low_box")).bvh(0)
    for k in ("leaf", "start", "count", "left", "right", "next", "box", "elements"):
        assert np.array_equal(g[k], b[k]), k
    digests = json.load(open(os.path.join(GOLDEN, "bvh_digests.json")))
    for name, meshes in digests.items():
        o = orc.OracleScene(scene_path(name))
        assert o.meshes == len(meshes)
        for m, want in enumerate(meshes):
            got = o.bvh(m)
            assert (got["node_count"], got["depth"], _digest(got)) == (want["nodes"], want["depth"], want["sha256"]), (name, m)


def test_xorwow_golden_and_stream_structure():
    g = np.load(os.path.join(GOLDEN, "xorwow_kat.npz"))
    assert np.array_equal(orc.xorwow(g["seeds"], g["subsequences"], 8), g["words"])
    # seeding as published for curand_init: seed 0, subsequence 0
    w = orc.xorwow(np.array([0], np.uint32), np.array([0], np.uint32), 0)[0]
    s0, s1 = 0xaad26b49, 0xf7dcefdd
    t0, t1 = (1099087573 * s0) & 0xffffffff, (2591861531 * s1) & 0xffffffff
    assert list(w) == [(6615241 + t1 + t0) & 0xffffffff, (123456789 + t0) & 0xffffffff, 362436069 ^ t0,
                       (521288629 + t1) & 0xffffffff, 88675123 ^ t1, (5783321 + t0) & 0xffffffff]
    # the first output follows the xorwow recurrence
    d, v = int(w[0]), [int(x) for x in w[1:]]
    t = v[0] ^ (v[0] >> 2)
    v4 = (v[4] ^ ((v[4] << 4) & 0xffffffff)) ^ (t ^ ((t << 1) & 0xffffffff))
    first = orc.xorwow(np.array([0], np.uint32), np.array([0], np.uint32), 1)[0][6]
    assert int(first) == (v4 + d + 362437) & 0xffffffff


def test_subsequence_jump_matches_rocrand_table():
    hdr = "/opt/rocm/include/rocrand/rocrand_xorwow_precomputed.h"
    if not os.path.exists(hdr):
        pytest.skip("rocRAND headers not installed")
    text = open(hdr).read()
    body = text[text.index("h_xorwow_sequence_jump_matrices"):]
    body = body[body.index("{") + 1:]
    nums = re.findall(r"\d+", body[:body.index("}")])
    table = np.array(nums[:800], np.uint64).astype(np.uint32).reshape(160, 5)
    assert np.array_equal(orc.seqjump_matrix(0), table)          # A^(2^67)
    # second table entry is A^(4 * 2^67)
    body2 = body[body.index("}") + 1:]
    body2 = body2[body2.index("{") + 1:]
    nums2 = re.findall(r"\d+", body2[:body2.index("}")])
    assert np.array_equal(orc.seqjump_matrix(1), np.array(nums2[:800], np.uint64).astype(np.uint32).reshape(160, 5))
    # subsequence 5 = digits (1, 1) base 4: state equals one application of each table
    a = orc.xorwow(np.array([77], np.uint32), np.array([5], np.uint32), 0)[0]
    b = orc.xorwow(np.array([77], np.uint32), np.array([0], np.uint32), 0)[0]
    def apply(m, v):
        r = np.zeros(5, np.uint32)
        for i in range(5):
            for j in range(32):
                if (int(v[i]) >> j) & 1:
                    r ^= m[i * 32 + j]
        return r
    v = apply(orc.seqjump_matrix(1), apply(orc.seqjump_matrix(0), b[1:]))
    assert np.array_equal(v, a[1:]) and a[0] == b[0]


@pytest.mark.parametrize("name", ["proj1_ball", "proj1_snowman", "proj5_box", "final_frame0042"])
def test_primary_hit_golden(name):
    g = np.load(os.path.join(GOLDEN, f"primary_{name}_160x90.npz"))
    o = orc.OracleScene(scene_path(name))
    o.set_resolution(160, 90)
    node, face, t = o.primary_hits(0)
    assert np.array_equal(node, g["node"].astype(np.int32))
    assert np.array_equal(face, g["face"])
    assert np.array_equal(t.view(np.uint32), g["t_bits"])


@pytest.mark.parametrize("name", ["proj13_box", "final_frame0042", "proj7_checkerboard"])
def test_accumulator_golden(name):
    g = np.load(os.path.join(GOLDEN, f"accum_{name}_64x36.npz"))
    o = orc.OracleScene(scene_path(name))
    o.set_resolution(64, 36)
    acc = o.render(0, int(g["spp"]))
    acc = o.render(1, int(g["spp"]), acc=acc)
    for k, n in enumerate(("results", "normals", "albedos")):
        assert np.array_equal(acc[k], g[n], equal_nan=True), n
    assert np.array_equal(orc.convert(acc[0], o.srgb, 0.5), g["rgb"])


def test_rows_and_threads_do_not_change_results():
    o = orc.OracleScene(scene_path("teapot_low_box"))
    o.set_resolution(96, 54)
    whole = o.render(0, 2, threads=1)
    parts = o.new_accumulators()
    o.render(0, 2, acc=parts, rows=(0, 20), threads=3)
    o.render(0, 2, acc=parts, rows=(20, 54), threads=2)
    for a, b in zip(whole, parts):
        assert np.array_equal(a, b, equal_nan=True)
