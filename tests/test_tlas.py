"""The opt-in walk over a BVH of the scene's object nodes (crt_set_tlas) - the one mode of the product that does NOT
reproduce the reference bit for bit.

What it changes, and what these tests hold it to:
  * objects are met in the order of the walk, not in node order, and a ray does not go through the ToLocal / FromLocal
    round trip of every node it enters on the way: last bits of a ray and the winner among hits at one distance can
    differ.  Primary rays of scenes without depth of field must still find the same node in every pixel.
  * boxes are culled by the distance at which a ray ENTERS them.  The reference culls by the distance its box test
    reports, which for a ray that starts inside a box is the way OUT (scene/rays.cuh:156), in its node scan
    (renderer/trace.cu:59) and in its BVH walk (scene/objects.cu:190-192); what such a ray finds then depends on the
    order of the nodes, which this mode does not keep.  The oracle has a switch for exactly this rule
    (oracle_set_node_cull): with it the mode must agree with the oracle almost pixel for pixel, and against the oracle
    as the reference has it the tolerance is PSNR >= 30 dB at equal seeds on scenes where rays seldom start inside the
    box of a node that holds something nearer (no refractive mesh around the camera's paths).
"""
import numpy as np
import pytest

from conftest import scene_path
import oracle_lib as orc

import cudaraytracing_amd as crt
from cudaraytracing_amd.synthetic import instance_grid_xml

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def scene_and_oracle(name, w, h, tmp_path):
    if name.startswith("grid"):
        s = crt.Scene.load_xml(instance_grid_xml(tmp_path, int(name[4:])), str(tmp_path))
    else:
        s = crt.Scene.load_file(scene_path(name))
    s.set_resolution(w, h)
    flat = str(tmp_path / "scene.crtscene")
    s.save_file(flat)
    return s, orc.OracleScene(flat)


@pytest.fixture
def tlas_ctx():
    ctx = crt.Context(0)
    ctx.set_tlas(True)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", ["proj1_ball", "proj1_snowman", "proj4_box", "proj5_box", "proj7_checkerboard", "grid14"])
def test_primary_rays_hit_the_same_nodes(gpu_ctx, tlas_ctx, tmp_path, name):
    """Scenes without depth of field: the camera ray of a pixel is the same ray in both modes, and it must hit the
    same node (or nothing) in all but a pixel or two in a hundred thousand, at the same distance up to the rounding of the round trips it no longer
    takes; the triangle may differ only where two triangles share the hit (an edge)."""
    s, o = scene_and_oracle(name, 640, 360, tmp_path)
    assert s.desc().camera.dof == 0
    gpu_ctx.upload(s)
    tlas_ctx.upload(s)
    for p in (0, 2):
        n0, f0, t0 = gpu_ctx.primary_hits(p)
        n1, f1, t1 = tlas_ctx.primary_hits(p)
        assert (n0 != n1).sum() <= 1e-5 * n0.size, (name, int((n0 != n1).sum()))      # (measured: one pixel of one scene, on a silhouette)
        hit = (n0 >= 0) & (n0 == n1)
        rel = np.abs(t0[hit] - t1[hit]) / np.abs(t0[hit])      # (grazing hits of a sphere: the root of the quadratic moves most)
        assert (rel > 1e-3).mean() <= 1e-4 and rel.max() < 0.05, (name, float(rel.max()))
        assert (f0 != f1).sum() <= 1e-4 * f0.size


ORDER_FREE = [("proj1_ball", 4), ("proj1_snowman", 4), ("proj4_box", 4), ("proj5_box", 4), ("proj13_box", 4), ("grid14", 4)]


@pytest.mark.parametrize("name,spp", ORDER_FREE)
def test_image_against_the_oracle(tlas_ctx, tmp_path, name, spp):
    """Against the oracle as the reference has it: PSNR >= 30 dB on the 8-bit image at equal seeds and spp (every pixel
    whose path takes another turn anywhere is a different Monte-Carlo sample, so a few per cent of differing pixels
    at 4 spp already cost this much).  Measured: 99 (both sphere scenes), 38.5, 39.3, 33.3, 33.3 dB."""
    s, o = scene_and_oracle(name, 640, 360, tmp_path)
    tlas_ctx.upload(s)
    tlas_ctx.render_pass(0, spp)
    tlas_ctx.convert(1.0, s.srgb)
    want = orc.convert(o.render(0, spp)[0], o.srgb, 1.0)
    assert psnr(tlas_ctx.read_image(), want) >= 30.0


@pytest.mark.parametrize("name,least_db,least_same", [("proj4_box", 60, 0.999), ("proj5_box", 45, 0.999), ("proj7_checkerboard", 40, 0.995),
                                                     ("proj13_box", 50, 0.999), ("final_frame0042", 40, 0.995), ("grid14", 45, 0.999)])
def test_image_against_the_oracle_that_culls_by_entry(tlas_ctx, tmp_path, name, least_db, least_same):
    """With the one rule switched in the oracle that the mode departs from on purpose, the two agree but for ties and
    last bits: measured 93.7, 55.2, 46.5, 60.8, 47.0, 53.5 dB and 5 to 352 differing pixels of 230 400.  (Scenes with a
    refractive mesh - the checkerboard's teapot, the final frames' glass - lie 15 to 25 dB further from the oracle as
    the reference has it: there the reference's rule decides what a ray inside the glass finds.)"""
    s, o = scene_and_oracle(name, 640, 360, tmp_path)
    tlas_ctx.upload(s)
    tlas_ctx.render_pass(0, 4)
    tlas_ctx.convert(1.0, s.srgb)
    got = tlas_ctx.read_image()
    orc.set_node_cull(True)
    try:
        want = orc.convert(o.render(0, 4)[0], o.srgb, 1.0)
    finally:
        orc.set_node_cull(False)
    assert psnr(got, want) >= least_db
    assert (got == want).all(axis=-1).mean() >= least_same


def test_chaotic_scene_agrees_on_average(tlas_ctx, tmp_path):
    """75 glossy spheres inside a glass teapot: paths are long chains of bounces off curved surfaces, which blow a
    last-bit difference of a ray up to a different path within a few bounces, so single pixels differ (about 2 % of
    them) - but not the light the image carries: the means over 8x8 blocks agree to 30 dB and the image mean to 5 %."""
    s, o = scene_and_oracle("proj5_glasspot", 320, 180, tmp_path)
    tlas_ctx.upload(s)
    tlas_ctx.render_pass(0, 8)
    got = np.clip(np.nan_to_num(tlas_ctx.read_accumulators()[0], nan=0.0, posinf=1.0), 0, 1) * 255
    orc.set_node_cull(True)
    try:
        want = np.clip(np.nan_to_num(o.render(0, 8)[0], nan=0.0, posinf=1.0), 0, 1) * 255
    finally:
        orc.set_node_cull(False)
    blocks = lambda a: a[:176].reshape(22, 8, 40, 8, 3).mean(axis=(1, 3))
    assert psnr(blocks(got), blocks(want)) >= 30.0
    assert abs(got.mean() - want.mean()) <= 0.05 * want.mean()


def test_many_nodes_are_what_it_is_for(gpu_ctx, tlas_ctx, tmp_path):
    """1 602 scene nodes (40 x 40 instances of one mesh, floor): the walk must beat the scan by groups of nodes by a
    factor of two at least (measured: 12 ms against 100 ms at 1280x720, 4 spp) and stay under 90 ms."""
    s, _ = scene_and_oracle("grid40", 1280, 720, tmp_path)
    ms = []
    for ctx in (gpu_ctx, tlas_ctx):
        ctx.upload(s)
        ctx.render_pass(0, 4)
        ctx.clear()
        ctx.render_pass(0, 4)
        ms.append(ctx.last_kernel_ms())
    assert ms[1] < 90.0 and ms[1] < 0.5 * ms[0], ms


def test_counting_build_walks_the_same_way(tlas_ctx, tmp_path):
    """The counting build of the walk computes the same image as the plain one, bit for bit, and its counters show a
    walk: far fewer node boxes entered than the scan tests, box steps on top of the meshes' own."""
    s, o = scene_and_oracle("grid14", 320, 180, tmp_path)
    tlas_ctx.upload(s)
    tlas_ctx.render_pass(0, 2)
    plain = tlas_ctx.read_accumulators()
    tlas_ctx.clear()
    tlas_ctx.render_pass_counted(0, 2)
    counted = tlas_ctx.read_accumulators()
    for a, b in zip(plain, counted):
        assert np.array_equal(a, b, equal_nan=True)
    st = tlas_ctx.stats()
    _, cnt = o.render(0, 2, count=True)
    rays = st["closest_rays"] + st["shadow_rays"]
    assert 0 < st["node_enters"] < 0.25 * cnt["node_box_tests"]
    assert st["bvh_interior"] > rays      # at least the root of the BVH over the nodes, per ray


def test_walk_is_deterministic_and_partitions_like_the_scan(tlas_ctx, tmp_path):
    """Not bit-exact against the reference, but exact against itself: a query's walk depends on nothing but the query,
    so two renders agree bit for bit, and the strips of two ranks put together are the image of one."""
    s, _ = scene_and_oracle("final_frame0042", 320, 180, tmp_path)
    tlas_ctx.upload(s)
    tlas_ctx.render_pass(0, 3)
    whole = [a.copy() for a in tlas_ctx.read_accumulators()]
    tlas_ctx.clear()
    tlas_ctx.render_pass(0, 3)
    for a, b in zip(whole, tlas_ctx.read_accumulators()):
        assert np.array_equal(a, b, equal_nan=True)
    tlas_ctx.clear()
    for rank in (0, 1):
        tlas_ctx.set_partition(rank, 2, 8)
        tlas_ctx.render_pass(0, 3)
    tlas_ctx.set_partition(0, 1, 16)
    for a, b in zip(whole, tlas_ctx.read_accumulators()):
        assert np.array_equal(a, b, equal_nan=True)


def test_deep_mesh_under_the_walk(tlas_ctx, tmp_path):
    """A mesh whose BVH is far deeper than the part of a query's stack that lives in LDS (126 k triangles, depth 22;
    7 entries in LDS): the walk over the nodes, the marker under the mesh walk and the mesh's own entries share one
    stack, most of which is in device memory here.  Against the oracle that culls by entry."""
    src = crt.Scene.load_file(scene_path("proj5_box"))
    syn = src.make_instanced(0, 20, seed=1, width=192, height=108)
    flat = str(tmp_path / "instanced20.crtscene")
    syn.save_file(flat)
    o = orc.OracleScene(flat)
    assert o.bvh(0)["depth"] > 17
    tlas_ctx.upload(syn)
    tlas_ctx.render_pass(0, 3)
    tlas_ctx.convert(1.0, syn.srgb)
    got = tlas_ctx.read_image()
    orc.set_node_cull(True)
    try:
        want = orc.convert(o.render(0, 3)[0], o.srgb, 1.0)
    finally:
        orc.set_node_cull(False)
    assert psnr(got, want) >= 40.0
    assert (got == want).all(axis=-1).mean() >= 0.995
    tlas_ctx.stats()      # raises if the kernel reported a protocol error


def test_scene_without_objects(gpu_ctx, tlas_ctx, tmp_path):
    """Nothing to build a tree over: every query ends at once, and both modes render the same (environment only)."""
    (tmp_path / "empty.xml").write_text(
        '<xml><scene><background r="0.2" g="0.4" b="0.6"/><environment r="0.2" g="0.4" b="0.6"/>'
        '<light type="point" name="l"><intensity value="10"/><position x="0" y="0" z="5"/><size value="1"/></light></scene>'
        '<camera><position x="0" y="-5" z="1"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/><fov value="40"/>'
        '<width value="96"/><height value="64"/></camera></xml>')
    s = crt.Scene.load_xml(str(tmp_path / "empty.xml"), str(tmp_path))
    images = []
    for ctx in (gpu_ctx, tlas_ctx):
        ctx.upload(s)
        ctx.clear()
        ctx.render_pass(0, 2)
        ctx.stats()
        images.append([a.copy() for a in ctx.read_accumulators()])
    for a, b in zip(*images):
        assert np.array_equal(a, b, equal_nan=True)


def test_switching_back_restores_parity(tmp_path):
    """The switch takes effect at upload and only there: the same context renders bit-exact images again once it is
    off."""
    s, o = scene_and_oracle("proj5_box", 192, 108, tmp_path)
    ctx = crt.Context(0)
    try:
        ctx.set_tlas(True)
        ctx.upload(s)
        ctx.render_pass(0, 2)
        ctx.set_tlas(False)
        ctx.upload(s)
        ctx.clear()
        ctx.render_pass(0, 2)
        want = o.render(0, 2)
        assert np.array_equal(ctx.read_accumulators()[0], want[0], equal_nan=True)
    finally:
        ctx.close()
