"""crt_render_pass_noalloc (the entry point that makes no device allocation: its workspace is a static array of
the code object) against crt_render_pass, through the C ABI: accumulators and ray counters bit for bit, on calls
in a row that share the workspace, a later one smaller than the one before."""
import numpy as np
import pytest

from conftest import scene_path

import cudaraytracing_amd as crt

pytestmark = pytest.mark.gpu


def render_both(ctx, scene, passes):
    """Accumulators and ray counters after `passes` = [(pass_id, samples), ...] with each entry point."""
    out = []
    for noalloc in (False, True):
        ctx.upload(scene)
        ctx.clear()
        for pass_id, samples in passes:
            if noalloc:
                ctx.render_pass_noalloc(pass_id, samples)
            else:
                ctx.render_pass(pass_id, samples)
        st = ctx.stats()
        out.append((ctx.read_accumulators(), (st["closest_rays"], st["shadow_rays"])))
    return out


def check(ctx, scene, passes):
    (acc_a, rays_a), (acc_b, rays_b) = render_both(ctx, scene, passes)
    assert rays_a == rays_b and rays_a[0] > 0
    for a, b, name in zip(acc_a, acc_b, ("results", "normals", "albedos")):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name


def test_no_alloc_entry_point_equals_render_pass(gpu_ctx):
    big = crt.Scene.load_file(scene_path("final_frame0042"))
    big.set_resolution(320, 180)
    check(gpu_ctx, big, [(0, 3)])
    # second call in a row, smaller input, another pass number (its sequence jump goes through the workspace too)
    small = crt.Scene.load_file(scene_path("proj5_box"))
    small.set_resolution(96, 54)
    check(gpu_ctx, small, [(2, 2), (0, 1)])
    # a partition of the image: fewer pixels than lanes
    gpu_ctx.upload(small)
    gpu_ctx.set_partition(1, 4, 8)
    gpu_ctx.clear()
    gpu_ctx.render_pass(1, 2)
    want = gpu_ctx.read_accumulators()
    gpu_ctx.clear()
    gpu_ctx.render_pass_noalloc(1, 2)
    got = gpu_ctx.read_accumulators()
    for a, b in zip(want, got):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_no_alloc_entry_point_on_a_deep_tree(gpu_ctx):
    """A BVH deeper than the traversal stack in LDS: the overflow entries live in the workspace as well."""
    src = crt.Scene.load_file(scene_path("proj5_box"))
    syn = src.make_instanced(0, 20, seed=1, width=160, height=90)
    check(gpu_ctx, syn, [(0, 2)])


def test_no_alloc_entry_point_argument_errors(gpu_ctx):
    s = crt.Scene.load_file(scene_path("proj1_ball"))
    s.set_resolution(64, 36)
    gpu_ctx.upload(s)
    with pytest.raises(crt.CrtError):
        gpu_ctx.render_pass_noalloc(-1, 1)
    with pytest.raises(crt.CrtError):
        gpu_ctx.render_pass_noalloc(0, 0)
