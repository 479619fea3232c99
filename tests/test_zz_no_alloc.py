"""crt_render_pass_noalloc (the entry point that makes no device allocation: its workspace is a static array of
the code object) against crt_render_pass, through the C ABI: accumulators and ray counters bit for bit, on calls
in a row that share the workspace, a later one smaller than the one before."""
import ctypes

import numpy as np
import pytest

from conftest import scene_path
import oracle_lib as orc

import cudaraytracing_amd as crt
from cudaraytracing_amd import _capi as capi

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


def test_no_alloc_entry_point_against_the_oracle(gpu_ctx):
    """Directly against the CPU oracle (the other tests compare with crt_render_pass), two passes so that a sequence jump
    from the workspace's table is used."""
    name = "proj5_box"
    s = crt.Scene.load_file(scene_path(name))
    s.set_resolution(96, 54)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass_noalloc(0, 2)
    gpu_ctx.render_pass_noalloc(3, 1)
    got = gpu_ctx.read_accumulators()
    o = orc.OracleScene(scene_path(name))
    o.set_resolution(96, 54)
    want = o.render(3, 1, acc=o.render(0, 2))
    for a, b, what in zip(got, want, ("results", "normals", "albedos")):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), what
    with pytest.raises(crt.CrtError):
        gpu_ctx.render_pass_noalloc(64, 1)      # beyond the passes whose sequence jumps the workspace holds


def test_no_alloc_entry_point_under_graph_capture(gpu_ctx):
    """Recorded into a HIP graph on a capturing stream and replayed: stream capture refuses allocation, synchronisation and
    copies from pageable memory, so a capture that ends well is the check that the call makes none of them.  Free device
    memory is the same before and after the direct calls."""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipGetErrorString.restype = ctypes.c_char_p

    def ok(rc, what):
        assert rc == 0, f"{what}: {hip.hipGetErrorString(rc).decode()}"

    s = crt.Scene.load_file(scene_path("final_frame0042"))
    s.set_resolution(160, 90)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, 2)
    gpu_ctx.render_pass(5, 1)
    want = gpu_ctx.read_accumulators()

    free0, free1, total = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    gpu_ctx.render_pass_noalloc(0, 1)          # (anything the runtime sets up lazily on a first launch is paid here)
    ok(hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)), "hipMemGetInfo")
    gpu_ctx.render_pass_noalloc(0, 2)
    gpu_ctx.render_pass_noalloc(5, 1)
    ok(hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)), "hipMemGetInfo")
    assert free1.value == free0.value, "the call changed the amount of free device memory"

    stream, graph, graph_exec = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    ok(hip.hipStreamCreate(ctypes.byref(stream)), "hipStreamCreate")
    try:
        gpu_ctx.clear()
        ok(hip.hipStreamBeginCapture(stream, 0), "hipStreamBeginCapture")          # hipStreamCaptureModeGlobal
        rc_a = capi.lib().crt_render_pass_noalloc(gpu_ctx._h, 0, 2, stream, 1)
        rc_b = capi.lib().crt_render_pass_noalloc(gpu_ctx._h, 5, 1, stream, 1)
        ok(hip.hipStreamEndCapture(stream, ctypes.byref(graph)), "hipStreamEndCapture")
        assert rc_a == 0 and rc_b == 0, capi.lib().crt_last_error(gpu_ctx._h).decode()
        # nothing has run yet: the accumulators are still clear
        assert not any(a.any() for a in gpu_ctx.read_accumulators())
        ok(hip.hipGraphInstantiate(ctypes.byref(graph_exec), graph, None, None, ctypes.c_size_t(0)), "hipGraphInstantiate")
        ok(hip.hipGraphLaunch(graph_exec, stream), "hipGraphLaunch")
        ok(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
        got = gpu_ctx.read_accumulators()
        for a, b, what in zip(got, want, ("results", "normals", "albedos")):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), what
        # the graph can be replayed: a second launch adds the same samples again
        ok(hip.hipGraphLaunch(graph_exec, stream), "hipGraphLaunch")
        ok(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
        gpu_ctx.clear()
        gpu_ctx.render_pass(0, 2); gpu_ctx.render_pass(5, 1); gpu_ctx.render_pass(0, 2); gpu_ctx.render_pass(5, 1)
        twice = gpu_ctx.read_accumulators()
        gpu_ctx.clear()
        # (one replay at a time: two launches of one executable graph may overlap, and calls that overlap on a GPU race)
        ok(hip.hipGraphLaunch(graph_exec, stream), "hipGraphLaunch")
        ok(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
        ok(hip.hipGraphLaunch(graph_exec, stream), "hipGraphLaunch")
        ok(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
        for a, b, what in zip(gpu_ctx.read_accumulators(), twice, ("results", "normals", "albedos")):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), what
    finally:
        if graph_exec: hip.hipGraphExecDestroy(graph_exec)
        if graph: hip.hipGraphDestroy(graph)
        hip.hipStreamDestroy(stream)
