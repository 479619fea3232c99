"""world_size-2 gloo test of the strip partition and the frame gather (CPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, scene_path

from cudaraytracing_amd.distributed import gather_rows, owned_rows


def test_owned_rows_partition_the_image():
    for H in (1, 15, 16, 17, 90, 720, 1080):
        for parts in (1, 2, 3, 4, 8):
            seen = np.concatenate([owned_rows(p, parts, H) for p in range(parts)])
            assert sorted(seen.tolist()) == list(range(H))
    assert owned_rows(1, 2, 40).tolist() == list(range(16, 32))
    assert owned_rows(0, 2, 40).tolist() == list(range(0, 16)) + list(range(32, 40))


def _worker(rank, world, port, scene, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = orc.OracleScene(scene)
        o.set_resolution(96, 54)
        acc = o.new_accumulators()
        rows = owned_rows(rank, world, 54)
        # render only the owned strips (the CPU oracle stands in for the GPU renderer here)
        for y0 in sorted(set(int(r) // 16 * 16 for r in rows)):
            o.render(0, 2, acc=acc, rows=(y0, min(y0 + 16, 54)), threads=1)
        rgb = torch.from_numpy(orc.convert(acc[0], o.srgb, 1.0))
        full_rgb = gather_rows(rgb, rank, world)
        full_acc = gather_rows(torch.from_numpy(acc[0]), rank, world)
        if rank == 0:
            np.savez(out_path, rgb=full_rgb.numpy(), acc=full_acc.numpy())
        else:
            assert full_rgb is None and full_acc is None
    finally:
        dist.destroy_process_group()


def test_two_rank_frame_equals_single_rank(tmp_path):
    import oracle_lib as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    scene = scene_path("teapot_low_box")
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, port, scene, out), nprocs=2, join=True)
    got = np.load(out)
    o = orc.OracleScene(scene)
    o.set_resolution(96, 54)
    acc = o.render(0, 2)
    assert np.array_equal(got["acc"], acc[0], equal_nan=True)
    assert np.array_equal(got["rgb"], orc.convert(acc[0], o.srgb, 1.0))


def _gpu_worker(rank, world, port, scene, size, spp, strip_rows, out_path):
    """One rank of a frame shared between GPUs, as bench.py runs it: a fresh process, its own context, its share of the
    strips, ConvertColors, then the gather of the owned rows (over gloo on host copies: both ranks share the one GPU of
    the test box, and RCCL wants one device per rank)."""
    sys.path.insert(0, ROOT)
    import cudaraytracing_amd as crt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h = size
        s = crt.Scene.load_file(scene)
        s.set_resolution(w, h)
        ctx = crt.Context(0)
        ctx.upload(s)
        ctx.set_partition(rank, world, strip_rows)
        for frame in range(2):      # twice: the second frame goes through the cached gather buffers and the measured tile order
            ctx.clear()
            ctx.render_pass(0, spp)
            ctx.convert(1.0, s.srgb)
            rgb = torch.from_numpy(ctx.read_image().reshape(h, w, 3))
            acc = torch.from_numpy(ctx.read_accumulators()[0].reshape(h, w, 3))
            full_rgb = gather_rows(rgb, rank, world, strip_rows)
            full_acc = gather_rows(acc, rank, world, strip_rows)
        st = ctx.stats()
        rays = torch.tensor([st["closest_rays"] + st["shadow_rays"]], dtype=torch.int64)
        dist.all_reduce(rays)
        if rank == 0:
            np.savez(out_path, rgb=full_rgb.numpy(), acc=full_acc.numpy(), rays=rays.numpy())
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_the_gpu_assemble_the_single_context_frame(tmp_path, world):
    """The multi-rank path end to end on the GPU: every rank a fresh process with its own context and partition, the
    HIP kernels, ConvertColors, gather_rows.  The assembled frame must equal the frame of one context alone and the
    oracle's, bit for bit, and the ranks' ray counts must add up to the whole frame's."""
    import oracle_lib as orc
    import cudaraytracing_amd as crt
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    name, size, spp, strip_rows = "final_frame0042", (160, 90), 2, 8
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_gpu_worker, args=(world, port, scene_path(name), size, spp, strip_rows, out), nprocs=world, join=True)
    got = np.load(out)
    w, h = size
    sc = crt.Scene.load_file(scene_path(name))
    sc.set_resolution(w, h)
    ctx = crt.Context(0)
    try:
        ctx.upload(sc)
        ctx.clear()
        ctx.render_pass(0, spp)
        ctx.convert(1.0, sc.srgb)
        one_rgb = ctx.read_image().reshape(h, w, 3)
        one_acc = ctx.read_accumulators()[0].reshape(h, w, 3)
        st = ctx.stats()
    finally:
        ctx.close()
    assert np.array_equal(got["rgb"], one_rgb)
    assert np.array_equal(got["acc"].view(np.uint32), one_acc.view(np.uint32))
    assert int(got["rays"][0]) == st["closest_rays"] + st["shadow_rays"]
    o = orc.OracleScene(scene_path(name))
    o.set_resolution(w, h)
    ref = o.render(0, spp)
    assert np.array_equal(got["acc"].view(np.uint32), ref[0].reshape(h, w, 3).view(np.uint32))
    assert np.array_equal(got["rgb"], orc.convert(ref[0], o.srgb, 1.0).reshape(h, w, 3))
