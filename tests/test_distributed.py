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
