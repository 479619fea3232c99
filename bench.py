#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the per-pixel trace path.

Workload (BASELINE.json configs[3], the configuration the metric is quoted on):
scenes/final frame0042 at 1920x1080, 64 samples per pixel in one pass, image strips
split over the N GPUs of one node, framebuffer gathered to rank 0 over RCCL.
`--config 5` selects BASELINE.json configs[4] instead: the synthetic scene of 1582 teapot
copies flattened into one mesh (9 998 240 triangles, one BVH), 3840x2160, 256 spp.

One "step" is one whole frame: clear accumulators, trace kernel, ConvertColors,
gather of the owned rows to rank 0 and the device-to-host copy of the 8-bit image.
The scene is resident in HBM before the timed region.  A ray is one closest-hit query
or one shadow query toward a point light.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import cudaraytracing_amd as crt
from cudaraytracing_amd.distributed import gather_rows

FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X fp32 vector peak (FMA counted as 2)
HBM_PEAK_GBS = 8000.0


# Image strips of this many rows go round-robin to the ranks: on the benchmark frame 8-row strips (one tile row)
# leave the busiest of 8 ranks 0.8 % above the mean, 16-row strips 4 %.
STRIP_ROWS = 8


def algorithmic_work(c, pixels):
    """Bytes and flops one launch needs by the counting rule of the design notes: 24 B per scene-node
    box, 112 B per entered node (two transforms + tags), 56 B per expanded BVH interior node (two child
    boxes + two references), 8 B per leaf, 36 B per triangle tested, 72 B per accepted triangle
    (normals + texture vertices), 12 B per bilinear texture evaluation, 36 B of accumulators per pixel;
    25 flops per slab test, 66 per node enter/leave, 55 per triangle, 40 per sphere, 15 per plane."""
    nbytes = (24 * c["node_box_tests"] + 112 * c["node_enters"] + 56 * c["bvh_interior"] + 8 * c["bvh_leaves"] +
              36 * c["tri_tests"] + 72 * c["tri_hits"] + 12 * c["texel_quads"] + 36 * pixels)
    flops = (25 * (c["node_box_tests"] + 2 * c["bvh_interior"]) + 66 * c["node_enters"] + 55 * c["tri_tests"] +
             40 * c["sphere_tests"] + 15 * c["plane_tests"])
    return nbytes, flops


def kernel_source_digest():
    """SHA-1 over the sources the trace kernel is built from, comments and white space left out (a reworded comment does
    not make a measurement stale, a changed statement does); a committed counter summary carries the digest of the
    sources it was measured on, so a stale summary is recognised."""
    import hashlib, re
    h = hashlib.sha1()
    for rel in ("cudaraytracing_amd/csrc/device/kernels.hip", "cudaraytracing_amd/csrc/device/dev_scene.h",
                "cudaraytracing_amd/csrc/crt_detmath.h", "cudaraytracing_amd/csrc/crt_vec.h"):
        with open(os.path.join(ROOT, rel), "r", encoding="utf-8", errors="replace") as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)       # block comments
        text = re.sub(r"//[^\n]*", " ", text)                    # line comments (no string literal of these files holds //)
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()


def profiled_counters(config):
    """Counters of the committed rocprofv3 --pmc passes over this same command (tools/profile_bench.sh; PMC counters
    cannot be read from inside the process).  Returns (per-launch dict, note); the dict is None when there is no
    summary for the workload or when it was measured on other kernel sources than the ones in the tree."""
    try:
        index = json.load(open(os.path.join(ROOT, "profiles", "latest.json")))
        path = os.path.join(ROOT, "profiles", index[config])
        sums, launches, digest = {}, {}, None
        for line in open(path):
            if line.startswith("# kernel_source_sha1:"):
                digest = line.split(":", 1)[1].strip()
            f = line.split()
            if "trace_kernel<false," in line and len(f) >= 3 and f[-1].startswith("["):
                sums[f[-3]] = float(f[-2]); launches[f[-3]] = float(f[-1].strip("[]"))
    except (OSError, KeyError, ValueError):
        return None, "no committed counter summary for this workload"
    if digest != kernel_source_digest():
        return None, "the committed counter summary was measured on other kernel sources (re-run tools/profile_bench.sh)"
    return {k: v / launches[k] for k, v in sums.items() if launches[k]}, os.path.relpath(path, ROOT)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene_file, scene_label, width, height, spp, full_spp, threads):
    """Times the CPU oracle (glibc flavour, multithreaded) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    o = orc.OracleScene(scene_file, "glibc")
    o.set_resolution(width, height)
    t0 = time.perf_counter()
    _, cnt = o.render(0, spp, threads=threads, count=True)
    dt = time.perf_counter() - t0
    rays = cnt["closest_rays"] + cnt["shadow_rays"]
    return {"value": rays / dt * 1e-6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model(), "host_cpus_visible": os.cpu_count(),
            "cores_note": "threads = OMP_NUM_THREADS, the CPU share of a one-GPU box (the machine shows more CPUs than the job may use)"
                          if os.environ.get("OMP_NUM_THREADS") else "threads = all CPUs the host shows",
            "sample": f"{scene_label} {width}x{height}, {spp} spp of the {full_spp}, all rows, {dt:.1f} s of wall time",
            "ms_per_frame_extrapolated": dt * 1e3 * full_spp / spp}


def progress(msg):
    """One line on stderr per phase: the JSON result is the only thing on stdout."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=(3, 5), help="3: BASELINE configs[3] (the headline), 5: the synthetic ten-million-triangle scene")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--copies", type=int, default=1582, help="config 5: teapot copies")
    ap.add_argument("--cpu-spp", type=int, default=24, help="samples per pixel of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline-counters", action="store_true")
    args = ap.parse_args()
    synthetic = args.config == 5
    if synthetic:
        args.scene = args.scene or "proj5_box"
        args.width, args.height, args.spp = args.width or 3840, args.height or 2160, args.spp or 256
        args.steps = args.steps or 2
    else:
        args.scene = args.scene or "final_frame0042"
        args.width, args.height, args.spp = args.width or 1920, args.height or 1080, args.spp or 64
        args.steps = args.steps or 5
    headline = (not synthetic and (args.scene, args.width, args.height, args.spp) == ("final_frame0042", 1920, 1080, 64)) or \
               (synthetic and (args.copies, args.width, args.height, args.spp) == (1582, 3840, 2160, 256))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs a launcher: python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    # Rehearsal of the multi-rank path on a box with one GPU (CRT_BENCH_REHEARSAL=1): every rank renders on device 0
    # and the collectives run over gloo on host copies.  Its timings mean nothing; it exists to exercise the code.
    rehearsal = os.environ.get("CRT_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    coll_device = torch.device("cpu") if rehearsal else device

    scene_file = os.path.join(ROOT, "tests", "golden", "scenes", args.scene + ".crtscene")
    scene = crt.Scene.load_file(scene_file)
    if synthetic:
        scene = scene.make_instanced(0, args.copies, 1, args.width, args.height)
        scene_label = f"synthetic scene: {args.copies} copies of {args.scene} mesh 0 in one BVH ({scene.desc().meshes[0].nf} triangles), floor, point light"
    else:
        scene.set_resolution(args.width, args.height)
        scene_label = {"final_frame0042": "scenes/final/frame0042.xml"}.get(args.scene, args.scene + ".crtscene")
    progress(f"scene ready: {scene_label}; uploading (BVH build, re-layout, copy to the device)")
    t0 = time.perf_counter()
    ctx = crt.Context(local_rank)
    ctx.upload(scene)                      # BVH build + re-layout + H2D, outside the timed region
    upload_ms = (time.perf_counter() - t0) * 1e3
    ctx.set_partition(rank, world, STRIP_ROWS)
    H, W = args.height, args.width
    rgb_dev = torch.zeros((H, W, 3), dtype=torch.uint8, device=device)
    rgb_host = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
    srgb = scene.srgb
    gather_ms = []

    def frame():
        stream = torch.cuda.current_stream().cuda_stream
        ctx.clear(stream=stream, wait=False)
        ctx.render_pass(0, args.spp, stream=stream, wait=False)
        ctx.convert(1.0, srgb, out_device_ptr=rgb_dev.data_ptr(), stream=stream, wait=False)
        if world > 1:
            # the exchange step, timed on its own (the wait for this rank's kernel is part of it) so that a run on
            # several GPUs shows where its time goes
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            full = gather_rows(rgb_dev.cpu() if rehearsal else rgb_dev, rank, world, STRIP_ROWS)
            e1.record()
            gather_ms.append((e0, e1))
        else:
            full = rgb_dev
        if rank == 0:
            rgb_host.copy_(full, non_blocking=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    progress(f"upload {upload_ms:.0f} ms; {args.warmup} warm-up frame(s)")
    for _ in range(args.warmup):
        frame()
    fence()
    ctx.kernel_times()                      # drop warm-up launches
    gather_ms.clear()
    progress(f"timing {args.steps} frame(s)")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ctx.kernel_times()
    k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
    g_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_ms])) if gather_ms else 0.0

    st = ctx.stats()                        # counters of the last frame (cleared at each frame start); raises on a kernel protocol error
    rays = torch.tensor([st["closest_rays"] + st["shadow_rays"], st["closest_rays"], st["shadow_rays"]], dtype=torch.float64, device=coll_device)
    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
    per_rank = torch.zeros(world * 2, dtype=torch.float64, device=coll_device)
    per_rank[2 * rank], per_rank[2 * rank + 1] = k_ms, g_ms
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays_per_frame = float(rays[0].item())
    ms_per_step = elapsed / args.steps * 1e3
    per_rank = per_rank.cpu().tolist()

    roofline = None
    if rank == 0 and not args.no_roofline_counters:
        # traversal counters of this rank's share of the frame, from the counting build of the kernel (untimed)
        progress(f"{ms_per_step:.1f} ms per frame; counting traversal work (untimed)")
        ctx.clear()
        ctx.render_pass_counted(0, args.spp if not synthetic else min(args.spp, 8))
        c = ctx.stats()
        if synthetic and args.spp > 8:      # the counted build is slow: count 8 spp and scale (every sample does the same kind of work)
            c = {k: v * args.spp / 8.0 for k, v in c.items()}
        ctx.kernel_times()
        own_pixels = sum(1 for y in range(H) if (y // STRIP_ROWS) % world == rank) * W
        nbytes, flops = algorithmic_work(c, own_pixels)
        prof, prof_note = profiled_counters("config5" if synthetic else "config3") if (headline and world == 1) else (None, "no committed counter summary for this workload")
        gbs = nbytes / (k_ms * 1e-3) * 1e-9
        tfs = flops / (k_ms * 1e-3) * 1e-12
        traffic = issue = lanes = clock_ghz = None
        if prof:
            if "FETCH_SIZE" in prof and "WRITE_SIZE" in prof:
                traffic = (prof["FETCH_SIZE"] + prof["WRITE_SIZE"]) * 1024.0          # KiB counters
            if "GRBM_GUI_ACTIVE" in prof and "SQ_INSTS_VALU" in prof:
                cycles = prof["GRBM_GUI_ACTIVE"] / 8.0                                # the counter sums the 8 XCDs
                issue = prof["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles)               # 4 issue cycles per wave64 instruction, 1024 SIMDs
            if "SQ_THREAD_CYCLES_VALU" in prof and prof.get("SQ_ACTIVE_INST_VALU"):
                lanes = prof["SQ_THREAD_CYCLES_VALU"] / prof["SQ_ACTIVE_INST_VALU"]
        hbm = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
               "traffic": traffic,
               "traffic_note": "FETCH_SIZE + WRITE_SIZE per launch (separate rocprofv3 --pmc passes; gfx950 tallies 128-byte read requests as 64 bytes, "
                               "so the read half is a lower bound; Infinity Cache hits are included), from " + prof_note if traffic else prof_note,
               "algorithmic_bytes_per_launch": nbytes}
        valu = {"bound": "valu", "achieved": tfs, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfs / FP32_VECTOR_PEAK_TFLOPS,
                "peak_note": "157.3 TFLOP/s = 256 CUs x 4 SIMDs x 32 lanes x 2 (FMA) x 2.4 GHz; against the lane-operation rate without "
                             "the FMA factor (256 CUs x 128 lanes x 2.4 GHz = 78.6 Tlane-op/s) the fraction doubles; the kernel is built without contraction",
                "frac_of_lane_op_rate": tfs / (FP32_VECTOR_PEAK_TFLOPS / 2.0),
                "algorithmic_flops_per_launch": flops,
                "vector_issue_slots_used": issue, "active_lanes_per_vector_instruction": lanes,
                "counters_from": prof_note if prof else None,
                "counters_note": None if prof else prof_note}
        # The binding limit goes on top: vector instruction issue for the cache-resident scenes, HBM for the scene no cache holds.
        top, nested = (hbm, valu) if synthetic else (valu, hbm)
        roofline = dict(top)
        roofline.update({"kernel": "trace_kernel", "kernel_ms_avg": k_ms, "kernel_launches_timed": int(len(kernel_ms)),
                         "traffic": traffic, "other": nested,
                         "note": "no MFMA on this path (not a dense contraction). algorithmic work = traversal counters of the counting build "
                                 "of the kernel (full scene-node scan and every BVH expansion, as the reference walks them) priced by the rule in "
                                 "bench.py::algorithmic_work, divided by the measured duration of the plain kernel; the plain kernel passes over "
                                 "scene nodes and BVH children that are certain to be missed, so it fetches less than that (see traffic)"})

    if rehearsal and rank == 0 and world > 1:
        # the gathered frame must equal a frame rendered by one context alone
        torch.cuda.synchronize()
        gathered = rgb_host.numpy().copy()
        ctx.set_partition(0, 1, STRIP_ROWS)
        ctx.clear(); ctx.render_pass(0, args.spp); ctx.convert(1.0, srgb)
        same = bool((ctx.read_image() == gathered).all())
        ctx.set_partition(rank, world, STRIP_ROWS)
        print(f"rehearsal: gathered frame of {world} ranks identical to a single-context frame: {same}", file=sys.stderr)
        if not same:
            sys.exit("rehearsal: multi-rank frame differs from the single-context frame")

    cpu = None
    if rank == 0 and world == 1 and args.cpu_spp > 0:
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        progress(f"CPU baseline on {threads} threads")
        if synthetic:
            # The checker builds its BVH with the reference's own algorithm (every split rescans the node's elements 93
            # times), which takes hours on ten million triangles: the CPU sample is the same generator with 64 of the
            # copies, at a quarter of the width and of the height, ten times --cpu-spp samples.
            import tempfile
            few = min(args.copies, 64)
            with tempfile.TemporaryDirectory() as tmp:
                small = crt.Scene.load_file(scene_file).make_instanced(0, few, 1, W // 4, H // 4)
                path = os.path.join(tmp, "synthetic.crtscene")
                small.save_file(path)
                cpu = cpu_baseline(path, f"synthetic scene with {few} of the {args.copies} copies", W // 4, H // 4, min(args.spp, 10 * args.cpu_spp), args.spp, threads)
                cpu["ms_per_frame_extrapolated"] = None     # a smaller scene: its rate is the baseline, a frame time would not be
        else:
            cpu = cpu_baseline(scene_file, scene_label, W, H, args.cpu_spp, args.spp, threads)

    if rank == 0:
        cfg_name = "BASELINE.json configs[4]" if synthetic else "BASELINE.json configs[3]"
        out = {
            "metric": f"Mrays/s ({'synthetic 10M triangles, 4K, 256 spp' if synthetic else 'scenes/final, 1080p, 64 spp'})",
            "value": rays_per_frame / (ms_per_step * 1e-3) * 1e-6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_frame": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic scene generated from the reference's teapot mesh, fixed seeds" if synthetic else "reference scene file (baked), fixed seeds",
            "config": {"workload": f"{scene_label} at {W}x{H}, {args.spp} spp in 1 pass, path-traced GI" + (f" ({cfg_name})" if headline else " (not a BASELINE.json configuration)"),
                       "scene": args.scene, "width": W, "height": H, "spp": args.spp, "passes": 1,
                       "parallelism": f"{STRIP_ROWS}-row image strips round-robin over {world} GPU(s), scene replicated, RGB8 rows gathered to rank 0"},
            "rays_per_frame": rays_per_frame, "closest_rays": float(rays[1].item()), "shadow_rays": float(rays[2].item()),
            "frame_includes": "accumulator clear, trace kernel, ConvertColors, row gather, D2H of the 8-bit image",
            "per_rank": [{"rank": r, "kernel_ms": per_rank[2 * r], "gather_ms": per_rank[2 * r + 1]} for r in range(world)],
            "scheduling": "tiles are handed out costliest first using the per-tile cost the previous frame measured (warm-up frame for the first timed one); results do not depend on the order",
            "scene_upload_ms": upload_ms,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
