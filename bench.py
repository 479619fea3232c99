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
                "cudaraytracing_amd/csrc/device/api_device.hip",      # launch parameters and their defaults
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
        kernel_seconds = None
        for line in open(path):
            if line.startswith("# kernel_source_sha1:"):
                digest = line.split(":", 1)[1].strip()
            f = line.split()
            if "trace_kernel<false," in line and len(f) >= 3 and f[-1].startswith("["):
                sums[f[-3]] = float(f[-2]); launches[f[-3]] = float(f[-1].strip("[]"))
            elif line.startswith('"') and "trace_kernel<false," in line and kernel_seconds is None:
                import csv
                row = next(csv.reader([line]))          # --stats: name, calls, total ns, average ns, ...
                kernel_seconds = float(row[3]) * 1e-9
    except (OSError, KeyError, ValueError, IndexError, StopIteration):
        return None, "no committed counter summary for this workload"
    if digest != kernel_source_digest():
        return None, "the committed counter summary was measured on other kernel sources (re-run tools/profile_bench.sh)"
    per_launch = {k: v / launches[k] for k, v in sums.items() if launches[k]}
    if kernel_seconds:
        per_launch["kernel_seconds"] = kernel_seconds     # average duration of the kernel in the --stats pass of the same command
    return per_launch, os.path.relpath(path, ROOT)


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


def parse_args():
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
    ap.add_argument("--no-secondary", action="store_true", help="default run only: leave out the short runs of the ten-million-triangle configuration and of the many-node scene")
    ap.add_argument("--tlas", action="store_true", help="walk a BVH over the scene nodes instead of the reference's node scan (crt_set_tlas): NOT bit-exact, "
                                                        "reported with \"parity\": false and without the roofline pricing of the parity walk")
    return ap.parse_args()


def fill_defaults(args):
    """Workload defaults of the chosen configuration; returns True when the workload is the BASELINE.json configuration itself."""
    synthetic = args.config == 5
    if synthetic:
        args.scene = args.scene or "proj5_box"
        args.width, args.height, args.spp = args.width or 3840, args.height or 2160, args.spp or 256
        args.steps = args.steps or 2
        return (args.copies, args.width, args.height, args.spp) == (1582, 3840, 2160, 256)
    args.scene = args.scene or "final_frame0042"
    args.width, args.height, args.spp = args.width or 1920, args.height or 1080, args.spp or 64
    args.steps = args.steps or 5
    return (args.scene, args.width, args.height, args.spp) == ("final_frame0042", 1920, 1080, 64)


def run_workload(args, headline, rank, world, local_rank, device, coll_device, rehearsal):
    """Times args.steps frames of one workload and returns rank 0's result object (None on the other ranks)."""
    synthetic = args.config == 5
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
    ctx.set_tlas(args.tlas)
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

    # The first frame after an upload hands tiles out in storage order; from the second one on the order is costliest
    # first, by the cost the frame before measured.  The first frame is therefore always run, timed on its own and
    # reported (first_frame_ms); it counts as one of the warm-up frames.
    progress(f"upload {upload_ms:.0f} ms; first frame (tiles in storage order), then {max(args.warmup - 1, 0)} more warm-up frame(s)")
    fence()
    t0 = time.perf_counter()
    frame()
    fence()
    first_frame_ms = (time.perf_counter() - t0) * 1e3
    first_kernel = ctx.kernel_times()
    for _ in range(max(args.warmup - 1, 0)):
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
    t = torch.tensor([elapsed, first_frame_ms], dtype=torch.float64, device=coll_device)
    per_rank = torch.zeros(world * 2, dtype=torch.float64, device=coll_device)
    per_rank[2 * rank], per_rank[2 * rank + 1] = k_ms, g_ms
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    elapsed, first_frame_ms = float(t[0].item()), float(t[1].item())
    rays_per_frame = float(rays[0].item())
    ms_per_step = elapsed / args.steps * 1e3
    per_rank = per_rank.cpu().tolist()

    roofline = None
    if rank == 0 and not args.no_roofline_counters and not args.tlas:
        roofline = price_roofline(args, ctx, scene, synthetic, headline, world, rank, k_ms, len(kernel_ms), ms_per_step)

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

    out = None
    if rank == 0:
        cfg_name = "BASELINE.json configs[4]" if synthetic else "BASELINE.json configs[3]"
        out = {
            "metric": f"Mrays/s ({'synthetic 10M triangles, 4K, 256 spp' if synthetic else 'scenes/final, 1080p, 64 spp'})",
            "value": rays_per_frame / (ms_per_step * 1e-3) * 1e-6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": ms_per_step, "ms_per_frame": ms_per_step,
            "first_frame_ms": first_frame_ms, "first_frame_kernel_ms": float(first_kernel[0]) if len(first_kernel) else None,
            "first_frame_note": "the first frame after the scene upload hands tiles out in storage order (what a single crt_render of a scene pays); "
                                "the timed frames use the tile costs the frame before them measured",
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic scene generated from the reference's teapot mesh, fixed seeds" if synthetic else "reference scene file (baked), fixed seeds",
            "config": {"workload": f"{scene_label} at {W}x{H}, {args.spp} spp in 1 pass, path-traced GI" + (f" ({cfg_name})" if headline else " (not a BASELINE.json configuration)"),
                       "scene": args.scene, "width": W, "height": H, "spp": args.spp, "passes": 1, "parity": not args.tlas,
                       "parallelism": f"{STRIP_ROWS}-row image strips round-robin over {world} GPU(s), scene replicated, RGB8 rows gathered to rank 0"},
            "rays_per_frame": rays_per_frame, "closest_rays": float(rays[1].item()), "shadow_rays": float(rays[2].item()),
            "frame_includes": "accumulator clear, trace kernel, ConvertColors, row gather, D2H of the 8-bit image",
            "per_rank": [{"rank": r, "kernel_ms": per_rank[2 * r], "gather_ms": per_rank[2 * r + 1]} for r in range(world)],
            "scheduling": "tiles are handed out costliest first using the per-tile cost the previous frame measured (first_frame_ms: the frame without that); results do not depend on the order",
            "scene_upload_ms": upload_ms,
            "roofline": roofline, "cpu_baseline": cpu,
        }
    ctx.close()
    return out


def price_roofline(args, ctx, scene, synthetic, headline, world, rank, k_ms, launches_timed, ms_per_step):
    """The roofline object: algorithmic work of one launch (counting build of the kernel, untimed) over the measured
    duration of the plain kernel, and what the committed counter passes over this same command measured."""
    H, W = args.height, args.width
    count_spp = args.spp if not synthetic else min(args.spp, 8)      # the counting build is slow: count 8 spp of the big scene and scale
    scale = args.spp / float(count_spp)
    own_pixels = sum(1 for y in range(H) if (y // STRIP_ROWS) % world == rank) * W

    def counted(label):
        progress(f"{ms_per_step:.1f} ms per frame; counting traversal work, {label} (untimed)")
        ctx.clear()
        ctx.render_pass_counted(0, count_spp)
        c = {k: v * scale for k, v in ctx.stats().items()}
        ctx.kernel_times()
        return c

    # 1. the walk as the reference takes it: every scene node's box, every BVH child a box test let through or not
    c_ref = counted("the reference's walk")
    bytes_ref, flops_ref = algorithmic_work(c_ref, own_pixels)
    # 2. the walk the plain kernel takes: scene nodes by the candidate mask, no missed interior BVH children
    os.environ["CRT_COUNTED_BY_MASK"] = "1"
    try:
        ctx.upload(scene)
        ctx.set_partition(rank, world, STRIP_ROWS)
        c_own = counted("the plain kernel's own walk")
    finally:
        del os.environ["CRT_COUNTED_BY_MASK"]
    bytes_own, flops_own = algorithmic_work(c_own, own_pixels)

    prof, prof_note = profiled_counters("config5" if synthetic else "config3") if (headline and world == 1) else (None, "no committed counter summary for this workload")
    seconds = k_ms * 1e-3
    traffic = issue = lanes = measured = None
    if prof:
        if "FETCH_SIZE" in prof and "WRITE_SIZE" in prof:
            fetch, write = prof["FETCH_SIZE"] * 1024.0, prof["WRITE_SIZE"] * 1024.0          # KiB counters
            traffic = fetch + write
            kernel_s = prof.get("kernel_seconds", seconds)
            measured = {"fetch_bytes": fetch, "write_bytes": write,
                        "gbs_low": (fetch + write) / kernel_s * 1e-9, "gbs_high": (2.0 * fetch + write) / kernel_s * 1e-9,
                        "frac_low": (fetch + write) / kernel_s * 1e-9 / HBM_PEAK_GBS, "frac_high": (2.0 * fetch + write) / kernel_s * 1e-9 / HBM_PEAK_GBS,
                        "kernel_seconds": kernel_s,
                        "note": "fabric-side bytes of the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate passes) over the kernel "
                                "duration of those passes; gfx950 tallies a 128-byte read request as 64 bytes, so the read half lies between "
                                "FETCH_SIZE and twice that (gbs_low / gbs_high); Infinity Cache hits are included"}
        if "GRBM_GUI_ACTIVE" in prof and "SQ_INSTS_VALU" in prof:
            cycles = prof["GRBM_GUI_ACTIVE"] / 8.0                                # the counter sums the 8 XCDs
            issue = prof["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles)               # 4 issue cycles per wave64 instruction, 1024 SIMDs
        if "SQ_THREAD_CYCLES_VALU" in prof and prof.get("SQ_ACTIVE_INST_VALU"):
            lanes = prof["SQ_THREAD_CYCLES_VALU"] / prof["SQ_ACTIVE_INST_VALU"]
    gbs_ref, gbs_own = bytes_ref / seconds * 1e-9, bytes_own / seconds * 1e-9
    tfs_ref, tfs_own = flops_ref / seconds * 1e-12, flops_own / seconds * 1e-12
    hbm = {"bound": "hbm", "achieved": gbs_own, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_own / HBM_PEAK_GBS,
           "achieved_note": "algorithmic bytes of the walk the plain kernel takes (scene nodes by the candidate mask, no missed interior BVH children) "
                            "over the kernel's measured duration; caches serve part of them, see `measured` for what crossed the fabric",
           "algorithmic_bytes_per_launch": bytes_own,
           "reference_walk": {"achieved": gbs_ref, "frac": gbs_ref / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_ref,
                              "note": "the same rule priced on the walk the reference takes (every scene node's box, every pushed BVH child): work the plain kernel "
                                      "proves unnecessary and skips, so not a bandwidth the kernel achieved"},
           "traffic": traffic, "measured": measured,
           "traffic_note": "FETCH_SIZE + WRITE_SIZE per launch, from " + prof_note if traffic else prof_note}
    valu = {"bound": "valu", "achieved": tfs_ref, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfs_ref / FP32_VECTOR_PEAK_TFLOPS,
            "achieved_note": "algorithmic flops of the reference's walk (the pricing rule counts every scene-node box test); on the plain kernel's own walk: `own_walk`",
            "own_walk": {"achieved": tfs_own, "frac": tfs_own / FP32_VECTOR_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops_own},
            "peak_note": "157.3 TFLOP/s = 256 CUs x 4 SIMDs x 32 lanes x 2 (FMA) x 2.4 GHz; against the lane-operation rate without "
                         "the FMA factor (256 CUs x 128 lanes x 2.4 GHz = 78.6 Tlane-op/s) the fraction doubles; the kernel is built without contraction",
            "frac_of_lane_op_rate": tfs_ref / (FP32_VECTOR_PEAK_TFLOPS / 2.0),
            "algorithmic_flops_per_launch": flops_ref,
            "vector_issue_slots_used": issue, "active_lanes_per_vector_instruction": lanes,
            "counters_from": prof_note if prof else None,
            "counters_note": None if prof else prof_note}
    # The binding limit goes on top: vector instruction issue for the cache-resident scenes, memory for the scene no cache holds.
    top, nested = (hbm, valu) if synthetic else (valu, hbm)
    roofline = dict(top)
    roofline.update({"kernel": "trace_kernel", "kernel_ms_avg": k_ms, "kernel_launches_timed": int(launches_timed),
                     "traffic": traffic, "other": nested,
                     "note": "no MFMA on this path (not a dense contraction). algorithmic work = traversal counters of the counting build of the kernel "
                             "priced by the rule in bench.py::algorithmic_work, divided by the measured duration of the plain kernel"})
    return roofline


def many_nodes_workload(local_rank, n_side=40, width=1280, height=720, spp=16, frames=2):
    """A scene of n_side^2 + 2 nodes (instances of one small mesh over a floor; cudaraytracing_amd/synthetic.py): kernel
    time of the bit-exact scan (by groups of nodes) beside the opt-in walk over a BVH of the nodes, which is not
    bit-exact.  Its distance from the reference's image is measured here against the scan's image of the same frame at
    equal seeds - the image the parity tests hold equal to the CPU oracle's bit for bit - so that nothing but the product
    runs in this leg."""
    import tempfile
    from cudaraytracing_amd.synthetic import instance_grid_xml

    def psnr(a, b):
        mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
        return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)

    with tempfile.TemporaryDirectory() as tmp:
        scene = crt.Scene.load_xml(instance_grid_xml(tmp, n_side), tmp)
    nodes = int(scene.desc().node_count)
    scene.set_resolution(width, height)
    lines, images = {}, {}
    for mode in ("scan", "tlas"):
        ctx = crt.Context(local_rank)
        ctx.set_tlas(mode == "tlas")
        ctx.upload(scene)
        ctx.render_pass(0, spp)
        ctx.kernel_times()
        for _ in range(frames):
            ctx.clear()
            ctx.render_pass(0, spp)
        ms = float(np.mean(ctx.kernel_times()))
        st = ctx.stats()
        rays = st["closest_rays"] + st["shadow_rays"]
        ctx.convert(1.0, scene.srgb)
        images[mode] = ctx.read_image()
        lines[mode] = {"kernel_ms": ms, "value": rays / ms * 1e-3, "unit": "Mrays/s", "rays_per_frame": rays}
        ctx.close()
    lines["scan"]["config"] = {"parity": True, "workload": f"{nodes}-node instance grid at {width}x{height}, {spp} spp, node scan by groups of nodes"}
    lines["tlas"]["config"] = {"parity": False, "workload": f"{nodes}-node instance grid at {width}x{height}, {spp} spp, BVH over the nodes (crt_set_tlas)"}
    lines["tlas"]["psnr_vs_bit_exact_image_db"] = psnr(images["tlas"], images["scan"])
    lines["tlas"]["identical_pixels"] = float((images["tlas"] == images["scan"]).all(axis=-1).mean())
    return {"scan": lines["scan"], "tlas": lines["tlas"], "frames_timed": frames,
            "psnr_note": f"8-bit images of this frame ({width}x{height}, {spp} spp, equal seeds): the walk's against the scan's; the scan's image is the one "
                         "tests/test_gpu_parity.py::test_many_instances_of_one_mesh holds equal to the CPU oracle's bit for bit. Against the oracle itself, and "
                         "against the oracle with the one rule switched that the walk departs from on purpose: tests/test_tlas.py, profiles/r03_tlas_probe.txt",
            "note": "kernel time only (no convert, no copy)"}


def main():
    args = parse_args()
    headline = fill_defaults(args)
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

    out = run_workload(args, headline, rank, world, local_rank, device, coll_device, rehearsal)

    # The default single-GPU run also witnesses the other BASELINE configuration that fits one GPU, in short form: the
    # ten-million-triangle scene at 4K and 256 spp, one frame after the first one (scene generation and upload take
    # ~15 s, a frame ~3 s).
    if out is not None and world == 1 and args.config == 3 and headline and not args.no_secondary and not args.tlas:
        progress("secondary workload: BASELINE.json configs[4] (one timed frame)")
        sec_args = argparse.Namespace(**vars(args))
        sec_args.config, sec_args.scene, sec_args.width, sec_args.height, sec_args.spp = 5, None, None, None, None
        sec_args.steps, sec_args.warmup, sec_args.cpu_spp = 1, 1, 0
        sec_headline = fill_defaults(sec_args)
        sec = run_workload(sec_args, sec_headline, rank, world, local_rank, device, coll_device, rehearsal)
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "first_frame_ms", "first_frame_kernel_ms", "dtype", "data", "config",
                "rays_per_frame", "scene_upload_ms", "roofline")
        out["secondary"] = {k: sec[k] for k in keep}
        progress("many-node scene: node scan beside the opt-in BVH over the nodes")
        out["many_nodes"] = many_nodes_workload(local_rank)

    if out is not None:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
