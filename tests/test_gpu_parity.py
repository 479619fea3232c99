"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Integer / index outputs (hit ids, ray counts, 8-bit image) and - because both sides use the
same pinned transcendentals and no fused multiply-add - the float accumulators are compared
bit for bit.  Against the glibc flavour of the oracle (what a host build of the reference
computes) the tolerance is PSNR >= 35 dB on the 8-bit image at equal seeds and spp."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path
import oracle_lib as orc

import cudaraytracing_amd as crt
from cudaraytracing_amd.distributed import owned_rows

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def load_both(name, w, h, flavour="det"):
    s = crt.Scene.load_file(scene_path(name))
    s.set_resolution(w, h)
    o = orc.OracleScene(scene_path(name), flavour)
    o.set_resolution(w, h)
    return s, o


def test_pinned_math_is_bit_identical_on_device(gpu_ctx):
    rs = np.random.RandomState(11)
    n = 1 << 18
    x = rs.rand(n).astype(np.float32)
    special = np.array([0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 1e-45, 3.4e38, -3.4e38, 0.5, 2, 88.7, -104, 1e9], np.float32)
    cases = [(0, x * 7 - 0.5, None), (1, x * 7 - 0.5, None), (2, (x - 0.5) * 220, None),
             (3, x * 1.5 - 0.25, rs.choice(np.array([1024, 20, 0.41666, 5, 2, 1 / 21, 100, 33, 0, -3, 21], np.float32), n)),
             (4, (x - 0.5) * 4, (rs.rand(n).astype(np.float32) - 0.5) * 4), (5, x * 2.2 - 1.1, None), (6, x * 2.2 - 1.1, None), (7, (x - 0.5) * 3, None)]
    for fn, a, b in cases:
        a = np.concatenate([a, special]).astype(np.float32)
        b = None if b is None else np.concatenate([b, special[::-1]]).astype(np.float32)
        g = gpu_ctx.math_selftest(fn, a, b)
        c = orc.math(fn, a, b)
        nan_both = np.isnan(g) & np.isnan(c)
        assert np.array_equal(g.view(np.uint32)[~nan_both], c.view(np.uint32)[~nan_both]), fn


def test_pinned_math_on_device_is_correctly_rounded(gpu_ctx):
    """The device functions against an evaluation that shares nothing with them or with the oracle's copy of the
    algorithm: numpy's 80-bit long double, rounded to float, a million operands per function."""
    from test_detmath import reference_cases
    for fn, a, b, want in reference_cases(1_000_000, seed=77):
        got = gpu_ctx.math_selftest(fn, a, b)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (fn, int((~same).sum()))


def test_seeded_sequence_matches(gpu_ctx):
    k = np.load(os.path.join(GOLDEN, "xorwow_kat.npz"))
    assert np.array_equal(gpu_ctx.rng_selftest(k["seeds"], k["subsequences"], 8), k["words"])
    rs = np.random.RandomState(5)
    seeds = rs.randint(0, 1 << 31, 512).astype(np.uint32)
    subs = rs.randint(0, 300, 512).astype(np.uint32)
    assert np.array_equal(gpu_ctx.rng_selftest(seeds, subs, 16), orc.xorwow(seeds, subs, 16))


@pytest.mark.parametrize("name", ["proj1_ball", "proj1_snowman", "proj5_box", "final_frame0042"])
def test_primary_hits_equal_golden(gpu_ctx, name):
    g = np.load(os.path.join(GOLDEN, f"primary_{name}_160x90.npz"))
    s = crt.Scene.load_file(scene_path(name))
    s.set_resolution(160, 90)
    gpu_ctx.upload(s)
    node, face, t = gpu_ctx.primary_hits(0)
    assert np.array_equal(node, g["node"].astype(np.int32))
    assert np.array_equal(face, g["face"])
    assert np.array_equal(t.view(np.uint32), g["t_bits"])


@pytest.mark.parametrize("name,w,h", [("proj1_ball", 1280, 720), ("proj5_glasspot", 640, 360), ("final_frame0001", 640, 360),
                                      ("proj7_checkerboard", 640, 360), ("final_frame0042", 640, 360)])
def test_primary_hits_equal_oracle(gpu_ctx, name, w, h):
    """BASELINE.json configs[0] (proj1 sphere scene, primary rays, scene resolution) and the mesh scenes."""
    s, o = load_both(name, w, h)
    gpu_ctx.upload(s)
    for p in (0, 3):
        node, face, t = gpu_ctx.primary_hits(p)
        onode, oface, ot = o.primary_hits(p)
        assert np.array_equal(node, onode) and np.array_equal(face, oface)
        assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))


@pytest.mark.parametrize("name,spp", [("proj13_box", 1), ("final_frame0042", 2), ("proj7_checkerboard", 2)])
def test_accumulators_equal_golden(gpu_ctx, name, spp):
    g = np.load(os.path.join(GOLDEN, f"accum_{name}_64x36.npz"))
    s = crt.Scene.load_file(scene_path(name))
    s.set_resolution(64, 36)
    gpu_ctx.upload(s)
    gpu_ctx.render_pass(0, spp)
    gpu_ctx.render_pass(1, spp)
    acc = gpu_ctx.read_accumulators()
    for k, n in enumerate(("results", "normals", "albedos")):
        assert same(acc[k], g[n]), n
    gpu_ctx.convert(0.5, s.srgb)
    assert np.array_equal(gpu_ctx.read_image(), g["rgb"])


CASES = [
    ("proj4_box", 320, 180, 4),            # spheres, ambient + point light, refraction
    ("teapot_low_box", 1280, 720, 1),      # BASELINE.json configs[1]: teapot-low BVH, shadows, 1 spp, scene resolution
    ("proj5_box", 320, 180, 4),            # two teapot.obj instances, NEE through a mesh
    ("proj5_glasspot", 256, 144, 2),       # 86 nodes, hollow teapot, nested transforms
    ("proj7_checkerboard", 320, 180, 16),  # BASELINE.json configs[2] at reduced size: textures, glossy + refractive, 16 spp
    ("proj13_box", 320, 180, 4),
    ("final_frame0001", 256, 144, 2),
    ("final_frame0042", 320, 180, 4),      # BASELINE.json configs[3] scene: 20 meshes, dof, sRGB
    ("proj5_plane", 320, 180, 4),
    ("proj1_snowman", 320, 180, 2),        # no lights at all: no light sample is taken
    ("proj7_checkerboard", 1280, 720, 16), # BASELINE.json configs[2] at its full size
]


@pytest.mark.parametrize("name,w,h,spp", CASES)
def test_render_is_bit_exact(gpu_ctx, name, w, h, spp):
    s, o = load_both(name, w, h)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, spp)
    acc = gpu_ctx.read_accumulators()
    ref, cnt = o.render(0, spp, count=True)
    for k, n in enumerate(("results", "normals", "albedos")):
        assert same(acc[k], ref[k]), (name, n)
    st = gpu_ctx.stats()
    assert (st["closest_rays"], st["shadow_rays"]) == (cnt["closest_rays"], cnt["shadow_rays"])
    gpu_ctx.convert(1.0, s.srgb)
    assert np.array_equal(gpu_ctx.read_image(), orc.convert(ref[0], o.srgb, 1.0))


def test_instanced_scene_is_bit_exact(gpu_ctx, tmp_path):
    """BASELINE.json configs[4] at reduced size: copies of teapot.obj flattened into one mesh (126 k triangles,
    BVH depth 22, so the traversal stack in LDS is deeper than on any shipped scene), floor, point light."""
    src = crt.Scene.load_file(scene_path("proj5_box"))
    syn = src.make_instanced(0, 20, seed=1, width=192, height=108)
    path = str(tmp_path / "instanced20.crtscene")
    syn.save_file(path)
    o = orc.OracleScene(path)
    assert o.bvh(0)["depth"] > 17
    gpu_ctx.upload(syn)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, 3)
    acc = gpu_ctx.read_accumulators()
    ref, cnt = o.render(0, 3, count=True)
    for k, n in enumerate(("results", "normals", "albedos")):
        assert same(acc[k], ref[k]), n
    st = gpu_ctx.stats()
    assert (st["closest_rays"], st["shadow_rays"]) == (cnt["closest_rays"], cnt["shadow_rays"])
    ids = gpu_ctx.primary_hits(0)
    oid = o.primary_hits(0)
    for a, b in zip(ids, oid):
        assert same(a, b)


def test_counted_kernel_matches_oracle_counters(gpu_ctx):
    s, o = load_both("final_frame0042", 192, 108)
    gpu_ctx.upload(s)
    gpu_ctx.render_pass_counted(0, 2)
    acc = gpu_ctx.read_accumulators()
    ref, cnt = o.render(0, 2, count=True)
    assert same(acc[0], ref[0])
    st = gpu_ctx.stats()
    # the kernel evaluates each material texture once per path segment while the reference (and the oracle's
    # counter) evaluates the same texels up to three times, so that one counter is only bounded
    assert 0 < st.pop("texel_quads") <= cnt.pop("texel_quads")
    assert st == cnt


def test_passes_accumulate_like_the_reference_loop(gpu_ctx):
    """PASSES > 1: pass p uses subsequence p of every pixel's stream and adds into the buffers;
    the displayed image divides by the passes so far (renderer/renderer.cu:39-61)."""
    s, o = load_both("final_frame0042", 160, 90)
    gpu_ctx.upload(s)
    ref = o.new_accumulators()
    for p in range(3):
        gpu_ctx.render_pass(p, 2)
        o.render(p, 2, acc=ref)
        gpu_ctx.convert(1.0 / np.float32(p + 1), s.srgb)
        assert np.array_equal(gpu_ctx.read_image(), orc.convert(ref[0], o.srgb, 1.0 / np.float32(p + 1))), p
    assert same(gpu_ctx.read_accumulators()[0], ref[0])
    # a large pass number exercises a multi-digit jump
    gpu_ctx.clear()
    gpu_ctx.render_pass(77, 1)
    assert same(gpu_ctx.read_accumulators()[0], o.render(77, 1)[0])


def test_partition_reproduces_the_whole_image(gpu_ctx):
    s, o = load_both("teapot_low_box", 200, 75)       # ragged: 75 rows, 200 = 25 tiles
    gpu_ctx.upload(s)
    gpu_ctx.render_pass(0, 2)
    whole = gpu_ctx.read_accumulators()[0]
    assert same(whole, o.render(0, 2)[0])
    merged = np.zeros_like(whole)
    for parts in (2, 3):
        merged[:] = 0
        for part in range(parts):
            gpu_ctx.set_partition(part, parts, 16)
            gpu_ctx.clear()
            gpu_ctx.render_pass(0, 2)
            acc = gpu_ctx.read_accumulators()[0]
            rows = owned_rows(part, parts, 75)
            others = np.setdiff1d(np.arange(75), rows)
            assert not acc[others].any()          # rows of other parts are never touched
            merged[rows] = acc[rows]
        assert same(merged, whole)
    gpu_ctx.set_partition(0, 1, 16)


def test_edge_sizes(gpu_ctx):
    for w, h in ((1, 1), (7, 3), (9, 17), (64, 8), (65, 9)):
        s, o = load_both("proj4_box", w, h)
        gpu_ctx.upload(s)
        gpu_ctx.render_pass(0, 3)
        assert same(gpu_ctx.read_accumulators()[0], o.render(0, 3, threads=1)[0]), (w, h)
    # a scene whose only object is missed by every ray, and a light-less scene
    s, o = load_both("proj1_ball", 32, 18)
    gpu_ctx.upload(s)
    gpu_ctx.render_pass(0, 2)
    assert same(gpu_ctx.read_accumulators()[0], o.render(0, 2)[0])


def test_tolerance_against_glibc_flavour(gpu_ctx):
    """Against the oracle built on glibc's libm (what a host build of the reference computes):
    PSNR of the 8-bit image >= 35 dB and >= 90 % of the pixels identical, at equal seeds and spp."""
    for name, spp in (("final_frame0042", 8), ("proj7_checkerboard", 16), ("proj5_box", 8)):
        s, o = load_both(name, 320, 180, "glibc")
        gpu_ctx.upload(s)
        gpu_ctx.clear()
        gpu_ctx.render_pass(0, spp)
        gpu_ctx.convert(1.0, s.srgb)
        img = gpu_ctx.read_image()
        ref = orc.convert(o.render(0, spp)[0], o.srgb, 1.0, "glibc")
        assert psnr(img, ref) >= 35.0, (name, psnr(img, ref))
        assert np.mean((img == ref).all(axis=2)) >= 0.90, name


def test_full_size_frame_properties(gpu_ctx):
    """BASELINE.json configs[3] at full size (frame0042, 1920x1080): strips checked against the oracle,
    the whole frame through size-independent properties."""
    s, o = load_both("final_frame0042", 1920, 1080)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, 4)
    acc = gpu_ctx.read_accumulators()
    st = gpu_ctx.stats()
    ref = o.new_accumulators()
    for y0 in (0, 528, 1072):
        o.render(0, 4, acc=ref, rows=(y0, y0 + 8))
        for k in range(3):
            assert same(acc[k][y0:y0 + 8], ref[k][y0:y0 + 8]), (y0, k)
    # properties: one closest ray per path at least, shadow rays never exceed closest rays,
    # radiance finite and non-negative, determinism across runs
    assert st["closest_rays"] >= 1920 * 1080 * 4 and st["shadow_rays"] <= st["closest_rays"]
    assert np.isfinite(acc[0]).all() and (acc[0] >= 0).all()
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, 4)
    assert same(gpu_ctx.read_accumulators()[0], acc[0])
    assert gpu_ctx.stats() == st


def test_ten_million_triangle_scene_at_full_size(gpu_ctx, monkeypatch):
    """BASELINE.json configs[4] at full size: 1582 teapot copies in one BVH (9 998 240 triangles, depth 29), 3840x2160.
    The CPU checker cannot build this tree in test time (the reference's builder rescans every node 93 times), so the
    frame is checked through properties that do not depend on size, and through two independent code paths of the
    kernel that must agree bit for bit: the traversal stack split between LDS and the side buffer (the default for a
    tree this deep), almost all of it in the side buffer, and all of it in LDS."""
    src = crt.Scene.load_file(scene_path("proj5_box"))
    syn = src.make_instanced(0, 1582, seed=1, width=3840, height=2160)
    assert syn.desc().meshes[0].nf == 9998240
    results = []
    for entries in (None, "4", "64"):
        if entries is None:
            monkeypatch.delenv("CRT_STACK_ENTRIES", raising=False)
        else:
            monkeypatch.setenv("CRT_STACK_ENTRIES", entries)
        # the whole 31-entry stack of every lane fits LDS only with fewer traversal waves in the block
        if entries == "64":
            monkeypatch.setenv("CRT_TRAV_WAVES", "6")
        gpu_ctx.upload(syn)
        gpu_ctx.clear()
        gpu_ctx.render_pass(0, 1)
        results.append((gpu_ctx.read_accumulators(), gpu_ctx.stats()))
    monkeypatch.delenv("CRT_STACK_ENTRIES", raising=False)
    monkeypatch.delenv("CRT_TRAV_WAVES", raising=False)
    gpu_ctx.upload(syn)
    (acc, st) = results[0]
    for other, st_other in results[1:]:
        assert st_other == st
        for k in range(3):
            assert same(other[k], acc[k]), k
    px = 3840 * 2160
    assert st["closest_rays"] >= px and st["shadow_rays"] <= st["closest_rays"]
    assert np.isfinite(acc[0]).all() and (acc[0] >= 0).all()
    # the teapots cover a good part of the frame: normals were accumulated for hits, misses add the view direction
    assert (np.abs(acc[1]).sum(axis=2) > 0).mean() > 0.99
    # determinism (the scene was uploaded afresh with the default block shape), and a second pass accumulates on top
    # of the first (renderer.cu:39-61)
    gpu_ctx.render_pass(0, 1)
    assert same(gpu_ctx.read_accumulators()[0], acc[0])
    gpu_ctx.render_pass(1, 1)
    two = gpu_ctx.read_accumulators()[0]
    assert (two >= acc[0]).all() and (two > acc[0]).any()
    # primary hits: every pixel that accumulated an albedo hit something or saw the environment; ids are in range
    node, face, t = gpu_ctx.primary_hits(0)
    d = syn.desc()
    mesh_nodes = [i for i in range(d.node_count) if d.nodes[i].object_kind == crt._capi.OBJ_MESH]
    assert len(mesh_nodes) == 1
    on_mesh = node == mesh_nodes[0]
    assert node.max() < d.node_count and node.min() >= -2 and on_mesh.mean() > 0.2
    assert (face[on_mesh] >= 0).all() and (face[on_mesh] < 9998240).all() and (face[~on_mesh] == -1).all()
    assert (t[node >= 0] > 0).all()


def test_denoise_hand_off(gpu_ctx):
    """PrepareForDenoise and the float overload of ConvertColors (math/color.cu:25-51): element-wise float
    products and the same 8-bit conversion, so both are compared exactly."""
    import torch
    s, o = load_both("final_frame0042", 160, 90)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    passes = 3
    for p in range(passes):
        gpu_ctx.render_pass(p, 2)
    acc = gpu_ctx.read_accumulators()
    mul = np.float32(1.0) / np.float32(passes)
    bufs = [torch.empty((90, 160, 3), dtype=torch.float32, device="cuda") for _ in range(3)]
    gpu_ctx.prepare_denoise(mul, *[b.data_ptr() for b in bufs])
    for b, a in zip(bufs, acc):
        assert same(b.cpu().numpy(), a * mul)
    # a pass-through "denoiser": the beauty image straight into the float conversion
    gpu_ctx.convert_floats(bufs[0].data_ptr(), s.srgb)
    img = gpu_ctx.read_image()
    assert np.array_equal(img, orc.convert(np.ascontiguousarray(acc[0] * mul), o.srgb, 1.0))
    gpu_ctx.convert(mul, s.srgb)
    assert np.array_equal(img, gpu_ctx.read_image())
    out = torch.zeros((90, 160, 3), dtype=torch.uint8, device="cuda")
    gpu_ctx.convert_floats(bufs[0].data_ptr(), 0, out_device_ptr=out.data_ptr())
    assert np.array_equal(out.cpu().numpy(), orc.convert(np.ascontiguousarray(acc[0] * mul), 0, 1.0))
    with pytest.raises(crt.CrtError):
        gpu_ctx.prepare_denoise(mul, None, None, None)


def test_errors_through_the_abi(gpu_ctx):
    import ctypes
    from cudaraytracing_amd import _capi as capi
    fresh = crt.Context(0)
    with pytest.raises(crt.CrtError) as e:
        fresh.render_pass(0, 1)
    assert e.value.code == capi.E_NOSCENE
    s = crt.Scene.load_file(scene_path("proj4_box"))
    s.set_resolution(16, 16)
    fresh.upload(s)
    for bad in ((-1, 1), (0, 0)):
        with pytest.raises(crt.CrtError) as e:
            fresh.render_pass(*bad)
        assert e.value.code == capi.E_INVALID
    with pytest.raises(crt.CrtError):
        fresh.set_partition(2, 2, 16)
    with pytest.raises(crt.CrtError):
        fresh.set_partition(0, 2, 12)
    d = s.desc()
    d.env_texture = 999
    assert capi.lib().crt_upload_scene(fresh._h, ctypes.byref(d)) == capi.E_INVALID
    fresh.close()


def test_renderer_front_end(gpu_ctx, tmp_path):
    """The Renderer mirror (LoadScene / BeginRendering / GetImage / SaveImage) end to end."""
    r = crt.Renderer(samples=2, passes=2, width=160, height=90)
    assert not r.LoadScene("/nonexistent.xml")
    assert r.LoadScene(scene_path("final_frame0042"))
    r.BeginRendering(wait=True)
    img = r.GetImage()
    assert not r.IsRendering() and img.passes == 3 and (img.width, img.height) == (160, 90)
    o = orc.OracleScene(scene_path("final_frame0042"))
    o.set_resolution(160, 90)
    ref = o.render(1, 2, acc=o.render(0, 2))
    assert np.array_equal(img.pixels, orc.convert(ref[0], o.srgb, 0.5))
    assert img.SaveImage(tmp_path / "out.png")
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(tmp_path / "out.png")), img.pixels)


@pytest.mark.parametrize("name,w,h,spp", [("final_frame0042", 384, 216, 4), ("proj5_box", 320, 180, 4), ("proj7_checkerboard", 320, 180, 8),
                                           ("proj4_box", 320, 180, 4), ("teapot_low_box", 640, 360, 2), ("final_frame0001", 256, 144, 2),
                                           ("proj13_box", 320, 180, 4), ("proj1_snowman", 320, 180, 2), ("proj5_plane", 320, 180, 4),
                                           ("proj5_glasspot", 640, 360, 4)])
def test_scan_candidate_mask_never_drops_a_hit(gpu_ctx, name, w, h, spp):
    """The plain kernels scan only the scene nodes a conservative test cannot rule out.  The counted kernel scans
    every node, as the reference does, and checks each box it finds hit against that mask: no violation may occur,
    and the two kernels must produce the same accumulators."""
    s = crt.Scene.load_file(scene_path(name))
    s.set_resolution(w, h)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, spp)
    plain = [a.copy() for a in gpu_ctx.read_accumulators()]
    gpu_ctx.clear()
    gpu_ctx.render_pass_counted(0, spp)
    counted = gpu_ctx.read_accumulators()
    for a, b in zip(plain, counted):
        assert same(a, b)
    st = gpu_ctx.scan_check()
    assert st["checks"] > 0 and st["violations"] == 0


def _nested_scene_xml(seed, tops=5):
    """A scene graph three levels deep: object nodes with object children (the scan's subtree skip for shadow rays),
    rotations and non-uniform scales (ray drift through ToLocal / FromLocal), planes and spheres, two point lights."""
    rs = np.random.RandomState(seed)
    def obj(depth, idx):
        kind = "plane" if rs.rand() < 0.2 else "sphere"
        s = rs.uniform(0.4, 1.6, 3) * (3.0 if depth == 0 else 0.6)
        t = rs.uniform(-6, 6, 3) if depth == 0 else rs.uniform(-1.5, 1.5, 3)
        axis = rs.uniform(-1, 1, 3)
        out = [f'<object type="{kind}" name="o{depth}_{idx}" material="m{int(rs.randint(0, 3))}">',
               f'<scale x="{s[0]:.4f}" y="{s[1]:.4f}" z="{s[2]:.4f}"/>',
               f'<rotate angle="{rs.uniform(0, 360):.3f}" x="{axis[0]:.4f}" y="{axis[1]:.4f}" z="{axis[2]:.4f}"/>',
               f'<translate x="{t[0]:.4f}" y="{t[1]:.4f}" z="{t[2]:.4f}"/>']
        if depth < 2:
            for k in range(int(rs.randint(1, 4))):
                out += obj(depth + 1, idx * 10 + k)
        out.append("</object>")
        return out
    body = []
    for k in range(tops):
        body += obj(0, k)
    body.append('<object type="plane" name="floor" material="m0"><scale value="40"/><translate z="-8"/></object>')
    mats = ['<material type="blinn" name="m0"><diffuse r="0.8" g="0.7" b="0.6"/><specular value="0.2"/><glossiness value="20"/></material>',
            '<material type="blinn" name="m1"><diffuse r="0.2" g="0.5" b="0.8"/><specular value="0.7"/><glossiness value="60"/></material>',
            '<material type="blinn" name="m2"><diffuse value="0.05"/><specular value="0.1"/><glossiness value="100"/>'
            '<refraction value="0.85" index="1.5"/><absorption r="0.02" g="0.01" b="0.03"/></material>']
    lights = ['<light type="point" name="l0"><intensity value="500"/><position x="10" y="-12" z="14"/><size value="2"/></light>',
              '<light type="point" name="l1"><intensity value="200"/><position x="-9" y="-6" z="9"/><size value="1"/></light>']
    cam = ('<camera><position x="0" y="-26" z="9"/><target x="0" y="0" z="-1"/><up x="0" y="0" z="1"/><fov value="42"/>'
           '<width value="256"/><height value="160"/></camera>')
    return "<xml><scene>" + "".join(body + mats + lights) + "</scene>" + cam + "</xml>"


@pytest.mark.parametrize("seed,tops,lo,hi", [(1, 5, 20, 64), (2, 5, 20, 64), (3, 5, 20, 64), (4, 13, 64, 128), (5, 14, 64, 128),
                                             (6, 30, 128, 256), (7, 75, 256, 1024)])
def test_nested_scene_graph_is_bit_exact(gpu_ctx, tmp_path, seed, tops, lo, hi):
    """Seeds 4 and 5 build graphs of 65 to 128 nodes: the candidate mask then has two words, and subtrees that a
    shadow ray skips straddle the word boundary.  Seeds 6 and 7 build graphs of more than 128 nodes: a mask bit then
    stands for a group of 2 or 8 consecutive nodes, and subtrees straddle group boundaries."""
    xml = tmp_path / f"nested{seed}.xml"
    xml.write_text(_nested_scene_xml(seed, tops))
    s = crt.Scene.load_xml(str(xml))
    assert lo < s.desc().node_count <= hi, s.desc().node_count
    flat = str(tmp_path / f"nested{seed}.crtscene")
    s.save_file(flat)
    o = orc.OracleScene(flat)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, 4)
    acc = gpu_ctx.read_accumulators()
    ref, cnt = o.render(0, 4, count=True)
    for k, n in enumerate(("results", "normals", "albedos")):
        assert same(acc[k], ref[k]), n
    st = gpu_ctx.stats()
    assert (st["closest_rays"], st["shadow_rays"]) == (cnt["closest_rays"], cnt["shadow_rays"])
    gpu_ctx.clear()
    gpu_ctx.render_pass_counted(0, 4)
    assert same(gpu_ctx.read_accumulators()[0], ref[0])
    chk = gpu_ctx.scan_check()
    assert chk["checks"] > 0 and chk["violations"] == 0
    full = gpu_ctx.stats(); full.pop("texel_quads"); cnt.pop("texel_quads")
    assert full == cnt


def _extreme_scene_xml(case):
    """Transforms that stress the drift bound behind the candidate mask: very unequal scales (ill-conditioned round
    trips through ToLocal / FromLocal), the whole scene far from the origin (large |p|), and many rotated nodes whose
    summed bound approaches the cut-off above which the mask is switched off."""
    rs = np.random.RandomState(100 + case)
    off = {0: (0.0, 0.0, 0.0), 1: (1.0e5, -2.0e5, 5.0e4), 2: (0.0, 0.0, 0.0)}[case]
    body = []
    count = 60 if case == 2 else 18
    for k in range(count):
        kind = "plane" if k % 5 == 4 else "sphere"
        if case == 0:
            sc = [10.0 ** rs.uniform(-2, 1.2) for _ in range(3)]          # ratios up to 1600 within one node
        else:
            sc = list(rs.uniform(0.5, 2.5, 3))
        t = rs.uniform(-9, 9, 3) + np.array(off)
        axis = rs.uniform(-1, 1, 3)
        body.append(f'<object type="{kind}" name="o{k}" material="m{k % 3}">'
                    f'<scale x="{sc[0]:.6g}" y="{sc[1]:.6g}" z="{sc[2]:.6g}"/>'
                    f'<rotate angle="{rs.uniform(0, 360):.3f}" x="{axis[0]:.4f}" y="{axis[1]:.4f}" z="{axis[2]:.4f}"/>'
                    f'<translate x="{t[0]:.9g}" y="{t[1]:.9g}" z="{t[2]:.9g}"/></object>')
    body.append(f'<object type="plane" name="floor" material="m0"><scale value="60"/><translate x="{off[0]:.9g}" y="{off[1]:.9g}" z="{off[2] - 10:.9g}"/></object>')
    mats = ['<material type="blinn" name="m0"><diffuse r="0.8" g="0.7" b="0.6"/><specular value="0.2"/><glossiness value="20"/></material>',
            '<material type="blinn" name="m1"><diffuse r="0.2" g="0.5" b="0.8"/><specular value="0.7"/><glossiness value="60"/></material>',
            '<material type="blinn" name="m2"><diffuse value="0.05"/><specular value="0.1"/><glossiness value="100"/>'
            '<refraction value="0.85" index="1.5"/></material>']
    lights = [f'<light type="point" name="l0"><intensity value="600"/><position x="{off[0] + 10:.9g}" y="{off[1] - 12:.9g}" z="{off[2] + 14:.9g}"/><size value="2"/></light>']
    cam = (f'<camera><position x="{off[0]:.9g}" y="{off[1] - 30:.9g}" z="{off[2] + 9:.9g}"/><target x="{off[0]:.9g}" y="{off[1]:.9g}" z="{off[2] - 1:.9g}"/>'
           '<up x="0" y="0" z="1"/><fov value="42"/><width value="224"/><height value="140"/></camera>')
    return "<xml><scene>" + "".join(body + mats + lights) + "</scene>" + cam + "</xml>"


@pytest.mark.parametrize("case", [0, 1, 2])
def test_candidate_mask_under_extreme_transforms(gpu_ctx, tmp_path, case):
    """Whether the mask is in use for such a scene is the upload's decision (it is switched off when the drift bound is
    too large to be useful); either way the plain kernel must equal the oracle, and where the mask is on, the counted
    kernel's check of every box hit against it must find no violation."""
    xml = tmp_path / f"extreme{case}.xml"
    xml.write_text(_extreme_scene_xml(case))
    s = crt.Scene.load_xml(str(xml))
    flat = str(tmp_path / f"extreme{case}.crtscene")
    s.save_file(flat)
    o = orc.OracleScene(flat)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, 3)
    acc = gpu_ctx.read_accumulators()
    ref, cnt = o.render(0, 3, count=True)
    for k, n in enumerate(("results", "normals", "albedos")):
        assert same(acc[k], ref[k]), n
    st = gpu_ctx.stats()
    assert (st["closest_rays"], st["shadow_rays"]) == (cnt["closest_rays"], cnt["shadow_rays"])
    gpu_ctx.clear()
    gpu_ctx.render_pass_counted(0, 3)
    assert same(gpu_ctx.read_accumulators()[0], ref[0])
    assert gpu_ctx.scan_check()["violations"] == 0


from cudaraytracing_amd.synthetic import instance_grid_xml as _instance_grid_xml      # (tools import it from here too)


@pytest.mark.parametrize("n_side", [14, 40])
def test_many_instances_of_one_mesh(gpu_ctx, tmp_path, n_side):
    """198 and 1602 scene nodes that share one BVH: the scan goes by groups of nodes (2 and 16 per mask bit).  Plain
    kernel, counted kernel (full scan, checks every box hit against the group mask) and oracle must agree bit for bit."""
    s = crt.Scene.load_xml(_instance_grid_xml(tmp_path, n_side), str(tmp_path))
    assert s.desc().node_count == n_side * n_side + 2 and s.desc().mesh_count == 1
    flat = str(tmp_path / "grid.crtscene")
    s.save_file(flat)
    o = orc.OracleScene(flat)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, 2)
    acc = gpu_ctx.read_accumulators()
    ref, cnt = o.render(0, 2, count=True)
    for k, n in enumerate(("results", "normals", "albedos")):
        assert same(acc[k], ref[k]), n
    st = gpu_ctx.stats()
    assert (st["closest_rays"], st["shadow_rays"]) == (cnt["closest_rays"], cnt["shadow_rays"])
    gpu_ctx.clear()
    gpu_ctx.render_pass_counted(0, 2)
    assert same(gpu_ctx.read_accumulators()[0], ref[0])
    chk = gpu_ctx.scan_check()
    assert chk["checks"] > 0 and chk["violations"] == 0
    full = gpu_ctx.stats(); full.pop("texel_quads"); cnt.pop("texel_quads")
    assert full == cnt


def test_first_scan_step_cases(gpu_ctx, tmp_path):
    """The shading wave takes the first step of every query's node scan for scenes of up to 64 nodes.  This scene puts
    each kind of first node in front of many rays: a mesh whose BVH root is a leaf (two triangles: left to the traversal
    lane), a mesh with an interior root, spheres and planes with children (a shadow ray that misses one skips its
    subtree), spheres between the lights and the floor (shadow queries blocked by their first node), glass."""
    _instance_grid_xml(tmp_path, 1)                     # writes blob.obj (128 triangles)
    (tmp_path / "quad.obj").write_text("v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nf 1 2 3\nf 1 3 4\n")
    rs = np.random.RandomState(11)
    body = []
    for k in range(6):
        x, y = rs.uniform(-5, 5, 2)
        body.append(f'<object type="obj" name="quad.obj" material="m{k % 3}"><scale value="{rs.uniform(0.6, 1.4):.3f}"/>'
                    f'<rotate angle="{rs.uniform(0, 360):.2f}" x="{rs.uniform(-1, 1):.3f}" y="{rs.uniform(-1, 1):.3f}" z="1"/>'
                    f'<translate x="{x:.3f}" y="{y:.3f}" z="{rs.uniform(0.5, 3):.3f}"/></object>')
    for k in range(8):
        x, y = rs.uniform(-6, 6, 2)
        kind = "sphere" if k % 4 else "plane"
        body.append(f'<object type="{kind}" name="s{k}" material="m{(k + 1) % 3}"><scale x="{rs.uniform(0.5, 1.2):.3f}" y="{rs.uniform(0.5, 1.2):.3f}" z="{rs.uniform(0.5, 1.2):.3f}"/>'
                    f'<rotate angle="{rs.uniform(0, 360):.2f}" x="{rs.uniform(-1, 1):.3f}" y="1" z="{rs.uniform(-1, 1):.3f}"/>'
                    f'<translate x="{x:.3f}" y="{y:.3f}" z="{rs.uniform(1, 5):.3f}"/>'
                    f'<object type="obj" name="blob.obj" material="m{k % 3}"><scale value="0.5"/><translate x="0.4" y="-0.3" z="1.6"/></object>'
                    f'<object type="sphere" name="c{k}" material="m2"><scale value="0.4"/><translate x="-1.2" z="0.3"/></object></object>')
    body.append('<object type="plane" name="floor" material="m0"><scale value="30"/></object>')
    mats = ['<material type="blinn" name="m0"><diffuse r="0.8" g="0.7" b="0.6"/><specular value="0.2"/><glossiness value="20"/></material>',
            '<material type="blinn" name="m1"><diffuse r="0.2" g="0.5" b="0.8"/><specular value="0.7"/><glossiness value="60"/></material>',
            '<material type="blinn" name="m2"><diffuse value="0.05"/><specular value="0.1"/><glossiness value="100"/>'
            '<refraction value="0.85" index="1.5"/><absorption r="0.02" g="0.01" b="0.03"/></material>']
    lights = ['<light type="point" name="l0"><intensity value="400"/><position x="4" y="-6" z="12"/><size value="1.5"/></light>',
              '<light type="point" name="l1"><intensity value="150"/><position x="-7" y="-3" z="7"/><size value="0.5"/></light>']
    cam = ('<camera><position x="0" y="-17" z="8"/><target x="0" y="0" z="1"/><up x="0" y="0" z="1"/><fov value="44"/>'
           '<width value="288"/><height value="176"/></camera>')
    (tmp_path / "first.xml").write_text("<xml><scene>" + "".join(body + mats + lights) + "</scene>" + cam + "</xml>")
    s = crt.Scene.load_xml(str(tmp_path / "first.xml"), str(tmp_path))
    assert 30 < s.desc().node_count <= 64 and s.desc().mesh_count == 2
    flat = str(tmp_path / "first.crtscene")
    s.save_file(flat)
    o = orc.OracleScene(flat)
    gpu_ctx.upload(s)
    gpu_ctx.clear()
    gpu_ctx.render_pass(0, 6)
    acc = gpu_ctx.read_accumulators()
    ref, cnt = o.render(0, 6, count=True)
    for k, n in enumerate(("results", "normals", "albedos")):
        assert same(acc[k], ref[k]), n
    st = gpu_ctx.stats()
    assert (st["closest_rays"], st["shadow_rays"]) == (cnt["closest_rays"], cnt["shadow_rays"])
    assert cnt["sphere_tests"] > 0 and cnt["plane_tests"] > 0 and cnt["tri_tests"] > 0
    gpu_ctx.clear()
    gpu_ctx.render_pass_counted(0, 6)
    assert same(gpu_ctx.read_accumulators()[0], ref[0])
    chk = gpu_ctx.scan_check()
    assert chk["checks"] > 0 and chk["violations"] == 0
    full = gpu_ctx.stats(); full.pop("texel_quads"); cnt.pop("texel_quads")
    assert full == cnt


@pytest.mark.parametrize("name,w,h,spp", [("final_frame0042", 640, 360, 8), ("proj5_glasspot", 480, 270, 6), ("proj7_checkerboard", 480, 270, 8)])
def test_paired_queries_change_nothing(gpu_ctx, monkeypatch, name, w, h, spp):
    """A pixel's shadow query and the closest-hit query of its next path segment may run side by side or one after the
    other; the kernel chooses by how many pixels its blocks have left.  Forced either way, accumulators and ray counts
    must equal the oracle's bit for bit (at this size the default mixes both: the blocks run out of pixels early)."""
    s, o = load_both(name, w, h)
    ref, cnt = o.render(0, spp, count=True)
    for setting in ("0", "100000", None):
        if setting is None:
            monkeypatch.delenv("CRT_PAIR_LIVE", raising=False)
        else:
            monkeypatch.setenv("CRT_PAIR_LIVE", setting)      # never / always
        gpu_ctx.upload(s)
        gpu_ctx.clear()
        gpu_ctx.render_pass(0, spp)
        acc = gpu_ctx.read_accumulators()
        for k, n in enumerate(("results", "normals", "albedos")):
            assert same(acc[k], ref[k]), (setting, n)
        st = gpu_ctx.stats()
        assert (st["closest_rays"], st["shadow_rays"]) == (cnt["closest_rays"], cnt["shadow_rays"]), setting
    monkeypatch.delenv("CRT_PAIR_LIVE", raising=False)
    gpu_ctx.upload(s)


def test_reciprocal_equals_division_for_every_float(gpu_ctx):
    """The kernels compute 1 / x as the hardware estimate plus one fused Newton step, with the division itself for
    operands or results outside the normal range.  All 2^32 bit patterns are compared on the device; a sample of
    them, and the special values, are also compared with numpy's division here."""
    differ = gpu_ctx.math_selftest(100, np.zeros(1, np.float32))
    assert differ[0] == 0
    rs = np.random.RandomState(5)
    x = rs.randint(0, 2 ** 32, 1 << 20, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x[:12] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38, 3.4028235e38, 8.5070587e37, 8.5070597e37, 1.0]
    got = gpu_ctx.math_selftest(8, x)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        want = np.float32(1.0) / x
    nan = np.isnan(want)
    assert (np.isnan(got) == nan).all()
    assert (got.view(np.uint32)[~nan] == want.view(np.uint32)[~nan]).all()

