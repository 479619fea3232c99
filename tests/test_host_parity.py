"""Host half of the product (loader, BVH builder, flat file, PNG) checked without a GPU."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, scene_path
import oracle_lib as orc

import cudaraytracing_amd as crt
from cudaraytracing_amd import _capi as capi


MESH_SCENES = ["teapot_low_box", "proj5_box", "proj5_glasspot", "final_frame0001", "final_frame0042"]


@pytest.mark.parametrize("name", MESH_SCENES)
def test_bvh_builder_equals_oracle(name):
    """Node arrays and element order of the fast builder equal the plain restatement, field by field."""
    s = crt.Scene.load_file(scene_path(name))
    o = orc.OracleScene(scene_path(name))
    d = s.desc()
    assert d.mesh_count == o.meshes
    for m in range(d.mesh_count):
        mine = crt.build_bvh(*s.mesh_arrays(m))
        ref = o.bvh(m)
        assert mine["node_count"] == ref["node_count"] and mine["depth"] == ref["depth"]
        for k in ("leaf", "start", "count", "left", "right", "next", "elements"):
            assert np.array_equal(mine[k].astype(np.int64), ref[k].astype(np.int64)), (name, m, k)
        assert np.array_equal(mine["box"], ref["box"]), (name, m)       # == treats -0 and +0 alike


def test_bvh_builder_edge_cases():
    # one triangle -> single leaf; no triangles -> empty leaf; identical triangles -> still terminates
    v = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], np.float32)
    b = crt.build_bvh(v, np.array([0, 1, 2], np.uint32), [0, 0, 0], [1, 1, 0])
    assert b["node_count"] == 1 and b["leaf"][0] == 1 and b["count"][0] == 1 and b["depth"] == 1
    b = crt.build_bvh(v, np.zeros(0, np.uint32), [0, 0, 0], [1, 1, 0])
    assert b["node_count"] == 1 and b["leaf"][0] == 1 and b["count"][0] == 0
    b = crt.build_bvh(v, np.tile(np.array([0, 1, 2], np.uint32), 8), [0, 0, 0], [1, 1, 0])
    assert b["leaf"][0] == 1 and b["count"][0] == 8
    with pytest.raises(crt.CrtError):
        crt.build_bvh(v, np.array([0, 1, 7], np.uint32), [0, 0, 0], [1, 1, 0])      # index out of range
    with pytest.raises(crt.CrtError) as e:                                            # 9 coincident triangles never separate
        crt.build_bvh(v, np.tile(np.array([0, 1, 2], np.uint32), 9), [0, 0, 0], [1, 1, 0])
    assert e.value.code == capi.E_LIMIT


def test_camera_frame_equals_oracle_after_resize():
    for name in ("proj1_ball", "final_frame0042", "proj7_checkerboard"):
        s = crt.Scene.load_file(scene_path(name))
        o = orc.OracleScene(scene_path(name))
        s.set_resolution(1920, 1080)
        o.set_resolution(1920, 1080)
        c = s.desc().camera
        mine = np.array([c.pixel_size, *c.cX, *c.cY, *c.cZ, *c.top_left_pixel, *c.position, c.dof], np.float32)
        assert np.array_equal(mine, o.camera()[:17]), name


XML = """<xml><scene>
  <!-- comment -->
  <object type="sphere" name="a" material="m1"><scale value="2"/><translate x="1" y="2" z="3"/>
     <object type="plane" name="child"><rotate angle="90" z="1"/><scale x="2" y="3" z="4"/><scale x="1" y="1" z="1"/></object>
  </object>
  <object name="group"><object type="obj" name="%(obj)s" material="m1"/></object>
  <object type="obj" name="%(obj)s"/>
  <material type="blinn" name="m1"><diffuse r="0.5" g="0.25" b="2"/><specular value="0"/><glossiness value="7"/>
     <refraction index="1.5" value="0.125"/><absorption r="0.1" g="0.2" b="0.3"/></material>
  <material type="blinn" name="m1"><diffuse r="1" g="1" b="1"/></material>
  <light type="point" name="p"><intensity value="3"/><position x="1" y="2" z="3"/></light>
  <light type="direct" name="d"><intensity value="0.5"/><direction x="0" y="0" z="-2"/></light>
  <light type="ambient" name="a"><intensity r="0.1" g="0.2" b="0.3"/></light>
</scene>
<camera gamma="sRGB"><position x="0" y="0" z="10"/><target x="0" y="0" z="0"/><up x="0" y="1" z="0"/>
  <fov value="90"/><width value="64"/><height value="32"/><dof value="0.5"/></camera></xml>"""

OBJ = """# a quad and a triangle with negative indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vt 1 1
vn 0 0 1
f 1/1/1 2/2/1 3/3/1 4/1/1
f -4/-3/-1 -3/-2/-1 -2/-1/-1
"""


def test_xml_loader_semantics(tmp_path):
    (tmp_path / "q.obj").write_text(OBJ)
    (tmp_path / "s.xml").write_text(XML % {"obj": "q.obj"})
    s = crt.Scene.load_xml(tmp_path / "s.xml", str(tmp_path))
    d = s.desc()
    # root + a + child + group + mesh + mesh, depth first; the mesh file is loaded once
    assert d.node_count == 6 and d.mesh_count == 1 and d.light_count == 3
    kinds = [d.nodes[i].object_kind for i in range(6)]
    assert kinds == [capi.OBJ_NONE, capi.OBJ_SPHERE, capi.OBJ_PLANE, capi.OBJ_NONE, capi.OBJ_MESH, capi.OBJ_MESH]
    assert [d.nodes[i].child_count for i in range(6)] == [3, 1, 0, 1, 0, 0]
    # material 0 is the default; a repeated name resolves to its first definition
    assert d.material_count == 3 and d.nodes[1].material_id == 1 and d.nodes[5].material_id == 0
    m = d.materials[1]
    assert (m.glossiness, m.ior) == (7.0, 1.5) and np.allclose(m.absorption[:], [0.1, 0.2, 0.3])
    tex = lambda i: list(d.textures[i].rgb[0:3])
    assert tex(m.diffuse) == [128, 64, 255] and tex(m.specular) == [0, 0, 0] and tex(m.refraction) == [32, 32, 32]
    assert tex(d.materials[0].diffuse) == [255, 255, 255] and tex(d.env_texture) == [0, 0, 0]
    # transforms: tm = T * R * S whatever the order in the file, the last element of a kind wins
    a = np.array(d.nodes[1].tm[:])
    assert np.allclose(a, [2, 0, 0, 0, 2, 0, 0, 0, 2, 1, 2, 3])
    c = np.array(d.nodes[2].tm[:]).reshape(4, 3)
    assert np.allclose(c[:3], 2 * np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1]]), atol=1e-6) and np.allclose(c[3], [1, 2, 3])
    itm = np.array(d.nodes[1].itm[:])
    assert np.allclose(itm, [0.5, 0, 0, 0, 0.5, 0, 0, 0, 0.5, -0.5, -1, -1.5])
    # boxes: the sphere node's box also covers its child plane; the group box is its child's
    assert np.allclose(d.nodes[1].box_min[:], [-1, 0, 1]) and np.allclose(d.nodes[1].box_max[:], [3, 4, 5])
    assert np.allclose(d.nodes[3].box_min[:], d.nodes[4].box_min[:])
    # lights
    L = [d.lights[i] for i in range(3)]
    assert [l.kind for l in L] == [capi.LIGHT_POINT, capi.LIGHT_DIRECT, capi.LIGHT_AMBIENT]
    assert L[0].size == 1.0 and list(L[0].intensity) == [3, 3, 3] and list(L[1].vec) == [0, 0, -1]
    # camera
    cam = d.camera
    assert (cam.width, cam.height, cam.srgb, cam.dof) == (64, 32, 1, 0.5)
    assert np.isclose(cam.pixel_size, 20.0 / 32) and np.allclose(cam.cX[:], [1, 0, 0]) and np.allclose(cam.cY[:], [0, 1, 0])
    assert np.allclose(cam.top_left_pixel[:], [-20 + 0.3125, 10 - 0.3125, 0])
    # mesh: quad fan-triangulated, negative indices resolved, third texture coordinate zero
    me = d.meshes[0]
    assert (me.nv, me.nf, me.nvn, me.nvt) == (4, 3, 1, 3)
    assert list(me.f[0:9]) == [0, 1, 2, 0, 2, 3, 0, 1, 2]
    assert list(me.ft[0:9]) == [0, 1, 2, 0, 2, 0, 0, 1, 2] and list(me.fn[0:9]) == [0] * 9
    assert list(me.vt[0:9]) == [0, 0, 0, 1, 0, 0, 1, 1, 0]
    # flat file round trip is lossless
    s.save_file(tmp_path / "s.crtscene")
    s2 = crt.Scene.load_file(tmp_path / "s.crtscene")
    r = s2.desc()
    assert bytes(ctypes.string_at(ctypes.byref(r.camera), ctypes.sizeof(r.camera))) == bytes(ctypes.string_at(ctypes.byref(d.camera), ctypes.sizeof(d.camera)))
    assert bytes(ctypes.string_at(r.nodes, 6 * ctypes.sizeof(capi.Node))) == bytes(ctypes.string_at(d.nodes, 6 * ctypes.sizeof(capi.Node)))
    assert list(r.meshes[0].f[0:9]) == list(me.f[0:9])


def test_xml_errors(tmp_path):
    for text, msg in [("<xml><scene/></xml>", "camera"), ("<xml><camera/></xml>", "scene"), ("<scene/>", "xml"),
                      ("<xml><scene><object material='nope' type='sphere'/></scene><camera><width value='4'/><height value='4'/></camera></xml>", "nope"),
                      ("<xml><scene></scene><camera><width value='0'/><height value='4'/></camera></xml>", "width"),
                      ("<xml><scene>", "end tag")]:
        p = tmp_path / "e.xml"
        p.write_text(text)
        with pytest.raises(crt.CrtError) as e:
            crt.Scene.load_xml(p)
        assert msg in str(e.value), text
    # a missing mesh or texture is not fatal: the node stays empty / the texture falls back to white * colour
    p = tmp_path / "m.xml"
    p.write_text("<xml><scene><object type='obj' name='missing.obj'/><material name='m'><diffuse texture='missing.png' r='0.5'/></material>"
                 "</scene><camera><width value='4'/><height value='4'/><fov value='40'/><target z='-1'/><up y='1'/></camera></xml>")
    sc = crt.Scene.load_xml(p)
    d = sc.desc()
    assert d.nodes[1].object_kind == capi.OBJ_NONE and list(d.textures[d.materials[1].diffuse].rgb[0:3]) == [128, 255, 255]


def test_png_round_trip(tmp_path):
    L = capi.lib()
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    path = str(tmp_path / "x.png").encode()
    assert L.crt_png_write_rgb8(path, capi.np_ptr(img), 53, 37) == 0
    p = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    assert L.crt_png_read_rgb8(path, ctypes.byref(p), ctypes.byref(w), ctypes.byref(h)) == 0
    back = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    L.crt_free(p)
    assert (w.value, h.value) == (53, 37) and np.array_equal(back, img)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(path.decode()).convert("RGB")), img)
    # every filter type and RGBA / 16-bit / palette inputs decode like PIL's RGB conversion of the high byte
    for mode in ("RGBA", "L", "P", "LA"):
        src = Image.fromarray(img).convert(mode)
        src.save(tmp_path / f"{mode}.png")
        assert L.crt_png_read_rgb8(str(tmp_path / f"{mode}.png").encode(), ctypes.byref(p), ctypes.byref(w), ctypes.byref(h)) == 0
        got = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
        L.crt_free(p)
        assert np.array_equal(got, np.asarray(src.convert("RGB"))), mode
    assert L.crt_png_read_rgb8(b"/nonexistent.png", ctypes.byref(p), ctypes.byref(w), ctypes.byref(h)) == capi.E_IO


def test_instanced_scene_generator(tmp_path):
    """The synthetic instanced scene (BASELINE.json configs[4] at reduced size) is a fixed function of its arguments,
    survives the flat-file round trip, and its BVH (deeper than any shipped mesh) equals the oracle's."""
    import hashlib
    src = crt.Scene.load_file(scene_path("proj5_box"))
    syn = src.make_instanced(0, 20, seed=1, width=320, height=180)
    d = syn.desc()
    assert (d.node_count, d.mesh_count, d.light_count, d.material_count) == (3, 1, 1, 3)
    v, f, lo, hi = syn.mesh_arrays(0)
    assert f.size == 3 * 20 * 6320 and v.size == 3 * 20 * 3241
    assert np.array_equal(np.asarray(lo), v.reshape(-1, 3).min(0)) and np.array_equal(np.asarray(hi), v.reshape(-1, 3).max(0))
    again = src.make_instanced(0, 20, seed=1, width=320, height=180).mesh_arrays(0)[0]
    other = src.make_instanced(0, 20, seed=2, width=320, height=180).mesh_arrays(0)[0]
    assert np.array_equal(v, again) and not np.array_equal(v, other)
    assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] == INSTANCED20_VERTEX_SHA
    # copies keep their shape: every triangle of copy k is a similarity image of the source triangle
    sv, sf, _, _ = src.mesh_arrays(0)
    e_src = np.linalg.norm(sv.reshape(-1, 3)[sf.reshape(-1, 3)[:, 1]] - sv.reshape(-1, 3)[sf.reshape(-1, 3)[:, 0]], axis=1)
    k = 7
    fk = f.reshape(-1, 3)[k * 6320:(k + 1) * 6320]
    e_k = np.linalg.norm(v.reshape(-1, 3)[fk[:, 1]] - v.reshape(-1, 3)[fk[:, 0]], axis=1)
    ratio = e_k[e_src > 0] / e_src[e_src > 0]
    assert 0.5 <= ratio.min() and ratio.max() <= 1.0 and ratio.max() - ratio.min() < 1e-4

    path = str(tmp_path / "instanced20.crtscene")
    syn.save_file(path)
    o = orc.OracleScene(path)
    mine = crt.build_bvh(v, f, lo, hi)
    ref = o.bvh(0)
    assert mine["depth"] == ref["depth"] > 17 and mine["node_count"] == ref["node_count"]
    for key in ("leaf", "start", "count", "left", "right", "next", "elements"):
        assert np.array_equal(mine[key].astype(np.int64), ref[key].astype(np.int64)), key
    assert np.array_equal(mine["box"], ref["box"])
    with pytest.raises(crt.CrtError):
        src.make_instanced(5, 20)                      # no such mesh
    with pytest.raises(crt.CrtError):
        src.make_instanced(0, 1 << 20)                 # more triangles than a leaf reference can address


INSTANCED20_VERTEX_SHA = "b4389332649debbf"      # generated by this build; pins the generator against silent changes


def test_parallel_bvh_build_equals_sequential(monkeypatch):
    """Above 2^18 triangles the builder works on several threads; node numbering, boxes and element order must
    not depend on the thread count (1 thread = the sequential pre-order pass)."""
    src = crt.Scene.load_file(scene_path("proj5_box"))
    v, f, lo, hi = src.make_instanced(0, 45, seed=3, width=64, height=36).mesh_arrays(0)
    assert f.size // 3 > 1 << 18
    built = []
    for threads in ("1", "5", "16"):
        monkeypatch.setenv("CRT_BUILD_THREADS", threads)
        built.append(crt.build_bvh(v, f, lo, hi))
    for other in built[1:]:
        assert other["node_count"] == built[0]["node_count"] and other["depth"] == built[0]["depth"]
        for key in ("leaf", "start", "count", "left", "right", "next", "elements", "box"):
            assert np.array_equal(other[key], built[0][key]), key
    # pre-order numbering: the left child follows its parent, next = first node after the subtree
    b = built[0]
    inner = np.flatnonzero(b["leaf"] == 0)
    assert np.array_equal(b["left"][inner], inner + 1)
    assert np.array_equal(b["next"][b["left"][inner]], b["right"][inner])
