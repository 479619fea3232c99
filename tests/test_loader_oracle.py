"""The product's scene front end (XML, OBJ, PNG readers, transforms, boxes, camera) against an independent Python
restatement of the reference's loader (tests/loader_oracle.py), bit for bit.  CPU only."""
import os

import numpy as np
import pytest

import cudaraytracing_amd as crt
import loader_oracle as lo

REF = "/root/reference"


def _write_fixture(tmp_path):
    """A small scene that uses every element of the schema, with awkward number formats, and a mesh file that uses
    every quirk of the reference's OBJ reader."""
    from PIL import Image
    rs = np.random.RandomState(7)
    os.makedirs(tmp_path / "tex"); os.makedirs(tmp_path / "mesh")
    Image.fromarray(rs.randint(0, 256, (5, 9, 3), dtype=np.uint8), "RGB").save(tmp_path / "tex" / "noise.png")
    Image.fromarray(rs.randint(0, 256, (4, 4, 4), dtype=np.uint8), "RGBA").save(tmp_path / "tex" / "alpha.PNG")
    Image.fromarray(rs.randint(0, 256, (3, 6), dtype=np.uint8), "L").save(tmp_path / "tex" / "grey.png")
    verts = rs.uniform(-2, 2, (40, 3))
    lines = ["# a comment", "", "   \t # indented comment", "o thing", "mtllib nothing.mtl"]
    lines += [f"v {x:.6f}\t{y:.7g}   {z:.5e}" for x, y, z in verts[:30]]
    lines += ["v 1 2", "v 3"]                                     # missing components
    lines += [f"v {x} {y} {z} 1.0" for x, y, z in verts[32:]]    # a fourth number is ignored
    lines += [f"vt {u:.4f} {v:.4f}" for u, v in rs.uniform(0, 1, (12, 2))]
    lines += [f"vn {x:.4f} {y:.4f} {z:.4f}" for x, y, z in rs.uniform(-1, 1, (9, 3))]
    # (all vertex data first: the product refuses a file whose faces start before its first vt / vn record, where the
    # reference would pair faces with the index triples of other faces)
    lines += ["f 1 2 3", "f 4 5 6 7", "f -1 -2 -3", "f 1 2 3 4 5 6 7 8"]
    lines += ["f 1/1 2/2 3/3", "f 4/4 5/5 6/6 7/7"]
    lines += ["f 1/1/1 2/2/2 3/3/3", "f 4//4 5//5 6//6", "f -3/-2/-1 -6/-5/-4 -9/-8/-7", "f 10/3/2 11/4/3 12/5/4 13/6/5 14/7/6",
              "s off", "g group", "usemtl none", "vp 1 2 3", "f   20   21\t22  "]
    # a polygon whose record is longer than the reader's line buffer
    lines += ["f " + " ".join(f"{1 + (k % 38)}/{1 + (k % 12)}/{1 + (k % 9)}" for k in range(180))]
    lines += ["f 30 31 32"]
    (tmp_path / "mesh" / "odd.obj").write_bytes(("\r\n".join(lines) + "\n").encode())
    (tmp_path / "mesh" / "tri.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 3 4\nf 1 4 2\nf 2 4 3")
    xml = """<xml>
  <scene>
    <object type="sphere" name="a" material="red"><scale value="2.5"/><translate x="1e0" y="-.5" z="3."/>
      <object type="PLANE" name="b" material="mirror"><rotate angle="33.3" x="1" y="2" z="3"/><scale x="4" y="0.25"/>
        <object type="obj" name="mesh/odd.obj" material="glass"><translate x="0.1" y="0.2" z="0.3" value="10"/>
          <rotate angle="-100" z="1"/><rotate angle="271.25" x="-1" y="0.5"/></object>
      </object>
      <object type="obj" name="mesh/tri.obj"><scale x="-1" y="1" z="-1"/></object>
    </object>
    <object name="group"><translate z="-4"/>
      <object type="obj" name="mesh/odd.obj" material="tex"><scale value="0.125"/><rotate angle="45" y="1"/></object>
      <object type="sphere" material="red"/>
    </object>
    <object type="obj" name="mesh/tri.obj" material="grey"><rotate angle="720.5" x="0.3" y="0.3" z="0.3"/></object>
    <material type="blinn" name="red"><diffuse r="0.9" g="0.15" b="0.1"/><specular value="0.3"/><glossiness value="55.5"/></material>
    <material type="blinn" name="mirror"><diffuse value="0"/><specular r="0.9" g="0.95" b="1.2"/><absorption value="0.5" r="0.1"/></material>
    <material type="blinn" name="glass"><refraction value="0.8" index="1.52"/><emission r="2" g="-1" b="300"/><absorption r="0.01" g="0.02" b="0.03"/></material>
    <material type="blinn" name="tex"><diffuse texture="tex/noise.png" r="0.5" g="1.5"/><specular texture="tex/alpha.PNG" value="0.75"/><emission texture="tex/missing.png" value="0.2"/></material>
    <material type="blinn" name="grey"><diffuse texture="tex/grey.png"/><refraction texture="tex/noise.png" value="0.4" index="1.1"/></material>
    <material type="blinn" name="red"><diffuse value="0.5"/></material>
    <light type="ambient" name="amb"><intensity value="0.2"/></light>
    <light type="direct" name="dir"><intensity r="0.5" g="0.6" b="0.7"/><direction x="1" y="-2" z="-3"/></light>
    <light type="point" name="pt"><intensity value="40" r="1" g="0.9" b="0.8"/><position x="3" y="4" z="5"/><size value="0.75"/></light>
    <light type="point" name="pt2"><position x="-3" z="5" value="2"/></light>
    <environment texture="tex/noise.png" value="0.9"/>
  </scene>
  <camera gamma="sRGB">
    <position x="0.5" y="-12" z="4.25"/><target x="0" y="0.1" z="1"/><up x="0.1" y="0" z="1"/><fov value="37.5"/>
    <width value="321"/><height value="123"/><dof value="0.05"/>
  </camera>
</xml>"""
    (tmp_path / "scene.xml").write_text(xml)
    return str(tmp_path / "scene.xml")


def test_front_end_equals_the_independent_restatement_on_a_scene_with_every_feature(tmp_path):
    path = _write_fixture(tmp_path)
    ref = lo.load_scene(path, str(tmp_path))
    s = crt.Scene.load_xml(path, str(tmp_path))
    assert len(ref["nodes"]) == 9 and len(ref["meshes"]) == 2 and len(ref["materials"]) == 7
    odd = ref["meshes"]["mesh/odd.obj"]
    assert odd["ft"] is not None and odd["fn"] is not None and len(odd["f"]) > 150       # fans, and the cut record
    assert lo.compare(s, ref) == []


def test_the_comparison_sees_single_bit_differences(tmp_path):
    path = _write_fixture(tmp_path)
    ref = lo.load_scene(path, str(tmp_path))
    s = crt.Scene.load_xml(path, str(tmp_path))
    ref["nodes"][3].itm[7] = np.nextafter(ref["nodes"][3].itm[7], np.float32(9))
    ref["nodes"][5].box.hi[1] = np.nextafter(ref["nodes"][5].box.hi[1], np.float32(-9))
    ref["materials"][4]["diffuse"][2][3, 1] ^= 1
    ref["camera"]["top_left_pixel"][2] = np.nextafter(ref["camera"]["top_left_pixel"][2], np.float32(9))
    ref["meshes"]["mesh/odd.obj"]["fn"][4, 0] += 1
    ref["lights"][2]["size"] = np.float32(0.76)
    diffs = lo.compare(s, ref)
    assert len(diffs) == 6, diffs


def _reference_scenes():
    if not os.path.isdir(REF + "/scenes"):
        return []
    out = []
    for d in sorted(os.listdir(REF + "/scenes")):
        files = sorted(f for f in os.listdir(f"{REF}/scenes/{d}") if f.endswith(".xml"))
        out += [f"{d}/{f}" for f in (files[::12] if d == "final" else files)]       # tools/check_loaders.py runs all of them
    return out


@pytest.mark.skipif(not os.path.isdir(REF + "/scenes"), reason="the reference's scene data is not on this machine")
def test_front_end_equals_the_independent_restatement_on_the_reference_scenes():
    import functools
    load_obj = lo.load_obj
    lo.load_obj = functools.lru_cache(maxsize=None)(load_obj)
    try:
        bad = {}
        names = _reference_scenes()
        assert len(names) >= 30
        for name in names:
            path = f"{REF}/scenes/{name}"
            diffs = lo.compare(crt.Scene.load_xml(path, REF), lo.load_scene(path, REF))
            if diffs:
                bad[name] = diffs[:3]
        assert bad == {}
    finally:
        lo.load_obj = load_obj
