"""Synthetic scenes in the reference's XML scene format, for tests, tools and the benchmark's many-node workload."""
import pathlib

import numpy as np


def instance_grid_xml(tmp_path, n_side):
    """n_side x n_side instances of one small mesh (one BVH shared by all of them), each rotated and scaled on its own,
    over a floor: the scene-node list is far longer than the candidate mask has bits."""
    rs = np.random.RandomState(11)
    # a bumpy octahedron, subdivided twice (128 triangles)
    verts = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    faces = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    for _ in range(2):
        mid, out = {}, []
        def midpoint(a, b):
            key = (min(a, b), max(a, b))
            if key not in mid:
                m = np.add(verts[a], verts[b]) * 0.5
                verts.append(tuple(m / np.linalg.norm(m)))
                mid[key] = len(verts) - 1
            return mid[key]
        for a, b, c in faces:
            ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
            out += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        faces = out
    verts = [tuple(np.array(v) * (1 + 0.15 * rs.rand())) for v in verts]
    (pathlib.Path(tmp_path) / "blob.obj").write_text("".join(f"v {x:.6f} {y:.6f} {z:.6f}\n" for x, y, z in verts) +
                                       "".join(f"f {a + 1} {b + 1} {c + 1}\n" for a, b, c in faces))
    body = []
    for j in range(n_side):
        for i in range(n_side):
            sc = rs.uniform(0.25, 0.45, 3)
            axis = rs.uniform(-1, 1, 3)
            body.append(f'<object type="obj" name="blob.obj" material="m{(i + j) % 3}">'
                        f'<scale x="{sc[0]:.4f}" y="{sc[1]:.4f}" z="{sc[2]:.4f}"/>'
                        f'<rotate angle="{rs.uniform(0, 360):.2f}" x="{axis[0]:.3f}" y="{axis[1]:.3f}" z="{axis[2]:.3f}"/>'
                        f'<translate x="{i - n_side / 2 + 0.5:.3f}" y="{j - n_side / 2 + 0.5:.3f}" z="{0.5 + 0.4 * rs.rand():.3f}"/></object>')
    body.append(f'<object type="plane" name="floor" material="m0"><scale value="{n_side}"/></object>')
    mats = ['<material type="blinn" name="m0"><diffuse r="0.8" g="0.7" b="0.6"/><specular value="0.2"/><glossiness value="20"/></material>',
            '<material type="blinn" name="m1"><diffuse r="0.2" g="0.5" b="0.8"/><specular value="0.7"/><glossiness value="60"/></material>',
            '<material type="blinn" name="m2"><diffuse value="0.05"/><specular value="0.1"/><glossiness value="100"/>'
            '<refraction value="0.85" index="1.5"/></material>']
    lights = [f'<light type="point" name="l0"><intensity value="{30 * n_side * n_side}"/><position x="{0.3 * n_side}" y="{-0.4 * n_side}" z="{n_side}"/><size value="1"/></light>']
    cam = (f'<camera><position x="{0.2 * n_side}" y="{-1.1 * n_side}" z="{0.7 * n_side}"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/>'
           '<fov value="40"/><width value="288"/><height value="176"/></camera>')
    (pathlib.Path(tmp_path) / "grid.xml").write_text("<xml><scene>" + "".join(body + mats + lights) + "</scene>" + cam + "</xml>")
    return str(pathlib.Path(tmp_path) / "grid.xml")
