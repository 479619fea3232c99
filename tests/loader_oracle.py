"""Independent restatement of the reference's scene front end, in Python, for checking the product's loaders.

TEST INFRASTRUCTURE ONLY: nothing under cudaraytracing_amd/ imports this file.  It shares no code with the product:
XML comes from xml.etree, PNG from PIL, numbers from the C library's strtof, and all arithmetic runs in numpy float32
scalars in the order the reference's host code uses, so that every float can be compared bit for bit.

What it follows (reference file:line):
  * Renderer::LoadScene, Scene::Load, CountNodes, AssignNodes      lib/xmlload.cu:106-303
  * Camera::Load, RenderInfo::Load                                 lib/xmlload.cu:307-354
  * LoadLight, *Light::Load, LoadMaterial                          lib/xmlload.cu:358-406
  * Loader::ReadFloat3 / ReadColor / ReadTexture / LoadTexture     lib/xmlload.cuh:59-127
  * Matrix (product, inverse, Translation / Rotation / Scale)      math/matrix.cuh:25-56, 111-172
  * asNorm, cross, length                                          math/float3.cuh:144-195
  * Color24, Texture constructors                                  math/color.cuh:29-43, scene/texture.cuh:12-16, scene/texture.cu:5-8
  * Node::FromLocal(Box), Node::CalculateBoundingBox               scene/scene.cuh:191-251
  * Mesh::LoadFromFileObj, Mesh::ComputeBoundingBox                scene/mesh.cu:7-200
Where the reference leaves a value unset (missing vertex components, a camera without <dof>) this file uses 0, as the
product documents it does.
"""
import ctypes
import ctypes.util
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

F = np.float32
BIG = F(3.402823466e+38)
DEG2RAD = F(0.017453292519943296)

_libc = ctypes.CDLL(ctypes.util.find_library("c"))
_libm = ctypes.CDLL(ctypes.util.find_library("m"))
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]
_libc.strtol.restype = ctypes.c_long
_libc.strtol.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
for _n in ("sinf", "cosf", "tanf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]


def strtof(text):
    """(value, characters consumed) of the C library's float parser; (0, 0) when nothing parses."""
    raw = text if isinstance(text, bytes) else text.encode()
    buf = ctypes.create_string_buffer(raw)
    end = ctypes.c_char_p()
    v = _libc.strtof(buf, ctypes.byref(end))
    used = ctypes.cast(end, ctypes.c_void_p).value - ctypes.addressof(buf)
    return F(v), used


def strtol(text):
    raw = text if isinstance(text, bytes) else text.encode()
    buf = ctypes.create_string_buffer(raw)
    end = ctypes.c_char_p()
    v = _libc.strtol(buf, ctypes.byref(end), 10)
    used = ctypes.cast(end, ctypes.c_void_p).value - ctypes.addressof(buf)
    return int(v), used


def sinf(x): return F(_libm.sinf(float(x)))
def cosf(x): return F(_libm.cosf(float(x)))
def tanf(x): return F(_libm.tanf(float(x)))


# ---------------------------------------------------------------- float3 (math/float3.cuh)
def v3(x, y, z): return [F(x), F(y), F(z)]
def add(a, b): return [a[0] + b[0], a[1] + b[1], a[2] + b[2]]
def sub(a, b): return [a[0] - b[0], a[1] - b[1], a[2] - b[2]]
def scale(a, s): return [a[0] * s, a[1] * s, a[2] * s]
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def length(a): return np.sqrt(dot(a, a))


def as_norm(a):
    s = F(1) / np.sqrt(dot(a, a))
    return scale(a, s)


def cross(a, b):
    """float3.cuh:154-156: three scaled basis vectors, subtracted and added in that order."""
    r = a[1] * b[2] - b[1] * a[2]
    u = a[0] * b[2] - b[0] * a[2]
    f = a[0] * b[1] - b[0] * a[1]
    right, up, fwd = v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)
    return add(sub(scale(right, r), scale(up, u)), scale(fwd, f))


# ---------------------------------------------------------------- Matrix (math/matrix.cuh), 12 cells, column major
def m_identity():
    m = [F(0)] * 12
    m[0] = m[4] = m[8] = F(1)
    return m


def m_mul(l, r):
    p = [F(0)] * 12
    for col in range(3):
        for row in range(3):
            p[3 * col + row] = l[row] * r[3 * col] + l[3 + row] * r[3 * col + 1] + l[6 + row] * r[3 * col + 2]
    for row in range(3):
        p[9 + row] = l[row] * r[9] + l[3 + row] * r[10] + l[6 + row] * r[11] + l[9 + row]
    return p


def m_inverse(c):
    o = [F(0)] * 12
    d_a = c[8] * c[10] - c[7] * c[11]
    d_b = c[6] * c[11] - c[8] * c[9]
    d_c = c[7] * c[9] - c[6] * c[10]
    o[0] = c[4] * c[8] - c[5] * c[7]
    o[1] = c[2] * c[7] - c[1] * c[8]
    o[2] = c[1] * c[5] - c[2] * c[4]
    o[3] = c[5] * c[6] - c[3] * c[8]
    o[4] = c[0] * c[8] - c[2] * c[6]
    o[5] = c[2] * c[3] - c[0] * c[5]
    o[6] = c[3] * c[7] - c[4] * c[6]
    o[7] = c[1] * c[6] - c[0] * c[7]
    o[8] = c[0] * c[4] - c[1] * c[3]
    o[9] = c[3] * d_a + c[4] * d_b + c[5] * d_c
    o[10] = c[0] * -d_a + c[1] * -d_b + c[2] * -d_c
    o[11] = (c[0] * (c[5] * c[10] - c[4] * c[11]) + c[1] * (c[3] * c[11] - c[5] * c[9])
             + c[2] * (c[4] * c[9] - c[3] * c[10]))
    det = c[0] * o[0] + c[1] * o[3] + c[2] * o[6]
    s = F(1) / det
    return [x * s for x in o]


def m_rotation(axis, degrees):
    m = m_identity()
    rad = DEG2RAD * degrees
    sina, cosa = sinf(rad), cosf(rad)
    t = F(1) - cosa
    a = scale(axis, t)
    txy, txz, tyz = a[0] * axis[1], a[0] * axis[2], a[1] * axis[2]
    s = scale(axis, sina)
    m[0] = a[0] * axis[0] + cosa; m[1] = txy + s[2]; m[2] = txz - s[1]
    m[3] = txy - s[2]; m[4] = a[1] * axis[1] + cosa; m[5] = tyz + s[0]
    m[6] = txz + s[1]; m[7] = tyz - s[0]; m[8] = a[2] * axis[2] + cosa
    return m


def m_point(c, v):
    return [c[0] * v[0] + c[3] * v[1] + c[6] * v[2] + c[9],
            c[1] * v[0] + c[4] * v[1] + c[7] * v[2] + c[10],
            c[2] * v[0] + c[5] * v[1] + c[8] * v[2] + c[11]]


# ---------------------------------------------------------------- Box (scene/rays.cuh:108-135)
class Box:
    def __init__(self):
        self.lo, self.hi = [BIG, BIG, BIG], [-BIG, -BIG, -BIG]

    def add_point(self, p):
        for k in range(3):
            if self.lo[k] > p[k]: self.lo[k] = p[k]
            if self.hi[k] < p[k]: self.hi[k] = p[k]

    def add_box(self, b):
        self.add_point(b.lo); self.add_point(b.hi)


# ---------------------------------------------------------------- XML access (lib/xmlload.cuh:32-145)
def tag_is(elem, name):
    return elem.tag.lower() == name.lower()


def child(elem, name):
    """Loader::Child: first child element of that exact name (tinyxml2 compares case-sensitively), or None."""
    if elem is None:
        return None
    for c in elem:
        if c.tag == name:
            return c
    return None


def read_float(elem, name="value"):
    """QueryFloatAttribute: None when the attribute is absent or does not start with a number."""
    if elem is None or name not in elem.attrib:
        return None
    v, used = strtof(elem.attrib[name])
    return v if used else None


def read_int(elem, name="value"):
    if elem is None or name not in elem.attrib:
        return None
    v, used = strtol(elem.attrib[name])
    return v if used else None


def read_triple(elem, names, default):
    """ReadFloat3 / ReadColor: None when the element is missing (the caller's value stays as it was)."""
    if elem is None:
        return None
    v = list(default)
    for k, n in enumerate(names):
        x = read_float(elem, n)
        if x is not None:
            v[k] = x
    f = read_float(elem)
    if f is not None:
        v = scale(v, f)
    return v


def read_float3(elem, default=(0, 0, 0)): return read_triple(elem, "xyz", v3(*default))
def read_color(elem, default=(1, 1, 1)): return read_triple(elem, "rgb", v3(*default))


def clamp_to_byte(f):
    """Color24::ClampToInt on the host: (int) of an out-of-range or NaN float is INT_MIN on x86."""
    f = F(f)
    v = int(f) if (f > F(-2147483648.0) and f < F(2147483648.0)) else -2 ** 31
    return 0 if v < 0 else (255 if v > 255 else v)


def color24(c):
    return [clamp_to_byte(c[0] * F(256)), clamp_to_byte(c[1] * F(256)), clamp_to_byte(c[2] * F(256))]


def read_texture(elem, default, asset_root):
    """Loader::ReadTexture: (width, height, uint8 array of shape (h*w, 3))."""
    if elem is None:
        return 1, 1, np.array([color24(v3(*default))], np.uint8)
    data = None
    name = elem.attrib.get("texture")
    if name and len(name) >= 3 and name[-3:].lower() == "png":
        from PIL import Image
        try:
            with Image.open(os.path.join(asset_root, name)) as im:
                rgb = np.asarray(im.convert("RGB"), np.uint8)
            h, w = rgb.shape[:2]
            data = rgb.reshape(-1, 3).astype(np.float32) * F(1.0 / 256.0)       # Color24::ToColor
        except OSError:
            data = None
    if data is None:
        w = h = 1
        data = np.ones((1, 3), np.float32)
    mult = v3(1, 1, 1)
    for k, n in enumerate("rgb"):
        x = read_float(elem, n)
        if x is not None:
            mult[k] = x
    f = read_float(elem)
    if f is not None:
        mult = scale(mult, f)
    data = data * np.array(mult, np.float32)                                    # float32 products, one rounding each
    q = data * F(256)
    ok = (q > F(-2147483648.0)) & (q < F(2147483648.0))
    iv = np.where(ok, np.trunc(np.where(ok, q, 0)), -2.0 ** 31)
    return w, h, np.clip(iv, 0, 255).astype(np.uint8)


# ---------------------------------------------------------------- OBJ (scene/mesh.cu:14-200)
_SKIP = re.compile(rb"(?:[ \t\n\v\f\r]+|#[^\n\r\0]*)*")
_LINE = re.compile(rb"[^\n\r\0]*")
_BLANK = b" \t\v\f"


def _records(data):
    """Buffer::ReadLine over the whole file: yields the records the reference's 1024-byte line buffer would hold."""
    p, n = 0, len(data)
    while p < n:
        p = _SKIP.match(data, p).end()
        if p >= n or data[p] == 0:
            return
        raw = _LINE.match(data, p).group()
        rec = b" ".join(raw.split())
        if len(rec) < 1023:
            yield rec
            p += len(raw) + 1
            continue
        # a record that fills the buffer: character by character, as the reference reads it
        out = bytearray()
        gap = False
        while len(out) < 1023:
            if p >= n or data[p] in b"\n\r\0":
                break
            c = data[p]
            if c in _BLANK:
                gap = True
            else:
                if gap:
                    out.append(32)
                gap = False
                out.append(c)
            p += 1
        # (the character after a cut has been taken from the stream already; so has a line end)
        p += 1
        yield bytes(out[:1023]) if len(out) <= 1023 else bytes(out)


def _is_command(rec, cmd):
    return rec.startswith(cmd) and (len(rec) == len(cmd) or rec[len(cmd):len(cmd) + 1] == b" ")


def _scan_floats(text):
    """sscanf("%f %f %f"): stops at the first field that is not a number; missing fields are 0 here."""
    out = [F(0), F(0), F(0)]
    for k in range(3):
        text = text.lstrip(b" ")
        v, used = strtof(text)
        if not used:
            break
        out[k] = v
        text = text[used:]
    return out


def load_obj(path):
    """Returns dict(v, vt, vn: float32 (n,3); f, ft, fn: uint32 (m,3) or None; box_min, box_max)."""
    with open(path, "rb") as fp:
        data = fp.read()
    v, vt, vn, f, ft, fn = [], [], [], [], [], []
    has_n = has_t = False
    M32 = 0xFFFFFFFF
    for rec in _records(data):
        if _is_command(rec, b"v"):
            v.append(_scan_floats(rec[2:]))
        elif _is_command(rec, b"vt"):
            vt.append(_scan_floats(rec[2:])); has_t = True
        elif _is_command(rec, b"vn"):
            vn.append(_scan_floats(rec[2:])); has_n = True
        elif _is_command(rec, b"f"):
            facevert, inspace, negative, kind, index = -1, True, False, 0, 0
            face, tface, nface = [0, 0, 0], [0, 0, 0], [0, 0, 0]
            for i in range(2, len(rec)):
                ch = rec[i]
                if ch == 32:
                    inspace = True
                    continue
                if inspace:
                    inspace, negative, kind, index = False, False, 0, 0
                    if facevert < 2:
                        if facevert == -1:
                            face, tface, nface = [0, 0, 0], [0, 0, 0], [0, 0, 0]
                        facevert += 1
                    else:
                        f.append(list(face)); face[1] = face[2]
                        if has_t: ft.append(list(tface)); tface[1] = tface[2]
                        if has_n: fn.append(list(nface)); nface[1] = nface[2]
                if ch == 47:
                    kind += 1; index = 0
                if ch == 45:
                    negative = True
                if 48 <= ch <= 57:
                    index = (index * 10 + (ch - 48)) & M32
                    if kind == 0:
                        face[facevert] = ((len(v) - index) if negative else (index - 1)) & M32
                    elif kind == 1:
                        tface[facevert] = ((len(vt) - index) if negative else (index - 1)) & M32; has_t = True
                    elif kind == 2:
                        nface[facevert] = ((len(vn) - index) if negative else (index - 1)) & M32; has_n = True
            f.append(list(face))
            if has_t: ft.append(list(tface))
            if has_n: fn.append(list(nface))
    va = np.array(v, np.float32).reshape(-1, 3)
    box = Box()
    if len(va):
        box.lo = list(va.min(axis=0)); box.hi = list(va.max(axis=0))
    arr = lambda x, t: np.array(x, t).reshape(-1, 3)
    return {"v": va, "vt": arr(vt, np.float32), "vn": arr(vn, np.float32), "f": arr(f, np.uint32),
            "ft": arr(ft, np.uint32) if ft else None, "fn": arr(fn, np.uint32) if fn else None,
            "box_min": [F(x) for x in box.lo], "box_max": [F(x) for x in box.hi]}


# ---------------------------------------------------------------- the scene (lib/xmlload.cu)
OBJ_NONE, OBJ_SPHERE, OBJ_PLANE, OBJ_MESH = 0, 1, 2, 3
LIGHT_AMBIENT, LIGHT_DIRECT, LIGHT_POINT = 0, 1, 2


class Node:
    def __init__(self):
        self.child_count, self.kind, self.mesh, self.material = 0, OBJ_NONE, None, 0
        self.tm, self.itm, self.box = m_identity(), m_identity(), Box()


def load_scene(xml_path, asset_root):
    root = ET.parse(xml_path).getroot()
    if root.tag != "xml":
        raise ValueError("no xml tag")
    xscene, xcam = child(root, "scene"), child(root, "camera")
    if xscene is None or xcam is None:
        raise ValueError("no scene / camera tag")
    sc = {"meshes": {}, "mesh_order": []}

    # materials and lights first (Scene::Load, xmlload.cu:168-194); slot 0 is the default material
    def material(e):
        m = {"glossiness": F(20), "ior": F(1), "absorption": v3(0, 0, 0)}
        m["diffuse"] = read_texture(child(e, "diffuse"), (1, 1, 1), asset_root)
        m["specular"] = read_texture(child(e, "specular"), (1, 1, 1), asset_root)
        g = read_float(child(e, "glossiness"))
        if g is not None: m["glossiness"] = g
        m["refraction"] = read_texture(child(e, "refraction"), (0, 0, 0), asset_root)
        ior = read_float(child(e, "refraction"), "index")
        if ior is not None: m["ior"] = ior
        m["emission"] = read_texture(child(e, "emission"), (0, 0, 0), asset_root)
        ab = read_color(child(e, "absorption"))
        if ab is not None: m["absorption"] = ab
        return m

    materials = [material(None)]
    names = {"__DEFAULT_MTL__": 0}
    lights = []
    for e in xscene:
        if tag_is(e, "material"):
            materials.append(material(e))
            names.setdefault(e.attrib.get("name", ""), len(materials) - 1)      # the probe finds the first insertion
        elif tag_is(e, "light"):
            ty = (e.attrib.get("type") or "").lower()
            li = {"kind": None, "intensity": v3(0, 0, 0), "vec": v3(0, 0, 0), "size": F(1)}
            c = read_color(child(e, "intensity"))
            if ty == "ambient":
                li["kind"] = LIGHT_AMBIENT
            elif ty == "direct":
                li["kind"] = LIGHT_DIRECT
                d = read_float3(child(e, "direction"))
                d = v3(0, 1, 0) if d is None else d
                s = F(1) / np.sqrt(dot(d, d))
                li["vec"] = scale(d, s)
            elif ty == "point":
                li["kind"] = LIGHT_POINT
                p = read_float3(child(e, "position"))
                if p is not None: li["vec"] = p
                s = read_float(child(e, "size"))
                if s is not None: li["size"] = s
            if c is not None and li["kind"] is not None:
                li["intensity"] = c
            lights.append(li)

    nodes = [Node()]

    def assign(e, parent):
        node = Node()
        nodes.append(node)
        ty = e.attrib.get("type")
        if ty is not None:
            t = ty.lower()
            if t == "sphere": node.kind = OBJ_SPHERE
            elif t == "plane": node.kind = OBJ_PLANE
            elif t == "obj":
                name = e.attrib.get("name")
                if name not in sc["meshes"]:
                    path = os.path.join(asset_root, name or "")
                    if os.path.isfile(path):
                        sc["meshes"][name] = load_obj(path)
                        sc["mesh_order"].append(name)
                if name in sc["meshes"]:
                    node.kind, node.mesh = OBJ_MESH, name
        mat = e.attrib.get("material")
        if mat is not None:
            node.material = names[mat]
        translation, rotation, scl = m_identity(), m_identity(), m_identity()
        for c in e:
            if tag_is(c, "scale"):
                s = read_float3(c, (1, 1, 1))
                scl = m_identity(); scl[0], scl[4], scl[8] = s
            elif tag_is(c, "rotate"):
                s = read_float3(c)
                a = read_float(c, "angle")
                rotation = m_rotation(as_norm(s), F(0) if a is None else a)
            elif tag_is(c, "translate"):
                t = read_float3(c)
                translation = m_identity(); translation[9], translation[10], translation[11] = t
        node.tm = m_mul(m_mul(m_mul(parent, translation), rotation), scl)
        node.itm = m_inverse(node.tm)
        for c in e:
            if tag_is(c, "object"):
                node.child_count += 1
                assign(c, node.tm)

    for e in xscene:
        if tag_is(e, "object"):
            nodes[0].child_count += 1
            assign(e, m_identity())

    # boxes, last node first (xmlload.cu:211-213, scene.cuh:191-251)
    for i in range(len(nodes) - 1, -1, -1):
        n = nodes[i]
        n.box = Box()
        if n.kind != OBJ_NONE:
            if n.kind == OBJ_SPHERE: lo, hi = v3(-1, -1, -1), v3(1, 1, 1)
            elif n.kind == OBJ_PLANE: lo, hi = v3(-1, -1, 0), v3(1, 1, 0)
            else: lo, hi = sc["meshes"][n.mesh]["box_min"], sc["meshes"][n.mesh]["box_max"]
            t = Box()
            for x in (lo[0], hi[0]):
                for y in (lo[1], hi[1]):
                    for z in (lo[2], hi[2]):
                        t.add_point(m_point(n.tm, [x, y, z]))
            n.box.add_box(t)
        children, skip, j = n.child_count, 0, i + 1
        while children > 0:
            if skip > 0:
                skip -= 1; skip += nodes[j].child_count
            else:
                n.box.add_box(nodes[j].box); skip = nodes[j].child_count; children -= 1
            j += 1

    env = read_texture(child(xscene, "environment"), (0, 0, 0), asset_root)

    # camera (Camera::Load, RenderInfo::Load)
    cam = {"dof": F(0), "fov": F(0)}
    pos = read_float3(child(xcam, "position")) or v3(0, 0, 0)
    target = read_float3(child(xcam, "target")) or v3(0, 0, 0)
    up = read_float3(child(xcam, "up")) or v3(0, 0, 0)
    for k in ("fov", "dof"):
        x = read_float(child(xcam, k))
        if x is not None: cam[k] = x
    cam["srgb"] = 1 if xcam.attrib.get("gamma") == "sRGB" else 0
    w, h = read_int(child(xcam, "width")), read_int(child(xcam, "height"))
    cam["width"], cam["height"] = w, h
    aspect = F(w) / F(h)
    fov_rad = cam["fov"] * DEG2RAD
    focal = length(sub(target, pos))
    cZ = as_norm(sub(target, pos))
    cX = cross(cZ, as_norm(up))
    cY = cross(cX, cZ)
    plane_h = F(2) * focal * tanf(fov_rad * F(0.5))
    plane_w = plane_h * aspect
    pixel = plane_h / F(h)
    centre = add(pos, scale(cZ, focal))
    corner = add(sub(centre, scale(cX, plane_w * F(0.5))), scale(cY, plane_h * F(0.5)))
    tl = add(corner, scale(sub(cX, cY), pixel * F(0.5)))
    cam.update(position=pos, target=target, up=up, cX=cX, cY=cY, cZ=cZ, pixel_size=pixel, top_left_pixel=tl)

    sc.update(nodes=nodes, materials=materials, lights=lights, env=env, camera=cam)
    return sc


# ---------------------------------------------------------------- comparison with the product's flat scene
def _bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def compare(scene, ref):
    """Differences between a cudaraytracing_amd.Scene and load_scene()'s result, as a list of strings (empty = equal).
    Floats are compared by bit pattern."""
    out = []
    d = scene.desc()

    def same(label, got, want):
        g, w = _bits(list(got)), _bits([F(x) for x in want])
        if g.shape != w.shape or not (g == w).all():
            out.append(f"{label}: {np.asarray(list(got), np.float32)} != {np.asarray(want, np.float32)}")

    def same_texture(label, index, want):
        t = d.textures[index]
        w, h, px = want
        if (t.width, t.height) != (w, h):
            out.append(f"{label}: size {t.width}x{t.height} != {w}x{h}"); return
        got = np.ctypeslib.as_array(t.rgb, shape=(w * h, 3))
        if not (got == px).all():
            out.append(f"{label}: {int((got != px).any(axis=1).sum())} of {w * h} texels differ")

    if d.node_count != len(ref["nodes"]):
        out.append(f"node count {d.node_count} != {len(ref['nodes'])}"); return out
    mesh_names = {}
    for i, n in enumerate(ref["nodes"]):
        p = d.nodes[i]
        if (p.child_count, p.object_kind, p.material_id) != (n.child_count, n.kind, n.material):
            out.append(f"node {i}: children / kind / material {(p.child_count, p.object_kind, p.material_id)} != {(n.child_count, n.kind, n.material)}")
        same(f"node {i} tm", p.tm, n.tm); same(f"node {i} itm", p.itm, n.itm)
        same(f"node {i} box_min", p.box_min, n.box.lo); same(f"node {i} box_max", p.box_max, n.box.hi)
        if n.kind == OBJ_MESH:
            if mesh_names.setdefault(p.mesh_id, n.mesh) != n.mesh:
                out.append(f"node {i}: mesh id {p.mesh_id} stands for two files")
    for mid, name in mesh_names.items():
        pm, rm = d.meshes[mid], ref["meshes"][name]
        for key, cnt, ptr, width in (("v", pm.nv, pm.v, 3), ("vn", pm.nvn, pm.vn, 3), ("vt", pm.nvt, pm.vt, 3)):
            want = rm[key]
            got = np.ctypeslib.as_array(ptr, shape=(cnt, 3)) if cnt else np.zeros((0, 3), np.float32)
            if got.shape != want.shape or not (_bits(got) == _bits(want)).all():
                out.append(f"mesh {name}: {key} differs ({got.shape} vs {want.shape})")
        for key, ptr in (("f", pm.f), ("fn", pm.fn), ("ft", pm.ft)):
            want = rm[key]
            if want is None or (key != "f" and len(rm["vn" if key == "fn" else "vt"]) == 0):
                continue                                  # index triples of a list the file never filled are not used
            got = np.ctypeslib.as_array(ptr, shape=(pm.nf, 3)) if ptr else None
            if got is None or got.shape != want.shape or not (got == want).all():
                out.append(f"mesh {name}: {key} differs")
        same(f"mesh {name} box_min", pm.box_min, rm["box_min"]); same(f"mesh {name} box_max", pm.box_max, rm["box_max"])
    if d.material_count != len(ref["materials"]):
        out.append(f"material count {d.material_count} != {len(ref['materials'])}")
    else:
        for i, m in enumerate(ref["materials"]):
            p = d.materials[i]
            same(f"material {i} glossiness / ior", [p.glossiness, p.ior], [m["glossiness"], m["ior"]])
            same(f"material {i} absorption", p.absorption, m["absorption"])
            for key in ("diffuse", "specular", "refraction", "emission"):
                same_texture(f"material {i} {key}", getattr(p, key), m[key])
    same_texture("environment", d.env_texture, ref["env"])
    if d.light_count != len(ref["lights"]):
        out.append(f"light count {d.light_count} != {len(ref['lights'])}")
    else:
        for i, li in enumerate(ref["lights"]):
            p = d.lights[i]
            if li["kind"] is None:
                continue                                  # unknown type: the reference leaves the slot unset
            if p.kind != li["kind"]:
                out.append(f"light {i}: kind {p.kind} != {li['kind']}")
            same(f"light {i} intensity", p.intensity, li["intensity"])
            if li["kind"] != LIGHT_AMBIENT: same(f"light {i} vector", p.vec, li["vec"])
            if li["kind"] == LIGHT_POINT: same(f"light {i} size", [p.size], [li["size"]])
    c, r = d.camera, ref["camera"]
    if (c.width, c.height, c.srgb) != (r["width"], r["height"], r["srgb"]):
        out.append(f"camera size / gamma {(c.width, c.height, c.srgb)} != {(r['width'], r['height'], r['srgb'])}")
    for key in ("position", "cX", "cY", "cZ", "top_left_pixel", "target"):
        same(f"camera {key}", getattr(c, key), r[key])
    same("camera dof / pixel size / fov", [c.dof, c.pixel_size, c.fov], [r["dof"], r["pixel_size"], r["fov"]])
    return out
