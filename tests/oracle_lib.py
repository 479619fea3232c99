"""ctypes wrapper of the CPU oracle (oracle/liboracle_{det,glibc}.so).  Test
infrastructure only: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(), never by the product package."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTER_NAMES = ("closest_rays", "shadow_rays", "paths", "node_box_tests", "node_enters", "bvh_interior",
                 "bvh_leaves", "tri_tests", "tri_hits", "sphere_tests", "plane_tests", "texel_quads")

_libs = {}


def load(flavour="det"):
    if flavour not in _libs:
        path = os.path.join(ROOT, "oracle", f"liboracle_{flavour}.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = C.CDLL(path)
        L.oracle_scene_load.restype = C.c_void_p
        L.oracle_scene_load.argtypes = [C.c_char_p]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_set_resolution.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.oracle_get_camera.argtypes = [C.c_void_p, C.c_void_p]
        for n in ("oracle_bvh_node_count", "oracle_bvh_depth", "oracle_mesh_face_count"):
            getattr(L, n).restype = C.c_uint32
            getattr(L, n).argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_bvh_arrays.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 8
        L.oracle_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                    C.c_uint32, C.c_int, C.c_void_p]
        L.oracle_primary_hits.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_count_center_hits.restype = C.c_uint64
        L.oracle_count_center_hits.argtypes = [C.c_void_p]
        L.oracle_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_float]
        L.oracle_xorwow.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_seqjump_matrix.argtypes = [C.c_uint32, C.c_void_p]
        L.oracle_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_libm_is_glibc.restype = C.c_int
        L.oracle_set_node_cull.argtypes = [C.c_int]
        _libs[flavour] = L
    return _libs[flavour]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, path, flavour="det"):
        self.L = load(flavour)
        self.h = self.L.oracle_scene_load(str(path).encode())
        if not self.h:
            raise RuntimeError(f"oracle cannot load {path}")
        self._refresh()

    def _refresh(self):
        info = np.zeros(7, np.uint32)
        self.L.oracle_scene_info(self.h, _p(info))
        self.width, self.height, self.nodes, self.lights, self.materials, self.meshes, self.srgb = [int(x) for x in info]

    def set_resolution(self, w, h):
        self.L.oracle_set_resolution(self.h, w, h)
        self._refresh()

    def camera(self):
        out = np.zeros(22, np.float32)
        self.L.oracle_get_camera(self.h, _p(out))
        return out

    def bvh(self, mesh):
        n = self.L.oracle_bvh_node_count(self.h, mesh)
        nf = self.L.oracle_mesh_face_count(self.h, mesh)
        r = {"node_count": n, "depth": self.L.oracle_bvh_depth(self.h, mesh), "leaf": np.zeros(n, np.uint8)}
        for k in ("start", "count", "left", "right", "next"):
            r[k] = np.zeros(n, np.uint32)
        r["box"] = np.zeros((n, 6), np.float32)
        r["elements"] = np.zeros(nf, np.uint32)
        self.L.oracle_bvh_arrays(self.h, mesh, _p(r["leaf"]), _p(r["start"]), _p(r["count"]), _p(r["left"]),
                                 _p(r["right"]), _p(r["next"]), _p(r["box"]), _p(r["elements"]))
        return r

    def new_accumulators(self):
        return [np.zeros((self.height, self.width, 3), np.float32) for _ in range(3)]

    def render(self, pass_id, samples, acc=None, rows=None, threads=None, count=False):
        """One pass; returns (results, normals, albedos[, counters])."""
        acc = acc or self.new_accumulators()
        y0, y1 = rows or (0, self.height)
        threads = threads or min(8, os.cpu_count() or 1)
        cnt = np.zeros(len(COUNTER_NAMES), np.uint64) if count else None
        self.L.oracle_render(self.h, pass_id, samples, _p(acc[0]), _p(acc[1]), _p(acc[2]), y0, y1, threads,
                             _p(cnt) if count else None)
        if count:
            return acc, dict(zip(COUNTER_NAMES, [int(x) for x in cnt]))
        return acc

    def primary_hits(self, pass_id=0, rows=None):
        y0, y1 = rows or (0, self.height)
        node = np.full((self.height, self.width), -1, np.int32)
        face = np.full((self.height, self.width), -1, np.int32)
        t = np.zeros((self.height, self.width), np.float32)
        self.L.oracle_primary_hits(self.h, pass_id, y0, y1, _p(node), _p(face), _p(t))
        return node, face, t

    def count_center_hits(self):
        return int(self.L.oracle_count_center_hits(self.h))

    def close(self):
        if self.h:
            self.L.oracle_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_node_cull(by_entry, flavour="det"):
    """by_entry: the node scan culls by where a ray enters a node's box - not the reference's rule (see oracle.cpp);
    only for telling the declared departure of the product's opt-in BVH over the nodes from a defect."""
    load(flavour).oracle_set_node_cull(1 if by_entry else 0)


def convert(results, srgb, pass_mul, flavour="det"):
    L = load(flavour)
    r = np.ascontiguousarray(results, np.float32)
    out = np.zeros(r.shape, np.uint8)
    L.oracle_convert(_p(r), _p(out), r.size // 3, int(srgb), C.c_float(pass_mul))
    return out


def xorwow(seeds, subsequences, draws, flavour="det"):
    L = load(flavour)
    s = np.ascontiguousarray(seeds, np.uint32)
    q = np.ascontiguousarray(subsequences, np.uint32)
    out = np.zeros((s.size, 6 + draws), np.uint32)
    L.oracle_xorwow(_p(s), _p(q), s.size, draws, _p(out))
    return out


def seqjump_matrix(k=0, flavour="det"):
    out = np.zeros((160, 5), np.uint32)
    load(flavour).oracle_seqjump_matrix(k, _p(out))
    return out


def math(fn, x, y=None, flavour="det"):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else x, np.float32)
    out = np.zeros_like(x)
    load(flavour).oracle_math(fn, _p(x), _p(y), x.size, _p(out))
    return out
