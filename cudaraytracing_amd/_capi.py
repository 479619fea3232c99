"""ctypes binding of libcrt_amd.so (C ABI declared in include/crt_amd.h).

The library is the product: if it is missing this module raises, there is no
fallback path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRT_AMD_LIBRARY") or os.path.join(_HERE, "libcrt_amd.so")   # override: kernel build variants under test

OK, E_INVALID, E_IO, E_HIP, E_NOSCENE, E_LIMIT = 0, -1, -2, -3, -4, -5
OBJ_NONE, OBJ_SPHERE, OBJ_PLANE, OBJ_MESH = 0, 1, 2, 3
LIGHT_AMBIENT, LIGHT_DIRECT, LIGHT_POINT = 0, 1, 2

F3 = C.c_float * 3


class Camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("position", F3), ("dof", C.c_float),
                ("srgb", C.c_int32), ("pixel_size", C.c_float), ("cX", F3), ("cY", F3), ("cZ", F3),
                ("top_left_pixel", F3), ("target", F3), ("up", F3), ("fov", C.c_float)]


class Node(C.Structure):
    _fields_ = [("child_count", C.c_uint32), ("object_kind", C.c_int32), ("mesh_id", C.c_int32),
                ("material_id", C.c_int32), ("box_min", F3), ("box_max", F3), ("tm", C.c_float * 12),
                ("itm", C.c_float * 12)]


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_int32), ("specular", C.c_int32), ("refraction", C.c_int32), ("emission", C.c_int32),
                ("glossiness", C.c_float), ("ior", C.c_float), ("absorption", F3)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("intensity", F3), ("vec", F3), ("size", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_uint8))]


class Mesh(C.Structure):
    _fields_ = [("nv", C.c_uint32), ("nf", C.c_uint32), ("nvn", C.c_uint32), ("nvt", C.c_uint32),
                ("v", C.POINTER(C.c_float)), ("f", C.POINTER(C.c_uint32)), ("vn", C.POINTER(C.c_float)),
                ("fn", C.POINTER(C.c_uint32)), ("vt", C.POINTER(C.c_float)), ("ft", C.POINTER(C.c_uint32)),
                ("box_min", F3), ("box_max", F3)]


class SceneDesc(C.Structure):
    _fields_ = [("camera", Camera), ("nodes", C.POINTER(Node)), ("node_count", C.c_uint32),
                ("materials", C.POINTER(Material)), ("material_count", C.c_uint32),
                ("lights", C.POINTER(Light)), ("light_count", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32),
                ("meshes", C.POINTER(Mesh)), ("mesh_count", C.c_uint32), ("env_texture", C.c_int32)]


class BvhArrays(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("elem_count", C.c_uint32), ("depth", C.c_uint32),
                ("leaf", C.POINTER(C.c_uint8)), ("elem_start", C.POINTER(C.c_uint32)),
                ("elem_count_per_node", C.POINTER(C.c_uint32)), ("left", C.POINTER(C.c_uint32)),
                ("right", C.POINTER(C.c_uint32)), ("next", C.POINTER(C.c_uint32)), ("box", C.POINTER(C.c_float)),
                ("elements", C.POINTER(C.c_uint32))]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("closest_rays", "shadow_rays", "paths", "node_box_tests", "node_enters",
                                           "bvh_interior", "bvh_leaves", "tri_tests", "tri_hits", "sphere_tests",
                                           "plane_tests", "texel_quads")] + [("sched_turns", C.c_uint64 * 12), ("sched_lanes", C.c_uint64 * 12), ("sched_cycles", C.c_uint64 * 12), ("path_cycles", C.c_uint64 * 8), ("scan_checks", C.c_uint64), ("scan_violations", C.c_uint64), ("errors", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_[:12]}

    def schedule(self):
        """Per operation (box, triangle, transform, path, fetch): wave turns and lanes served."""
        names = ("box", "triangle", "transform", "path", "exchange", "fetch")
        return {n: (int(self.sched_turns[i]), int(self.sched_lanes[i]), int(self.sched_cycles[i])) for i, n in enumerate(names)}


# every symbol include/crt_amd.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_PROTOS = {
    "crt_scene_load_xml": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_VP)]),
    "crt_scene_load_file": (C.c_int, [C.c_char_p, C.POINTER(_VP)]),
    "crt_scene_save_file": (C.c_int, [_VP, C.c_char_p]),
    "crt_scene_free": (None, [_VP]),
    "crt_scene_get_desc": (C.c_int, [_VP, C.POINTER(SceneDesc)]),
    "crt_scene_set_resolution": (C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    "crt_scene_make_instanced": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "crt_camera_derive": (C.c_int, [C.POINTER(Camera)]),
    "crt_host_last_error": (C.c_char_p, []),
    "crt_bvh_build": (C.c_int, [C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32,
                                C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_VP)]),
    "crt_bvh_get_arrays": (C.c_int, [_VP, C.POINTER(BvhArrays)]),
    "crt_bvh_free": (None, [_VP]),
    "crt_png_write_rgb8": (C.c_int, [C.c_char_p, _VP, C.c_uint32, C.c_uint32]),
    "crt_png_read_rgb8": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "crt_free": (None, [_VP]),
    "crt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "crt_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "crt_destroy": (None, [_VP]),
    "crt_set_tlas": (C.c_int, [_VP, C.c_int]),
    "crt_last_error": (C.c_char_p, [_VP]),
    "crt_upload_scene": (C.c_int, [_VP, C.POINTER(SceneDesc)]),
    "crt_set_partition": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32]),
    "crt_clear": (C.c_int, [_VP]),
    "crt_clear_async": (C.c_int, [_VP, _VP]),
    "crt_render_pass": (C.c_int, [_VP, C.c_int, C.c_int, _VP, C.c_int]),
    "crt_render_pass_noalloc": (C.c_int, [_VP, C.c_int, C.c_int, _VP, C.c_int]),
    "crt_convert": (C.c_int, [_VP, C.c_float, C.c_int, _VP, _VP, C.c_int]),
    "crt_prepare_denoise": (C.c_int, [_VP, C.c_float, _VP, _VP, _VP, _VP, C.c_int]),
    "crt_convert_floats": (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, C.c_int]),
    "crt_read_image": (C.c_int, [_VP, _VP]),
    "crt_read_accumulators": (C.c_int, [_VP, _VP, _VP, _VP]),
    "crt_device_buffers": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    "crt_primary_hits": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    "crt_get_stats": (C.c_int, [_VP, C.POINTER(Stats)]),
    "crt_render_pass_counted": (C.c_int, [_VP, C.c_int, C.c_int]),
    "crt_rng_selftest": (C.c_int, [_VP, _VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "crt_math_selftest": (C.c_int, [_VP, C.c_int, _VP, _VP, C.c_uint32, _VP]),
    "crt_knob_value": (C.c_uint32, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "crt_last_kernel_ms": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "crt_kernel_times": (C.c_int, [_VP, _VP, C.c_uint32, C.POINTER(C.c_uint32)]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


def lib():
    """Load libcrt_amd.so once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C cudaraytracing_amd/csrc` "
                               "(or __graft_entry__.build()); there is no fallback renderer")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class CrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"crt error {code}: {message}")
        self.code = code


def check_host(rc):
    if rc != OK:
        raise CrtError(rc, lib().crt_host_last_error().decode(errors="replace"))


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)
