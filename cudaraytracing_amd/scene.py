"""Host scene model: thin object over the crt_scene handle of the C ABI."""
import ctypes as C

import numpy as np

from . import _capi as capi


class Scene:
    """Owns a crt_scene (flat scene description).  Mirrors what Renderer::LoadScene
    (lib/xmlload.cu:106-149) leaves in the reference's global `theScene`."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def load_xml(cls, xml_path, asset_root=None):
        h = C.c_void_p()
        capi.check_host(capi.lib().crt_scene_load_xml(str(xml_path).encode(), asset_root.encode() if asset_root else None, C.byref(h)))
        return cls(h.value)

    @classmethod
    def load_file(cls, path):
        h = C.c_void_p()
        capi.check_host(capi.lib().crt_scene_load_file(str(path).encode(), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, path, asset_root=None):
        return cls.load_file(path) if str(path).endswith(".crtscene") else cls.load_xml(path, asset_root)

    def make_instanced(self, mesh_index, copies, seed=1, width=3840, height=2160):
        """Synthetic stress scene: `copies` jittered copies of one mesh of this scene flattened into one mesh."""
        h = C.c_void_p()
        capi.check_host(capi.lib().crt_scene_make_instanced(self._h, int(mesh_index), int(copies), int(seed),
                                                            int(width), int(height), C.byref(h)))
        return Scene(h.value)

    def save_file(self, path):
        capi.check_host(capi.lib().crt_scene_save_file(self._h, str(path).encode()))

    def set_resolution(self, width, height):
        capi.check_host(capi.lib().crt_scene_set_resolution(self._h, int(width), int(height)))

    def desc(self):
        d = capi.SceneDesc()
        capi.check_host(capi.lib().crt_scene_get_desc(self._h, C.byref(d)))
        return d

    @property
    def width(self):
        return int(self.desc().camera.width)

    @property
    def height(self):
        return int(self.desc().camera.height)

    @property
    def srgb(self):
        return int(self.desc().camera.srgb)

    def mesh_arrays(self, i):
        m = self.desc().meshes[i]
        v = np.ctypeslib.as_array(m.v, shape=(m.nv * 3,)).copy() if m.nv else np.zeros(0, np.float32)
        f = np.ctypeslib.as_array(m.f, shape=(m.nf * 3,)).copy() if m.nf else np.zeros(0, np.uint32)
        return v, f, np.array(m.box_min[:], np.float32), np.array(m.box_max[:], np.float32)

    def close(self):
        if self._h:
            capi.lib().crt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_bvh(v, f, box_min, box_max):
    """Run the deterministic builder (replacement of BVH::ConstructBVH, renderer/bvh.cu:7-35) and
    return its arrays as a dict of numpy arrays."""
    L = capi.lib()
    v = np.ascontiguousarray(v, np.float32)
    f = np.ascontiguousarray(f, np.uint32)
    bmin = np.ascontiguousarray(box_min, np.float32)
    bmax = np.ascontiguousarray(box_max, np.float32)
    h = C.c_void_p()
    fp = C.POINTER(C.c_float)
    up = C.POINTER(C.c_uint32)
    capi.check_host(L.crt_bvh_build(v.ctypes.data_as(fp), v.size // 3, f.ctypes.data_as(up), f.size // 3,
                                    bmin.ctypes.data_as(fp), bmax.ctypes.data_as(fp), C.byref(h)))
    try:
        a = capi.BvhArrays()
        capi.check_host(L.crt_bvh_get_arrays(h, C.byref(a)))
        n, e = a.node_count, a.elem_count
        def arr(p, count, width=1):
            return np.ctypeslib.as_array(p, shape=(count * width,)).copy() if count else np.zeros(0)
        return {"node_count": int(n), "depth": int(a.depth), "leaf": arr(a.leaf, n), "start": arr(a.elem_start, n),
                "count": arr(a.elem_count_per_node, n), "left": arr(a.left, n), "right": arr(a.right, n),
                "next": arr(a.next, n), "box": arr(a.box, n, 6).reshape(-1, 6), "elements": arr(a.elements, e)}
    finally:
        L.crt_bvh_free(h)
