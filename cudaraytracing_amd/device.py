"""Device context: one per GPU, wraps crt_ctx."""
import ctypes as C

import numpy as np

from . import _capi as capi


class Context:
    def __init__(self, device=0):
        h = C.c_void_p()
        rc = capi.lib().crt_create(int(device), C.byref(h))
        if rc != capi.OK:
            raise capi.CrtError(rc, capi.lib().crt_host_last_error().decode(errors="replace"))
        self._h = h
        self.width = self.height = 0
        self._scene = None

    def _check(self, rc):
        if rc != capi.OK:
            raise capi.CrtError(rc, capi.lib().crt_last_error(self._h).decode(errors="replace"))

    def set_tlas(self, enable=True):
        """From the next upload on, queries walk a BVH over the scene's object nodes instead of the reference's
        node scan (renderer/trace.cu:51, 166).  Faster on scenes of many nodes; images are no longer the
        reference's bit for bit."""
        self._check(capi.lib().crt_set_tlas(self._h, 1 if enable else 0))

    def upload(self, scene):
        d = scene.desc()
        self._check(capi.lib().crt_upload_scene(self._h, C.byref(d)))
        self.width, self.height = int(d.camera.width), int(d.camera.height)
        self._scene = scene   # keep the host arrays alive

    def set_partition(self, part, parts, strip_rows=16):
        self._check(capi.lib().crt_set_partition(self._h, part, parts, strip_rows))

    def clear(self, stream=None, wait=True):
        if wait and not stream:
            self._check(capi.lib().crt_clear(self._h))
        else:
            self._check(capi.lib().crt_clear_async(self._h, C.c_void_p(stream or 0)))

    def render_pass(self, pass_id, samples, stream=None, wait=True):
        self._check(capi.lib().crt_render_pass(self._h, pass_id, samples, C.c_void_p(stream or 0), 0 if wait else 1))

    def render_pass_noalloc(self, pass_id, samples, stream=None, wait=True):
        """render_pass on the library's static device workspace: makes no device allocation."""
        self._check(capi.lib().crt_render_pass_noalloc(self._h, pass_id, samples, C.c_void_p(stream or 0), 0 if wait else 1))

    def render_pass_counted(self, pass_id, samples):
        self._check(capi.lib().crt_render_pass_counted(self._h, pass_id, samples))

    def convert(self, pass_mul, srgb, out_device_ptr=None, stream=None, wait=True):
        self._check(capi.lib().crt_convert(self._h, float(pass_mul), int(srgb), C.c_void_p(out_device_ptr or 0),
                                           C.c_void_p(stream or 0), 0 if wait else 1))

    def prepare_denoise(self, pass_mul, beauty_ptr, normal_ptr, albedo_ptr, stream=None, wait=True):
        """Accumulators * pass_mul into three device buffers (width * height * 3 floats each) given by address."""
        self._check(capi.lib().crt_prepare_denoise(self._h, float(pass_mul), beauty_ptr, normal_ptr, albedo_ptr, stream, 0 if wait else 1))

    def convert_floats(self, rgb_in_ptr, srgb, out_device_ptr=None, stream=None, wait=True):
        """A float3 device image (e.g. a denoiser's output) to the 8-bit image."""
        self._check(capi.lib().crt_convert_floats(self._h, rgb_in_ptr, int(srgb), out_device_ptr, stream, 0 if wait else 1))

    def read_image(self):
        img = np.empty((self.height, self.width, 3), np.uint8)
        self._check(capi.lib().crt_read_image(self._h, capi.np_ptr(img)))
        return img

    def read_accumulators(self):
        out = [np.empty((self.height, self.width, 3), np.float32) for _ in range(3)]
        self._check(capi.lib().crt_read_accumulators(self._h, *[capi.np_ptr(a) for a in out]))
        return out

    def device_buffers(self):
        p = [C.c_void_p() for _ in range(4)]
        self._check(capi.lib().crt_device_buffers(self._h, *[C.byref(x) for x in p]))
        return [x.value for x in p]

    def primary_hits(self, pass_id=0):
        node = np.empty((self.height, self.width), np.int32)
        face = np.empty((self.height, self.width), np.int32)
        t = np.empty((self.height, self.width), np.float32)
        self._check(capi.lib().crt_primary_hits(self._h, pass_id, capi.np_ptr(node), capi.np_ptr(face), capi.np_ptr(t)))
        return node, face, t

    def stats(self):
        s = capi.Stats()
        self._check(capi.lib().crt_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def schedule_stats(self):
        s = capi.Stats()
        self._check(capi.lib().crt_get_stats(self._h, C.byref(s)))
        return s.schedule()

    def scan_check(self):
        """Counted pass: scene-node boxes found hit and how many of them the candidate mask would have dropped (must be 0)."""
        s = capi.Stats()
        self._check(capi.lib().crt_get_stats(self._h, C.byref(s)))
        return {"checks": int(s.scan_checks), "violations": int(s.scan_violations)}

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(capi.lib().crt_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def kernel_times(self):
        """Durations (ms) of the trace-kernel launches since the previous call."""
        buf = np.zeros(256, np.float32)
        n = C.c_uint32()
        self._check(capi.lib().crt_kernel_times(self._h, capi.np_ptr(buf), 256, C.byref(n)))
        return buf[:min(int(n.value), 256)].copy()

    def rng_selftest(self, seeds, subsequences, draws):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        subs = np.ascontiguousarray(subsequences, np.uint32)
        out = np.empty((seeds.size, 6 + draws), np.uint32)
        self._check(capi.lib().crt_rng_selftest(self._h, capi.np_ptr(seeds), capi.np_ptr(subs), seeds.size, draws, capi.np_ptr(out)))
        return out

    def math_selftest(self, fn, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y if y is not None else x, np.float32)
        out = np.empty_like(x)
        self._check(capi.lib().crt_math_selftest(self._h, fn, capi.np_ptr(x), capi.np_ptr(y), x.size, capi.np_ptr(out)))
        return out

    def close(self):
        if self._h:
            capi.lib().crt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
