"""Python mirror of the reference's `Renderer` / `RenderedImage`
(renderer/renderer.cuh:26-98) on top of the C ABI.  SAMPLES and PASSES
(renderer/settings.cuh:12,15) are arguments instead of compile-time constants."""
import threading

import numpy as np

from . import _capi as capi
from .device import Context
from .scene import Scene


class RenderedImage:
    def __init__(self):
        self.width = self.height = 0
        self.pixels = None      # (H, W, 3) uint8
        self.passes = 0

    def Init(self, w, h):
        self.width, self.height = w, h
        self.pixels = np.zeros((h, w, 3), np.uint8)
        self.passes = 0

    def SaveImage(self, filename):
        if self.pixels is None:
            return False
        px = np.ascontiguousarray(self.pixels)
        return capi.lib().crt_png_write_rgb8(str(filename).encode(), capi.np_ptr(px), self.width, self.height) == capi.OK


class Renderer:
    def __init__(self, samples=256, passes=8, device=0, width=None, height=None, asset_root=None):
        self.samples, self.passes = samples, passes
        self._size = (width, height) if width and height else None
        self._asset_root = asset_root
        self._device = device
        self._ctx = None
        self._scene = None
        self._file = None
        self._image = RenderedImage()
        self._rendering = False
        self._thread = None
        self.last_kernel_ms = 0.0
        self.last_stats = None

    def LoadScene(self, filename):
        try:
            scene = Scene.load(filename, self._asset_root)
            if self._size:
                scene.set_resolution(*self._size)
            if self._ctx is None:
                self._ctx = Context(self._device)
            self._ctx.upload(scene)
        except capi.CrtError as e:
            print(f"ERROR: {e}")
            return False
        self._scene, self._file = scene, str(filename)
        self._image.Init(scene.width, scene.height)
        return True

    def GetImage(self):
        return self._image

    def SceneFileName(self):
        return self._file

    def IsRendering(self):
        return self._rendering

    def BeginRendering(self, wait=False):
        if self._thread is not None:
            self._thread.join()
        self._rendering = True
        self._thread = threading.Thread(target=self._do_rendering)
        self._thread.start()
        if wait:
            self._thread.join()

    def _do_rendering(self):
        ctx, img = self._ctx, self._image
        img.passes = 0
        self.last_kernel_ms = 0.0
        try:
            ctx.clear()
            for p in range(self.passes):
                ctx.render_pass(p, self.samples)
                self.last_kernel_ms += ctx.last_kernel_ms()
                ctx.convert(1.0 / float(np.float32(p + 1)), self._scene.srgb)
                img.pixels = ctx.read_image()
                img.passes += 1
            self.last_stats = ctx.stats()
            img.passes += 1
        finally:
            self._rendering = False
