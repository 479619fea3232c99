"""MI355X-native renderer core for the per-pixel trace path of CUDARaytracing."""
from ._capi import CrtError, EXPORTED_SYMBOLS, LIB_PATH  # noqa: F401
from .device import Context  # noqa: F401
from .renderer import RenderedImage, Renderer  # noqa: F401
from .scene import Scene, build_bvh  # noqa: F401
