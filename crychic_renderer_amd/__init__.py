"""crychic_renderer_amd -- MI355X (gfx950) backend for CRYCHIC's per-pixel hot path.

Package layout (only what the path needs):
  csrc/            hand-written HIP kernels, the extern "C" ABI (include/crychic_hip.h) and host constant builders
  _lib.py          ctypes binding of libcrychic_hip.so (raises if the library is not built: no fallback)
  renderer.py      Python mirror of the reference's pass objects (Ssao, DeferredShading, ShadowMap, CRYCHIC.Draw)
  scene.py         synthetic input planes
  geometry.py      meshes / instances / materials of the reference scene for the producer passes (row f1)
  sharding.py      multi-GPU row-strip plan + RCCL all-gather of the composed frame
"""
from ._lib import (CrychicError, Camera, FrameDesc, Light, PassConstants, PassTimes, SsaoConstants, LIGHT_SKY,
                   check, lib)  # noqa: F401
from .renderer import Context, Ssao, DeferredShading, ShadowMap, Crychic, SceneGeometry  # noqa: F401
