"""rust-renderer_amd — MI355X-native path-tracing + ReSTIR core (drop-in for the reference's
reference_pt_pass + ReSTIR reservoir passes). Host mirror of the reference surface in api.py,
HIP kernels + C ABI in csrc/ (built to libutopian_hip.so by build.py)."""
from . import camera, distributed, gltf, launch, scenes, types  # noqa: F401
from .api import (  # noqa: F401
    FrameLoop,
    MultiGpuRenderer,
    Renderer,
    UtopianError,
    default_view,
    hip_versions,
    identity3x4,
    load_library,
    make_light,
    make_material,
    transform3x4,
)
from .build import build_library  # noqa: F401
from .types import *  # noqa: F401,F403
