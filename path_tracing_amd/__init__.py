"""MI355X-native ray-casting core for the petersn/path-tracing renderer.

The package holds only the hot path: host-side scene flattening (``host/``), the HIP
kernels behind a C ABI (``csrc/``, ``include/pt_hip.h``) and thin Python handles for
tests and benchmarks.
"""
from .scene import Film, GpuScene, HostScene, cli_camera, cli_lights, make_camera, make_lights  # noqa: F401
from . import meshgen, native  # noqa: F401
