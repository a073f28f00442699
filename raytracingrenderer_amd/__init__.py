"""MI355X-native path-tracing hot path of RayTracingRenderer.

The product is two shared libraries built in-tree (see ``__graft_entry__.build``):

* ``librtr_hip.so``  -- hand-written HIP kernels for gfx950 behind the C ABI of
  ``include/rtr.h``;
* ``librtr_host.so`` -- the C++ mirror of the reference's Scene / Object /
  Material / Render API (``host/rtr_host.hpp``) that flattens a scene and calls
  that C ABI.

This package is only a thin ctypes binding over both, used by the tests, the
benchmark and Python callers.  It has no CPU implementation of anything: if the
libraries are missing, importing :mod:`raytracingrenderer_amd.api` raises.
"""
from .api import (  # noqa: F401
    DIFFUSE, MICROFACET, MODE_AUTO, MODE_FUSED, MODE_WAVEFRONT, SPECULAR, HostScene, RenderParams, RtrError, Stats, default_scene,
    build_bvh, device_count, hip_lib, host_lib, partition_columns, save_ppm, self_test,
)
