"""tracy_amd -- MI355X-native `-kernel hip` render module for Tracy (host-side Python face).

Holds only what the per-sample path-tracing hot path needs: the HIP kernels + C ABI
(csrc/, built into lib/libtracy_hip.so), the host scene loader (lib/libtracy_scene.so) and
thin ctypes mirrors of the reference's Scene / TracyModule interfaces.
"""
from .scene import Scene, SceneError  # noqa: F401
from .module import HipTrace, HipTraceGroup, HipError, split_samples  # noqa: F401
