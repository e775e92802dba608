"""bhusie_amd — MI355X-native geodesic ray-trace pass behind the reference's RayPipeline surface.

The product is libbhray.so (hand-written gfx950 HIP kernels + a C ABI, include/bhray.h).  This
package is the thin Python host side used by the tests and bench: ctypes bindings plus mirrors of
the reference host types on the path (RayDetails, CameraUniform, BlackHoleUniform, Model,
RayPipeline ladder).  Nothing here computes pixels on the CPU; importing the bindings fails loudly
when the HIP library has not been built.
"""
import os as _os

# Frame slots run on separate HIP streams and ROCm maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; aliased
# streams serialise.  This host layer keeps up to 16 frames in flight (+ the communication streams of a gather), so it raises
# the limit unless the user set one — it must happen before the first HIP call.  (libbhray itself never touches the environment.)
# SIDE EFFECT: the variable is process-wide, so it also applies to torch or any other HIP user that initialises after this
# import; export GPU_MAX_HW_QUEUES yourself (any value, e.g. the ROCm default 4) before importing bhusie_amd to opt out.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from ._lib import lib, LibraryMissing, LIB_PATH  # noqa: F401
from .layouts import (BhrayConfig, BhrayCounters, BhrayTiming, BhrayDetails, BhrayCameraUniform,  # noqa: F401
                      BhrayBlackHoleUniform, BhrayBlackHole, BhrayModelDesc, BhrayNode, BhrayTriangle,
                      BhrayError, check)
from .scene import Camera, BlackHole, RayDetails  # noqa: F401
from .model import Model, load_model  # noqa: F401
from .renderer import RayPass, Renderer, PinnedFrame, ladder_from_base, ladder_for_frame, comm_unique_id, partition_rows, config_partition_rows, balance_slabs, rebalance_slabs  # noqa: F401
