"""orbslam_in_practice_amd -- MI355X (gfx950) hot path of Sunny-cpp/orbslam_In_practice.

The product is the C-ABI shared library built from csrc/*.hip (include/orbslam_amd.h); the Python
classes here mirror the reference's ORBextractor / ORBmatcher / Optimizer call signatures on top of
it for tests and benchmarks.  No CPU fallback: importing works anywhere, but creating an extractor,
matcher or solver without the HIP library / a HIP device raises.
"""
from ._lib import KP_DTYPE, OrbError, build  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import FrameFeatures, ORBmatcher  # noqa: F401
from .optimizer import Optimizer  # noqa: F401
from .tracker import Tracker  # noqa: F401
