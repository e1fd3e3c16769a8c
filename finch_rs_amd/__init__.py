"""finch_rs_amd -- MI355X-native MinHash sketching engine behind finch's sketching interface.

Only the hot path of onecodex/finch-rs lives here: hand-written gfx950 kernels +
the C ABI (csrc/, include/finch_hip.h) and the host-side mirror of the reference's interface for
that path.  Importing this package never falls back to a CPU implementation.
"""
from ._lib import (FinchHipError, SO_PATH, debug_env, debug_set, get_option, load, option_list,  # noqa: F401
                   set_option)
from .sketch_schemes import (BatchSketcher, DeviceBuffer, FinchError, HipSketcher, KmerCount, SketchParams,  # noqa: F401
                             device_count)
