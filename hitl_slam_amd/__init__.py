"""hitl_slam_amd -- MI355X (gfx950) backend for hitl-slam's batch re-optimisation path.

Only what the hot path needs: csrc/ (HIP kernels + the C ABI of include/hitl_gpu.h) and the
host-side mirror of the reference's JointOpt interface (api.py).  No CPU fallback exists:
using a Solver without the built HIP library raises.
"""
from . import abi  # noqa: F401
from .abi import (MODE_ODOM_HC, MODE_STF_ONLY, MODE_JOINT, CONVERGENCE, NO_CONVERGENCE, FAILURE,  # noqa: F401
                  HC_LINE_SEGMENT, HC_COLINEAR, HC_PERPENDICULAR, HC_PARALLEL, default_options)
from .api import Solver, HitlError, load_library, LIB_PATH, hc_rows, set_test_switch  # noqa: F401
