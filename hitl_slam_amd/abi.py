"""ctypes mirror of include/hitl_gpu.h (POD types and constants only).

Field order and types must match the header exactly; tests/test_abi.py checks the sizes
against values compiled from the header.
"""
import ctypes as C
import math

ABI_VERSION = 1

# status codes
OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_NO_DEVICE = 2
ERR_HIP = 3
ERR_OUT_OF_MEMORY = 4
ERR_NOT_READY = 5
ERR_UNSUPPORTED = 6
ERR_SOLVER_FAILURE = 7
ERR_COMM = 8

# modes (JointOptimization.cpp:1064-1138 / :1156-1256 / both)
MODE_ODOM_HC = 0
MODE_STF_ONLY = 1
MODE_JOINT = 2

# termination
CONVERGENCE = 0
NO_CONVERGENCE = 1
FAILURE = 2

# CorrectionType (human_constraints.h:8-16)
HC_LINE_SEGMENT = 2
HC_COLINEAR = 4
HC_PERPENDICULAR = 5
HC_PARALLEL = 6


class HumanConstraint(C.Structure):
    """== struct HumanConstraint (human_constraints.h:29-41), 28 bytes."""
    _fields_ = [
        ("type", C.c_uint32),
        ("constrained_pose_id", C.c_int32),
        ("anchor_pose_id", C.c_int32),
        ("delta_parallel", C.c_float),
        ("delta_perpendicular", C.c_float),
        ("delta_angle", C.c_float),
        ("relative_penalty_dir", C.c_float),
    ]


# numpy view of the same 28-byte row
import numpy as _np
HC_DTYPE = _np.dtype([("type", "<u4"), ("constrained_pose_id", "<i4"), ("anchor_pose_id", "<i4"), ("delta_parallel", "<f4"),
                      ("delta_perpendicular", "<f4"), ("delta_angle", "<f4"), ("relative_penalty_dir", "<f4")])
assert HC_DTYPE.itemsize == 28


class Options(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("device", C.c_int32),
        ("laser_std_dev", C.c_float),
        ("point_correlation_factor", C.c_float),
        ("point_match_threshold", C.c_float),
        ("max_stf_angle_error", C.c_float),
        ("max_correspondences_per_point", C.c_int32),
        ("num_skip_readings", C.c_int32),
        ("min_inter_pose_correspondence", C.c_int32),
        ("odom_radial_std_dev", C.c_float),
        ("odom_tangential_std_dev", C.c_float),
        ("odom_angular_std_dev", C.c_float),
        ("max_num_iterations", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("linear_solver_tolerance", C.c_double),
        ("max_linear_iterations", C.c_int32),
        ("max_band_poses", C.c_int32),
        ("rank", C.c_int32),
        ("world_size", C.c_int32),
        ("rccl_unique_id", C.c_void_p),
        ("stf_moments", C.c_int64),
        ("timing_level", C.c_int64),
        ("kd_tree_search", C.c_int64),
        ("tolerances_after_accept", C.c_int64),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32),
        ("iterations", C.c_int32),
        ("successful_steps", C.c_int32),
        ("unsuccessful_steps", C.c_int32),
        ("linear_iterations", C.c_int32),
        ("num_residuals", C.c_int32),
        ("num_hc_residuals", C.c_int32),
        ("band_poses", C.c_int32),
        ("num_pairs", C.c_int64),
        ("num_matches", C.c_int64),
        ("num_offband_pairs", C.c_int64),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("final_gradient_max_norm", C.c_double),
        ("final_trust_region_radius", C.c_double),
        ("total_ms", C.c_double),
        ("linearize_ms", C.c_double),
        ("linear_solver_ms", C.c_double),
        ("linearize_calls", C.c_int32),
        ("linearize_timed_calls", C.c_int32),
        ("submap_linear_solver", C.c_double),
        ("hbm_bytes", C.c_double),
        ("pcie_bytes", C.c_double),
        ("pcg_ms", C.c_double),
        ("pcg_timed_calls", C.c_double),
        ("pcg_timed_iterations", C.c_double),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            if name.startswith("reserved"):
                continue
            out[name] = getattr(self, name)
        return out


def default_options():
    """Python-side copy of hitl_options_default (include/hitl_gpu.h documents each value)."""
    o = Options()
    o.abi_version = ABI_VERSION
    o.device = 0
    o.laser_std_dev = 0.05
    o.point_correlation_factor = 1.0 / 40.0
    o.point_match_threshold = 0.15
    o.max_stf_angle_error = math.radians(25.0)
    o.max_correspondences_per_point = 6
    o.num_skip_readings = 1
    o.min_inter_pose_correspondence = 10
    o.odom_radial_std_dev = 0.03
    o.odom_tangential_std_dev = 0.03
    o.odom_angular_std_dev = 0.01
    o.max_num_iterations = 100
    o.jacobi_scaling = 1
    o.function_tolerance = 1e-6
    o.gradient_tolerance = 1e-10
    o.parameter_tolerance = 1e-8
    o.initial_trust_region_radius = 1e4
    o.max_trust_region_radius = 1e16
    o.min_trust_region_radius = 1e-32
    o.min_relative_decrease = 1e-3
    o.min_lm_diagonal = 1e-6
    o.max_lm_diagonal = 1e32
    o.linear_solver_tolerance = 1e-12
    o.max_linear_iterations = 500
    o.max_band_poses = 4
    o.rank = 0
    o.world_size = 1
    o.timing_level = 1
    o.kd_tree_search = 1
    o.rccl_unique_id = None
    return o


def copy_options(o):
    n = Options()
    C.memmove(C.byref(n), C.byref(o), C.sizeof(Options))
    return n
