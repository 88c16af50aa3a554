"""ctypes binding of libccmp.so (include/ccmp.h).  Fails loudly when the library is missing:
there is no CPU implementation of the hot path in this package."""
import ctypes as C
import os

from .build import LIBPATH


class CcmpError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__("%s failed: %s (%d)%s" % (what, _strerror(code), code, (" — " + detail) if detail else ""))


class CcmpProblem(C.Structure):
    """ccmp_problem of include/ccmp.h (row-major matrices)."""

    _fields_ = [
        ("axis", C.c_double * 42),
        ("offset", C.c_double * 42),
        ("ee", C.c_double * 6),
        ("R_tool", C.c_double * 18),
        ("base_R", C.c_double * 18),
        ("base_p", C.c_double * 6),
        ("init_R", C.c_double * 9),
        ("init_p", C.c_double * 3),
        ("lb", C.c_double * 7),
        ("ub", C.c_double * 7),
        ("joint_eps", C.c_double),
        ("tol_pos", C.c_double),
        ("tol_rot", C.c_double),
        ("step", C.c_double),
        ("delta", C.c_double),
        ("lambda_", C.c_double),
        ("start_joint", C.c_double * 14),
        ("obj_start_R", C.c_double * 9),
        ("obj_start_p", C.c_double * 3),
        ("obj_goal_R", C.c_double * 9),
        ("obj_goal_p", C.c_double * 3),
        ("t_o7_R", C.c_double * 18),
        ("t_o7_p", C.c_double * 6),
        ("max_iter", C.c_int32),
        ("jacobian_mode", C.c_int32),
        ("arm_index", C.c_int32 * 2),
    ]

    def copy(self):
        return CcmpProblem.from_buffer_copy(bytes(self))


class CcmpSphere(C.Structure):
    """ccmp_sphere of include/ccmp.h"""
    _fields_ = [("frame", C.c_int32), ("group", C.c_int32), ("c", C.c_double * 3), ("r", C.c_double)]


class CcmpBox(C.Structure):
    """ccmp_box of include/ccmp.h"""
    _fields_ = [("group", C.c_int32), ("reserved", C.c_int32), ("c", C.c_double * 3), ("R", C.c_double * 9), ("half", C.c_double * 3)]


CCMP_JAC_FD = 0
CCMP_JAC_ANALYTIC = 1

# every symbol include/ccmp.h declares (tests check the library exports all of them)
EXPORTS = [
    "ccmp_problem_from_yaml", "ccmp_problem_init", "ccmp_set_arms", "ccmp_set_base_frame", "ccmp_set_start", "ccmp_set_tolerance", "ccmp_set_calibration",
    "ccmp_ctx_create", "ccmp_ctx_destroy", "ccmp_ctx_set_waves_per_cu", "ccmp_ctx_set_schedule", "ccmp_ctx_set_option", "ccmp_ctx_get_option", "ccmp_ctx_option_info", "ccmp_ctx_describe", "ccmp_ctx_set_lpt", "ccmp_ctx_device", "ccmp_ctx_num_cus",
    "ccmp_function_batch", "ccmp_project_batch", "ccmp_is_satisfied_batch", "ccmp_joint_valid_batch",
    "ccmp_sample_project_batch", "ccmp_sample_near_project_batch", "ccmp_sample_gaussian_project_batch",
    "ccmp_compute_t_wo_batch", "ccmp_geodesic_batch", "ccmp_check_motion_batch", "ccmp_geodesic_batch_ex", "ccmp_geodesic_host_ex", "ccmp_check_motion_host", "ccmp_ambient_uniform_batch", "ccmp_enforce_bounds_batch", "ccmp_compact_valid", "ccmp_compact_valid_capped",
    "ccmp_project_host", "ccmp_function_host", "ccmp_is_satisfied_host", "ccmp_joint_valid_host", "ccmp_sample_project_host", "ccmp_sample_ref_project_host", "ccmp_geodesic_host", "ccmp_project_sharded_host", "ccmp_sample_project_sharded_host", "ccmp_sharded_host_last_timing",
    "ccmp_comm_create", "ccmp_comm_destroy", "ccmp_comm_last_timing", "ccmp_project_sharded", "ccmp_sample_project_sharded",
    "ccmp_scene_create", "ccmp_scene_destroy", "ccmp_scene_num_pairs", "ccmp_clearance_batch", "ccmp_clearance_host",
    "ccmp_strerror",
    "ccmp_last_hip_error", "ccmp_version", "ccmp_problem_sizeof",
]

_lib = None
# include/ccmp_debug.h: exported by lib/libccmp_debug.so only (the same sources with -DCCMP_DEBUG_HOOKS; select it with CCMP_LIBRARY)
DEBUG_EXPORTS = ["ccmp_detmath_probe", "ccmp_ctx_set_order_experimental", "ccmp_ctx_debug_lpt_pred", "ccmp_debug_fail_calls"]
DEBUG_LIBPATH = os.path.join(os.path.dirname(LIBPATH), "libccmp_debug.so")


def lib():
    """The loaded libccmp.so; raises if it has not been built (python -m closed_chain_motion_planner_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise ImportError(
            "libccmp.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950). There is no CPU fallback." % LIBPATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 and must be the one that
    # initialises the device if it is going to own the memory we launch on.  Loading libccmp.so first
    # would pull /opt/rocm's copy in, and the device then fails to open for whichever comes second
    # (seen on the GPU box: "no usable HIP device").  Importing torch first makes the loader resolve
    # libccmp's libamdhip64.so.7 dependency to the already-loaded runtime.
    try:
        import torch  # noqa: F401
    except ImportError:  # C-only consumers link libccmp.so against the system ROCm directly
        pass
    L = C.CDLL(LIBPATH)
    dp, u8p, u16p, vp = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.c_void_p
    pp = C.POINTER(CcmpProblem)
    sig = {
        "ccmp_problem_from_yaml": ([C.c_char_p, pp], C.c_int),
        "ccmp_problem_init": ([pp, C.c_char_p, C.c_int, C.c_char_p, C.c_int, dp, dp, dp, dp, dp], C.c_int),
        "ccmp_set_arms": ([pp, C.c_char_p, C.c_int, C.c_char_p, C.c_int], C.c_int),
        "ccmp_set_start": ([pp, dp], C.c_int),
        "ccmp_set_tolerance": ([pp, C.c_double, C.c_double], C.c_int),
        "ccmp_set_base_frame": ([pp, C.c_int, dp, dp], C.c_int),
        "ccmp_set_calibration": ([pp, C.c_int, dp], C.c_int),
        "ccmp_ctx_create": ([C.c_int, C.POINTER(vp)], C.c_int),
        "ccmp_ctx_destroy": ([vp], None),
        "ccmp_ctx_set_waves_per_cu": ([vp, C.c_int], C.c_int),
        "ccmp_ctx_set_schedule": ([vp, C.c_int, C.c_size_t], C.c_int),
        "ccmp_ctx_set_option": ([vp, C.c_char_p, C.c_long], C.c_int),
        "ccmp_ctx_get_option": ([vp, C.c_char_p, C.POINTER(C.c_long)], C.c_int),
        "ccmp_ctx_option_info": ([C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long),
                                  C.POINTER(C.c_char_p)], C.c_int),
        "ccmp_ctx_describe": ([vp, C.c_int, C.c_size_t, C.c_char_p, C.c_size_t], C.c_int),
        "ccmp_ctx_set_lpt": ([vp, C.c_int, C.c_size_t], C.c_int),
        "ccmp_ctx_device": ([vp], C.c_int),
        "ccmp_ctx_num_cus": ([vp], C.c_int),
        "ccmp_function_batch": ([vp, pp, vp, vp, C.c_size_t, vp], C.c_int),
        "ccmp_project_batch": ([vp, pp, vp, vp, vp, vp, C.c_size_t, vp], C.c_int),
        "ccmp_is_satisfied_batch": ([vp, pp, vp, vp, C.c_size_t, vp], C.c_int),
        "ccmp_joint_valid_batch": ([vp, pp, vp, vp, C.c_size_t, vp], C.c_int),
        "ccmp_sample_project_batch": ([vp, pp, C.c_uint64, C.c_uint64, vp, vp, vp, vp, C.c_size_t, vp], C.c_int),
        "ccmp_sample_near_project_batch": ([vp, pp, C.c_uint64, C.c_uint64, vp, C.c_int, C.c_double, vp, vp, vp, vp,
                                            C.c_size_t, vp], C.c_int),
        "ccmp_sample_gaussian_project_batch": ([vp, pp, C.c_uint64, C.c_uint64, vp, C.c_int, C.c_double, vp, vp, vp, vp,
                                                C.c_size_t, vp], C.c_int),
        "ccmp_compute_t_wo_batch": ([vp, pp, vp, C.c_int, vp, C.c_size_t, vp], C.c_int),
        "ccmp_geodesic_batch": ([vp, pp, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp], C.c_int),
        "ccmp_check_motion_batch": ([vp, pp, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp], C.c_int),
        "ccmp_geodesic_batch_ex": ([vp, pp, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp], C.c_int),
        "ccmp_geodesic_host_ex": ([vp, pp, dp, dp, C.c_size_t, C.c_int, dp, C.POINTER(C.c_int32), u8p, dp, dp, C.c_int, C.c_int], C.c_int),
        "ccmp_check_motion_host": ([vp, pp, dp, dp, C.c_size_t, C.c_int, dp, C.POINTER(C.c_int32), u8p], C.c_int),
        "ccmp_ambient_uniform_batch": ([vp, pp, C.c_uint64, C.c_uint64, vp, C.c_size_t, vp], C.c_int),
        "ccmp_enforce_bounds_batch": ([vp, vp, C.c_size_t, vp], C.c_int),
        "ccmp_compact_valid": ([vp, vp, vp, C.c_size_t, vp, vp, vp], C.c_int),
        "ccmp_compact_valid_capped": ([vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp], C.c_int),
        "ccmp_project_host": ([vp, pp, dp, dp, u8p, u16p, C.c_size_t], C.c_int),
        "ccmp_function_host": ([vp, pp, dp, dp, C.c_size_t], C.c_int),
        "ccmp_is_satisfied_host": ([vp, pp, dp, u8p, C.c_size_t], C.c_int),
        "ccmp_joint_valid_host": ([vp, pp, dp, u8p, C.c_size_t], C.c_int),
        "ccmp_sample_project_host": ([vp, pp, C.c_uint64, C.c_uint64, dp, u8p, u16p, C.c_size_t], C.c_int),
        "ccmp_sample_ref_project_host": ([vp, pp, C.c_int, C.c_uint64, C.c_uint64, dp, C.c_double, dp, u8p, u16p, C.c_size_t], C.c_int),
        "ccmp_geodesic_host": ([vp, pp, dp, dp, C.c_size_t, C.c_int, dp, C.POINTER(C.c_int32), u8p], C.c_int),
        "ccmp_project_sharded_host": ([C.POINTER(vp), C.c_int, pp, dp, dp, u8p, u16p, C.c_size_t], C.c_int),
        "ccmp_sample_project_sharded_host": ([C.POINTER(vp), C.c_int, pp, C.c_uint64, C.c_uint64, dp, u8p, u16p, C.c_size_t], C.c_int),
        "ccmp_sharded_host_last_timing": ([C.POINTER(vp), C.c_int, dp, dp], C.c_int),
        "ccmp_comm_create": ([C.POINTER(vp), C.c_int, C.POINTER(vp)], C.c_int),
        "ccmp_comm_destroy": ([vp], None),
        "ccmp_comm_last_timing": ([vp, dp, dp], C.c_int),
        "ccmp_project_sharded": ([vp, pp, dp, C.c_size_t, dp, u8p, u16p, C.c_size_t, dp, C.c_size_t, C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint64)], C.c_int),
        "ccmp_sample_project_sharded": ([vp, pp, C.c_uint64, C.c_uint64, C.c_size_t, dp, u8p, u16p, C.c_size_t, dp, C.c_size_t,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)], C.c_int),
        "ccmp_scene_create": ([vp, C.POINTER(CcmpSphere), C.c_int, C.POINTER(CcmpBox), C.c_int, C.POINTER(C.c_uint32), C.POINTER(vp)], C.c_int),
        "ccmp_scene_destroy": ([vp], None),
        "ccmp_scene_num_pairs": ([vp], C.c_int),
        "ccmp_clearance_batch": ([vp, pp, vp, vp, vp, C.c_size_t, C.c_double, vp, vp, vp, vp], C.c_int),
        "ccmp_clearance_host": ([vp, pp, vp, dp, C.c_size_t, C.c_double, dp, C.POINTER(C.c_int32), u8p], C.c_int),
        "ccmp_strerror": ([C.c_int], C.c_char_p),
        "ccmp_last_hip_error": ([], C.c_char_p),
        "ccmp_version": ([], C.c_int),
        "ccmp_problem_sizeof": ([], C.c_size_t),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)  # AttributeError here = library out of date with the header
        fn.argtypes = args
        fn.restype = res
    debug_sig = {
        "ccmp_ctx_set_order_experimental": ([vp, vp], C.c_int),
        "ccmp_ctx_debug_lpt_pred": ([vp, vp, C.c_size_t], C.c_int),
        "ccmp_detmath_probe": ([vp, vp, vp, vp, C.c_size_t, vp], C.c_int),
        "ccmp_debug_fail_calls": ([vp, C.c_int], C.c_int),
    }
    for name, (args, res) in debug_sig.items():  # present in the debug build only (include/ccmp_debug.h)
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
    if L.ccmp_problem_sizeof() != C.sizeof(CcmpProblem):
        raise ImportError("ccmp_problem layout mismatch between libccmp.so and the Python binding")
    _lib = L
    return L


CALL_PROJECT, CALL_SAMPLE_PROJECT, CALL_PROJECT_ANALYTIC, CALL_GEODESIC, CALL_GEODESIC_BUDGET = range(5)  # ccmp.h: CCMP_CALL_*


def option_table():
    """every tuning option of a context as the library itself states it (ccmp_ctx_option_info): name, built-in default, range,
    meaning — no device needed"""
    L = lib()
    out, i = [], 0
    while True:
        name, doc = C.c_char_p(), C.c_char_p()
        d, lo, hi = C.c_long(), C.c_long(), C.c_long()
        if L.ccmp_ctx_option_info(i, C.byref(name), C.byref(d), C.byref(lo), C.byref(hi), C.byref(doc)) != 0:
            return out
        out.append({"name": name.value.decode(), "default": d.value, "lo": lo.value, "hi": hi.value, "doc": doc.value.decode()})
        i += 1


def get_option(ctx_handle, name):
    """ccmp_ctx_get_option; ctx_handle None = the built-in default"""
    v = C.c_long()
    check(lib().ccmp_ctx_get_option(ctx_handle, name.encode(), C.byref(v)), "ccmp_ctx_get_option(%s)" % name)
    return v.value


def describe(ctx_handle, call_kind, n):
    """ccmp_ctx_describe: one line naming the kernels and thresholds the policy takes for a call of n samples / edges
    (ctx_handle None = the built-in policy on a 256-CU device)"""
    buf = C.create_string_buffer(2048)
    rc = lib().ccmp_ctx_describe(ctx_handle, int(call_kind), int(n), buf, len(buf))
    if rc < 0:
        check(rc, "ccmp_ctx_describe")
    return buf.value.decode()


def _strerror(code):
    try:
        return lib().ccmp_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "error"


def check(code, what):
    if code != 0:
        detail = ""
        if code == -2:
            detail = lib().ccmp_last_hip_error().decode()
        raise CcmpError(code, what, detail)
