"""ctypes binding of libbmpc.so (include/bmpc.h).  No fallback: if the shared library is missing
or no HIP device is usable, the product path raises -- it never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbmpc.so")

ABI_VERSION = 11

# every symbol include/bmpc.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = (
    "bmpc_abi_version", "bmpc_last_error", "bmpc_supported_horizon", "bmpc_supported_horizon_path", "bmpc_solver_path", "bmpc_rescue_enabled",
    "bmpc_effective_penalties",
    "bmpc_default_params",
    "bmpc_create", "bmpc_destroy", "bmpc_set_params", "bmpc_get_params",
    "bmpc_solve_batch", "bmpc_solve_batch_f64", "bmpc_solve_batch_device", "bmpc_synchronize",
    "bmpc_host_io", "bmpc_solve_batch_io", "bmpc_host_io_generation",
    "bmpc_debug_assemble", "bmpc_debug_set_profile", "bmpc_last_kernel_ms",
    "bmpc_foot_position_world", "bmpc_foot_position_world_device",
    "bmpc_low_level_control", "bmpc_low_level_control_device",
    "bmpc_gait_default", "bmpc_contact_sequence", "bmpc_contact_sequence_device",
    "bmpc_set_warm_start", "bmpc_reset_warm_start", "bmpc_rollout_device", "bmpc_set_dispatch_order",
)


class CGait(C.Structure):
    """`bmpc_gait` of include/bmpc.h."""
    _fields_ = [("period", C.c_int32), ("offset", C.c_int32 * 2), ("duty", C.c_int32 * 2)]


class CHostViews(C.Structure):
    """`bmpc_host_views` of include/bmpc.h: the arrays of a handle's page-locked I/O block."""
    _fields_ = [(n, C.c_void_p) for n in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu", "controls", "states",
                                           "iters", "residuals", "status", "nfactor")]


class BmpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbmpc error {code}: {msg}")
        self.code = code


class CParams(C.Structure):
    """struct bmpc_params (include/bmpc.h)."""
    _fields_ = [
        ("h", C.c_int32), ("half", C.c_int32),
        ("dt", C.c_double), ("kv", C.c_double),
        ("x_cmd", C.c_double * 12), ("Q", C.c_double * 13), ("R", C.c_double * 12),
        ("m", C.c_double), ("I", C.c_double * 9),
        ("lt", C.c_double), ("lh", C.c_double), ("g", C.c_double), ("mu", C.c_double),
        ("f_max", C.c_double * 3), ("f_min", C.c_double * 3),
        ("tau_max", C.c_double * 3), ("tau_min", C.c_double * 3),
        ("rho", C.c_double), ("rho_eq_scale", C.c_double), ("rho_lo", C.c_double),
        ("rho_hi_f", C.c_double), ("rho_hi_m", C.c_double), ("kappa", C.c_double), ("alpha", C.c_double),
        ("eps_pri", C.c_double), ("eps_dua", C.c_double),
        ("max_iter", C.c_int32), ("check_every", C.c_int32), ("adapt_start", C.c_int32),
        ("adapt_every", C.c_int32), ("max_refactor", C.c_int32), ("warm_adapt_start", C.c_int32),
        ("path", C.c_int32), ("penalty_mode", C.c_int32), ("rescue", C.c_int32), ("accel", C.c_int32),
        ("adapt_early", C.c_int32), ("adapt_late", C.c_int32), ("adapt_busy", C.c_int32), ("adapt_flips", C.c_int32),
        ("confirm_from", C.c_int32), ("reserved0", C.c_int32), ("kappa_confirm", C.c_double),
        ("kp", C.c_double * 9), ("kd", C.c_double * 9), ("swingHeight", C.c_double), ("hip_offset", C.c_double * 3),
    ]


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so.7; if libbmpc.so pulled in
    /opt/rocm's copy first and torch then loaded its own, the second runtime finds no device ("No HIP GPUs
    are available").  So when torch is installed, its copy is loaded (globally) before libbmpc.so, whose
    DT_NEEDED libamdhip64.so.7 then resolves to it -- whether or not the caller ever imports torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libbmpc.so once and declare prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  biped_mpc_py_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    lib.bmpc_abi_version.restype = ip
    lib.bmpc_last_error.restype = C.c_char_p
    lib.bmpc_supported_horizon.argtypes = [ip]
    lib.bmpc_supported_horizon_path.argtypes = [ip, ip]
    lib.bmpc_solver_path.argtypes = [vp]
    lib.bmpc_rescue_enabled.argtypes = [vp]
    lib.bmpc_rescue_enabled.restype = ip
    lib.bmpc_effective_penalties.argtypes = [C.POINTER(CParams), C.POINTER(C.c_double)]
    lib.bmpc_default_params.argtypes = [C.POINTER(CParams), ip]
    lib.bmpc_create.argtypes = [C.POINTER(vp), C.POINTER(CParams), ip, ip]
    lib.bmpc_destroy.argtypes = [vp]
    lib.bmpc_set_params.argtypes = [vp, C.POINTER(CParams)]
    lib.bmpc_get_params.argtypes = [vp, C.POINTER(CParams)]
    ptrs14 = [vp] * 12
    lib.bmpc_solve_batch.argtypes = [vp, ip] + ptrs14
    lib.bmpc_solve_batch_f64.argtypes = [vp, ip] + ptrs14
    lib.bmpc_solve_batch_device.argtypes = [vp, ip] + ptrs14 + [vp]
    lib.bmpc_synchronize.argtypes = [vp]
    lib.bmpc_host_io.argtypes = [vp, ip, ip, ip, ip, C.POINTER(CHostViews)]
    lib.bmpc_solve_batch_io.argtypes = [vp, ip]
    lib.bmpc_debug_assemble.argtypes = [vp, ip] + [vp] * 10
    lib.bmpc_debug_set_profile.argtypes = [vp, vp]
    lib.bmpc_foot_position_world.argtypes = [vp, ip, vp, vp, vp]
    lib.bmpc_foot_position_world_device.argtypes = [vp, ip, vp, vp, vp, vp]
    lib.bmpc_low_level_control.argtypes = [vp, ip] + [vp] * 8
    lib.bmpc_low_level_control_device.argtypes = [vp, ip] + [vp] * 9
    lib.bmpc_last_kernel_ms.argtypes = [vp, fp]
    lib.bmpc_gait_default.argtypes = [C.POINTER(CGait), ip]
    lib.bmpc_contact_sequence.argtypes = [vp, ip, vp, C.POINTER(CGait), vp, vp]
    lib.bmpc_contact_sequence_device.argtypes = [vp, ip, vp, C.POINTER(CGait), vp, vp, vp]
    lib.bmpc_set_warm_start.argtypes = [vp, ip, ip, C.c_double]
    lib.bmpc_reset_warm_start.argtypes = [vp]
    lib.bmpc_set_dispatch_order.argtypes = [vp, vp, ip]
    lib.bmpc_rollout_device.argtypes = [vp, ip, ip, vp, vp, vp, C.POINTER(CGait), vp, vp, vp, vp, vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name != "bmpc_last_error":
            fn.restype = ip
    if lib.bmpc_abi_version() != ABI_VERSION:
        raise ImportError(f"libbmpc ABI {lib.bmpc_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise BmpcError(rc, load().bmpc_last_error().decode("utf-8", "replace"))
