"""ctypes driver of tests/emu/bmpc_emu.cpp (TEST INFRASTRUCTURE): the solve kernel's source executed on the CPU,
one thread per lane.  Builds the shared object on first use (host clang from ROCm: the kernel source uses clang
vector extensions)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libbmpc_emu.so")
CLANG = os.environ.get("BMPC_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


def build(force=False):
    srcs = [os.path.join(HERE, "bmpc_emu.cpp"), os.path.join(ROOT, "biped_mpc_py_amd", "csrc", "bmpc_kernels.hip"),
            os.path.join(ROOT, "biped_mpc_py_amd", "csrc", "bmpc_stage.hip"), os.path.join(ROOT, "include", "bmpc.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call([CLANG, "-std=c++20", "-O1", "-pthread", "-fPIC", "-shared", "-D_GNU_SOURCE",
                               "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-x", "c++", srcs[0], "-o", SO])
    return SO


def threads(h):
    lib = C.CDLL(build())
    return int(lib.bmpc_emu_threads(int(h)))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def solve(cparams, x_fb, foot, contact, phase, x_cmd=None, mu=None, assemble_only=False, warm=None, warm_load=False,
          warm_shift=0, warm_theta=0.5):
    """Same marshalling as BatchSolver.solve / assemble.  Returns dict.  `warm`: None, or a float64 array
    (B, threads(h), 6) that receives the final solver state and, with warm_load, provides the start."""
    lib = C.CDLL(build())
    # the emulation takes the penalties as they are: the scaling of `penalty_mode` is the library's host arithmetic
    from biped_mpc_py_amd import _lib as _bl
    eff = (C.c_double * 5)()
    _bl.check(_bl.load().bmpc_effective_penalties(C.byref(cparams), eff))
    cp2 = type(cparams)()
    C.memmove(C.byref(cp2), C.byref(cparams), C.sizeof(cparams))
    cp2.rho, cp2.rho_eq_scale, cp2.rho_lo, cp2.rho_hi_f, cp2.rho_hi_m, cp2.penalty_mode = eff[0], eff[1] / eff[0], eff[2], eff[3], eff[4], 1
    cparams = cp2
    h = int(cparams.h)
    x_fb = np.ascontiguousarray(np.asarray(x_fb, np.float32).reshape(-1, 12))
    B = x_fb.shape[0]
    foot = np.ascontiguousarray(np.asarray(foot, np.float32).reshape(B, 6))
    contact = np.ascontiguousarray(np.asarray(contact).reshape(B, h, 2).astype(np.uint8))
    phase = np.ascontiguousarray(np.asarray(phase, np.int32).reshape(B))
    x_cmd = None if x_cmd is None else np.ascontiguousarray(np.asarray(x_cmd, np.float32).reshape(B, 12))
    mu = None if mu is None else np.ascontiguousarray(np.asarray(mu, np.float32).reshape(B, h, 2))
    out = dict(controls=np.zeros((B, h, 12), np.float32), states=np.zeros((B, h, 13), np.float32),
               iters=np.zeros(B, np.int32), residuals=np.zeros((B, 2), np.float32), status=np.zeros(B, np.int32),
               nfactor=np.zeros(B, np.int32), x_ref=np.zeros((B, h, 12)), foot_ref=np.zeros((B, h, 6)),
               Gt=np.zeros((B, 6 * h, 6 * h)), qt=np.zeros((B, 6 * h)))
    lib.bmpc_emu_solve.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 16 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
    rc = lib.bmpc_emu_solve(C.byref(cparams), B, _ptr(x_fb), _ptr(foot), _ptr(contact), _ptr(phase), _ptr(x_cmd), _ptr(mu),
                            _ptr(out["controls"]), _ptr(out["states"]), _ptr(out["iters"]), _ptr(out["residuals"]),
                            _ptr(out["status"]), _ptr(out["nfactor"]), _ptr(out["x_ref"]), _ptr(out["foot_ref"]),
                            _ptr(out["Gt"]), _ptr(out["qt"]), 1 if assemble_only else 0,
                            _ptr(warm), 1 if warm_load else 0, 0 if warm is None else 1, int(warm_shift), float(warm_theta))
    if rc != 0:
        raise RuntimeError("bmpc_emu_solve failed")
    return out
