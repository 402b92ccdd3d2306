"""Host side of the drop-in: the reference's `solve_mpc` call surface over libbmpc.so.

    states, controls = solve_mpc(x_fb, t, foot, mpc, biped, contact)        # REF:187, 304, 487

returns new fp64 arrays `states (h,13)`, `controls (h,12)` exactly like the reference, so
`u0 = controls[0, :].reshape(-1, 1)` -> `lowLevelControl(...)` (REF:493-494) keeps working.  The
three prints of REF:190-192 are not reproduced.  `solve_mpc_batch` is the same for B instances.
All arithmetic happens in the HIP kernels; this module only marshals arrays.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .params import pack_params, params_key


def phase_index(t, mpc_or_dt, h=None):
    """k = int(t // dt) % h, evaluated in fp64 on the host exactly as REF:56-57 / REF:99-100."""
    if h is None:
        dt, h = mpc_or_dt.dt, mpc_or_dt.h
    else:
        dt = mpc_or_dt
    return int(t // dt) % int(h)


def phase_indices(t, dt, h):
    """`phase_index` for an array of times: np.floor_divide on float64 has CPython's `//` semantics (the quotient is
    corrected by the fmod remainder, REF:56-57 / 99-100 rely on exactly that at step boundaries), so this is the
    same arithmetic without a Python loop (tests/test_host_logic.py holds it to `phase_index` on and next to 3000
    step boundaries).  t: array of seconds (>= 0) -> int32 array."""
    tt = np.asarray(t, np.float64).reshape(-1)
    k = np.floor_divide(tt, np.float64(dt))
    return (k.astype(np.int64) % int(h)).astype(np.int32)


def _contact_u8(contact, B, h):
    """(B, h, 2) uint8 contact table with entries in {0, 1}.  uint8 / bool input takes one pass (a max), anything
    else is checked value by value."""
    c = np.asarray(contact)
    if c.shape[-1] != 2 or c.size != B * h * 2:
        raise ValueError(f"contact must have shape (B, {h}, 2)")
    if c.dtype == np.uint8 or c.dtype == np.bool_:
        c8 = c.view(np.uint8) if c.dtype == np.bool_ else c
        if c8.size and c8.max() > 1:
            raise ValueError("contact entries must be 0 or 1")
    else:
        c8 = c.astype(np.uint8)
        if not np.array_equal(c8, c) or (c8.size and c8.max() > 1):
            raise ValueError("contact entries must be 0 or 1")
    return np.ascontiguousarray(c8.reshape(B, h, 2))


def get_contact_sequence(t, mpc, half=None):
    """REF:50-59.  Default: the reference's 20x2 table of 5-on/5-off, rows k..k+9 (ten rows whatever
    `mpc.h` is -- reference quirk).  With `half` given: the same schedule with that half period, continued
    periodically, rows k..k+h-1 (what `BatchSolver.contact_sequence` computes on the device)."""
    if half is None:
        half_, nrow = 5, 10
    else:
        half_, nrow = int(half), int(mpc.h)
    k = phase_index(t, mpc)
    leg0 = ((k + np.arange(nrow)) // half_) % 2 == 0
    return np.stack([leg0, ~leg0], axis=1).astype(int)


def _ptr(a):
    """Address of a NumPy array as an integer (what a `c_void_p` parameter takes; building a ctypes pointer object per argument
    costs ~2 us each, fourteen of them per solve)."""
    return None if a is None else a.__array_interface__["data"][0]


class _HandleOwner:
    """The native handle's lifetime as a Python object: `bmpc_destroy` runs when the LAST reference goes -- the solver's own, or
    that of an array `solve_inplace` returned.  Those arrays are views of the handle's page-locked I/O block, which
    `bmpc_destroy` frees; every view keeps this object alive through the buffer it is built on, so closing (or losing) the
    solver while results are still held defers the destruction instead of leaving them pointing at freed memory."""

    def __init__(self, lib, value):
        self._lib, self._value = lib, value

    def __del__(self):
        try:
            if self._value:
                self._lib.bmpc_destroy(C.c_void_p(self._value))
                self._value = None
        except Exception:
            pass


class BatchSolver:
    """Owns one `bmpc_handle` (device memory + stream) for a fixed parameter block."""

    def __init__(self, mpc=None, biped=None, half=None, device=0, max_batch=65536, solver_options=None,
                 cparams=None):
        self._lib = _lib.load()
        self.cparams = cparams if cparams is not None else pack_params(mpc, biped, half, solver_options)
        self.h = int(self.cparams.h)
        self.device = int(device)
        self.max_batch = int(max_batch)
        self._h = C.c_void_p()
        _lib.check(self._lib.bmpc_create(C.byref(self._h), C.byref(self.cparams), self.device, self.max_batch))
        self._owner = _HandleOwner(self._lib, self._h.value)
        self._io, self._io_key = None, None

    def set_params(self, cparams):
        """Replace the parameter block of this handle (same horizon): `bmpc_set_params`."""
        _lib.check(self._lib.bmpc_set_params(self._h, C.byref(cparams)))
        self.cparams = cparams

    def close(self):
        """Give the handle up.  It is destroyed at once unless arrays returned by `solve_inplace` are still alive: those are views
        of the handle's page-locked block, and the native handle (device memory included) then lives until the last of them goes
        -- copy what must outlive the solver if the memory is to come back now."""
        self._io, self._io_key = None, None
        self._owner = None                         # (the last reference unless views are alive: _HandleOwner.__del__ destroys)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host arrays --------------------------------------------------------------------------
    def _marshal(self, x_fb, foot, contact, phase, x_cmd, mu):
        h = self.h
        x_fb = np.ascontiguousarray(np.asarray(x_fb, np.float32).reshape(-1, 12))
        B = x_fb.shape[0]
        foot = np.ascontiguousarray(np.asarray(foot, np.float32).reshape(B, 6))
        contact = _contact_u8(contact, B, h)
        phase = np.ascontiguousarray(np.asarray(phase, np.int32).reshape(B))
        if x_cmd is not None:
            x_cmd = np.ascontiguousarray(np.asarray(x_cmd, np.float32).reshape(B, 12))
        if mu is not None:
            mu = np.ascontiguousarray(np.asarray(mu, np.float32).reshape(B, h, 2))
        return B, x_fb, foot, contact, phase, x_cmd, mu

    def solve(self, x_fb, foot, contact, phase, x_cmd=None, mu=None, want_states=True, out=None):
        """Host arrays in, host arrays out (fp32 over PCIe, fp64 returned -- the reference's dtype, REF:300-304; the
        widening happens inside `bmpc_solve_batch_f64` while the results are unpacked, overlapped with the solve).  Returns
        (states (B,h,13) | None, controls (B,h,12), info).  `out`: optional (states | None, controls) fp64 C-contiguous
        arrays of those shapes to write into (a control loop reuses its buffers instead of allocating 8 MB per call)."""
        B, x_fb, foot, contact, phase, x_cmd, mu = self._marshal(x_fb, foot, contact, phase, x_cmd, mu)
        h = self.h
        if out is not None:
            states, controls = out
            for a, shp in ((states, (B, h, 13)), (controls, (B, h, 12))):
                if a is not None and not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == shp):
                    raise ValueError(f"out arrays must be C-contiguous float64 of shape {shp}")
            if controls is None:
                raise ValueError("out = (states | None, controls): controls is required")
            if not want_states:
                states = None
        else:
            controls = np.empty((B, h, 12), np.float64)
            states = np.empty((B, h, 13), np.float64) if want_states else None
        iters = np.empty(B, np.int32)
        status = np.empty(B, np.int32)
        nfactor = np.empty(B, np.int32)
        resid = np.empty((B, 2), np.float32)
        _lib.check(self._lib.bmpc_solve_batch_f64(
            self._h, B, _ptr(x_fb), _ptr(foot), _ptr(contact), _ptr(phase), _ptr(x_cmd), _ptr(mu),
            _ptr(controls), _ptr(states), _ptr(iters), _ptr(resid), _ptr(status), _ptr(nfactor)))
        info = dict(iters=iters, status=status, nfactor=nfactor, residuals=resid)
        return states, controls, info

    def _io_views(self, B, with_x_cmd, with_mu, with_states):
        """NumPy views of the handle's page-locked I/O block laid out for batches of B (`bmpc_host_io`); cached per layout."""
        # (the key carries the C side's layout generation: a raw bmpc_host_io on this handle, or one that failed part-way,
        #  moves it, and the cached views -- whose offsets belong to the layout they were made for -- are not trusted any more)
        key = (B, bool(with_x_cmd), bool(with_mu), bool(with_states), int(self._lib.bmpc_host_io_generation(self._h)))
        if self._io_key == key:
            return self._io
        self._io, self._io_key = None, None
        v = _lib.CHostViews()
        _lib.check(self._lib.bmpc_host_io(self._h, B, int(key[1]), int(key[2]), int(key[3]), C.byref(v)))
        key = key[:4] + (int(self._lib.bmpc_host_io_generation(self._h)),)
        h = self.h
        owner = self._owner

        def view(addr, dtype, shape):
            if not addr:
                return None
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            buf = (C.c_char * n).from_address(addr)
            buf._owner = owner                     # the array's base: keeps the native handle alive as long as the view is
            return np.frombuffer(buf, dtype=dtype).reshape(shape)

        self._io = dict(x_fb=view(v.x_fb, np.float32, (B, 12)), foot=view(v.foot, np.float32, (B, 6)),
                        contact=view(v.contact, np.uint8, (B, h, 2)), phase=view(v.phase, np.int32, (B,)),
                        x_cmd=view(v.x_cmd, np.float32, (B, 12)), mu=view(v.mu, np.float32, (B, h, 2)),
                        controls=view(v.controls, np.float64, (B, h, 12)), states=view(v.states, np.float64, (B, h, 13)),
                        iters=view(v.iters, np.int32, (B,)), residuals=view(v.residuals, np.float32, (B, 2)),
                        status=view(v.status, np.int32, (B,)), nfactor=view(v.nfactor, np.int32, (B,)))
        self._io_key = key
        return self._io

    def solve_inplace(self, x_fb, foot, contact, phase, x_cmd=None, mu=None, want_states=True):
        """Host arrays in, results IN the solver's own buffers: what a control loop that keeps its arrays wants.  The inputs are
        converted (fp32) straight into the handle's page-locked I/O block and cross PCIe in one copy; the batch goes out in up to
        three chunked launches on prioritised streams; the kernels' epilogues widen to the reference's fp64 (REF:300-304) and store
        `controls` and the per-instance counters straight into the block's host arrays, `states` are stored in HBM and follow by
        copy engine chunk by chunk -- except the last chunk's, which go the way of the controls -- so there is no unpacking pass
        (`bmpc_host_io` / `bmpc_solve_batch_io`, include/bmpc.h).  Returns (states | None, controls, info) as VIEWS of that block:
        their CONTENT is valid until the next `solve_inplace` of this solver (copy what must outlive it); the MEMORY stays valid
        as long as a view is held -- the views keep the native handle alive past `close()` / garbage collection of the solver
        (`_HandleOwner`).  Same values as `solve`, bit for bit."""
        h = self.h
        xf = np.asarray(x_fb)
        B = xf.size // 12
        io = self._io_views(B, x_cmd is not None, mu is not None, want_states)
        np.copyto(io["x_fb"], xf.reshape(B, 12), casting="same_kind")
        np.copyto(io["foot"], np.asarray(foot).reshape(B, 6), casting="same_kind")
        cc = np.asarray(contact)
        if cc.dtype == np.uint8 and cc.size == B * h * 2:          # (the fast path: one pass for the 0 / 1 check, one for the copy)
            if cc.size and cc.max() > 1:
                raise ValueError("contact entries must be 0 or 1")
            np.copyto(io["contact"], cc.reshape(B, h, 2))
        else:
            np.copyto(io["contact"], _contact_u8(cc, B, h))
        np.copyto(io["phase"], np.asarray(phase).reshape(B), casting="same_kind")
        if x_cmd is not None:
            np.copyto(io["x_cmd"], np.asarray(x_cmd).reshape(B, 12), casting="same_kind")
        if mu is not None:
            np.copyto(io["mu"], np.asarray(mu).reshape(B, h, 2), casting="same_kind")
        rc = self._lib.bmpc_solve_batch_io(self._h, B)
        if rc != 0:
            self._io, self._io_key = None, None    # (whatever went wrong: the next call lays the block out afresh)
            _lib.check(rc)
        info = dict(iters=io["iters"], status=io["status"], nfactor=io["nfactor"], residuals=io["residuals"])
        return io["states"], io["controls"], info

    def assemble(self, x_fb, foot, contact, phase, x_cmd=None, mu=None, want_matrices=True):
        """Assembly stage only (parity tests, reference generators): x_ref (B,h,12), foot_ref (B,h,6) and -- with
        `want_matrices`, dense family only (h <= 20) -- Gt (B,6h,6h), qt (B,6h), else None for both.  Without them the
        launch runs on the handle's own kernel family at every supported horizon and nothing of size (6h)^2 is allocated."""
        B, x_fb, foot, contact, phase, x_cmd, mu = self._marshal(x_fb, foot, contact, phase, x_cmd, mu)
        h = self.h
        x_ref = np.zeros((B, h, 12)); foot_ref = np.zeros((B, h, 6))
        Gt = np.zeros((B, 6 * h, 6 * h)) if want_matrices else None
        qt = np.zeros((B, 6 * h)) if want_matrices else None
        _lib.check(self._lib.bmpc_debug_assemble(
            self._h, B, _ptr(x_fb), _ptr(foot), _ptr(contact), _ptr(phase), _ptr(x_cmd), _ptr(mu),
            _ptr(x_ref), _ptr(foot_ref), _ptr(Gt), _ptr(qt)))
        return x_ref, foot_ref, Gt, qt

    # ---- device-resident (torch tensors are only a way to own HBM and a stream) ----------------
    def solve_device(self, x_fb, foot, contact, phase, x_cmd=None, mu=None, controls=None, states=None,
                     iters=None, residuals=None, status=None, nfactor=None, stream=None):
        """Inputs/outputs are CUDA(HIP) torch tensors on this solver's device (fp32 / uint8 / int32,
        contiguous).  Asynchronous on `stream` (default: torch's current stream).  Returns the
        output tensors; nothing crosses PCIe.  `stream` is a raw hipStream_t value; torch's default stream
        is HIP's null stream (value 0) and is passed on as such, so the launch is ordered against the
        surrounding torch work like any torch kernel."""
        import torch
        B = x_fb.shape[0]
        h = self.h
        dev = x_fb.device

        def chk(t, dtype, shape):
            if t is None:
                return 0
            if t.device != dev or t.dtype != dtype or not t.is_contiguous() or tuple(t.shape) != shape:
                raise ValueError(f"expected contiguous {dtype} tensor of shape {shape} on {dev}")
            return t.data_ptr()

        if dev.type != "cuda" or dev.index != self.device:
            raise ValueError(f"tensors must live on cuda:{self.device}")
        if controls is None:
            controls = torch.empty((B, h, 12), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(self._lib.bmpc_solve_batch_device(
            self._h, B, chk(x_fb, torch.float32, (B, 12)), chk(foot, torch.float32, (B, 6)),
            chk(contact, torch.uint8, (B, h, 2)), chk(phase, torch.int32, (B,)),
            chk(x_cmd, torch.float32, (B, 12)) or None, chk(mu, torch.float32, (B, h, 2)) or None,
            chk(controls, torch.float32, (B, h, 12)), chk(states, torch.float32, (B, h, 13)) or None,
            chk(iters, torch.int32, (B,)) or None, chk(residuals, torch.float32, (B, 2)) or None,
            chk(status, torch.int32, (B,)) or None, chk(nfactor, torch.int32, (B,)) or None, st))
        return controls, states

    # ---- the step either side of the solve ------------------------------------
    def foot_position_world(self, x_fb, q):
        """Batched REF:406-424 `getFootPositionWorld`: x_fb (B,12), q (B,10) -> pf_w (B,6) fp64."""
        x_fb = np.ascontiguousarray(np.asarray(x_fb, np.float32).reshape(-1, 12))
        B = x_fb.shape[0]
        q = np.ascontiguousarray(np.asarray(q, np.float32).reshape(B, 10))
        pf = np.empty((B, 6), np.float32)
        _lib.check(self._lib.bmpc_foot_position_world(self._h, B, _ptr(x_fb), _ptr(q), _ptr(pf)))
        return pf.astype(np.float64)

    def low_level_control(self, x_fb, t, pf_w, q, qd, contact0, u0):
        """Batched REF:444-470 `lowLevelControl`: contact0 (B,2) = contact[0, 0:2], u0 (B,12) = controls[0]
        -> tau (B,10) fp64."""
        x_fb = np.ascontiguousarray(np.asarray(x_fb, np.float32).reshape(-1, 12))
        B = x_fb.shape[0]
        t = np.ascontiguousarray(np.asarray(t, np.float64).reshape(B))
        pf_w = np.ascontiguousarray(np.asarray(pf_w, np.float32).reshape(B, 6))
        q = np.ascontiguousarray(np.asarray(q, np.float32).reshape(B, 10))
        qd = np.ascontiguousarray(np.asarray(qd, np.float32).reshape(B, 10))
        c0 = _contact_u8(np.asarray(contact0).reshape(B, 1, 2), B, 1).reshape(B, 2)
        u0 = np.ascontiguousarray(np.asarray(u0, np.float32).reshape(B, 12))
        tau = np.empty((B, 10), np.float32)
        _lib.check(self._lib.bmpc_low_level_control(self._h, B, _ptr(x_fb), _ptr(t), _ptr(pf_w), _ptr(q), _ptr(qd),
                                                    _ptr(c0), _ptr(u0), _ptr(tau)))
        return tau.astype(np.float64)

    def _gait(self, period, offset, duty):
        """`bmpc_gait` for a schedule that departs from this solver's default (None: the default)."""
        if period is None and offset is None and duty is None:
            return None
        gait = _lib.CGait()
        _lib.check(self._lib.bmpc_gait_default(C.byref(gait), int(self.cparams.half)))
        if period is not None:
            gait.period = int(period)
        if offset is not None:
            gait.offset[0], gait.offset[1] = int(offset[0]), int(offset[1])
        if duty is not None:
            gait.duty[0], gait.duty[1] = int(duty[0]), int(duty[1])
        return gait

    def contact_sequence(self, t, period=None, offset=None, duty=None, want_contact=True):
        """Batched gait scheduler on the device (REF:50-59 and the phase index of REF:99-100).
        t (B,) fp64 -> phase (B,) int32, contact (B,h,2) uint8.  Default schedule: the reference's, at this
        solver's half period; otherwise leg g stands at schedule step n iff ((n + offset[g]) % period) < duty[g]."""
        t = np.ascontiguousarray(np.asarray(t, np.float64).reshape(-1))
        B = t.shape[0]
        gait = self._gait(period, offset, duty)
        phase = np.empty(B, np.int32)
        contact = np.empty((B, self.h, 2), np.uint8) if want_contact else None
        _lib.check(self._lib.bmpc_contact_sequence(self._h, B, _ptr(t), None if gait is None else C.byref(gait),
                                                   _ptr(phase), _ptr(contact)))
        return phase, contact

    def contact_sequence_device(self, t, phase=None, contact=None, period=None, offset=None, duty=None, stream=None):
        """`contact_sequence` on device tensors: t (B,) float64 CUDA tensor -> (phase int32 (B,), contact uint8
        (B,h,2)), asynchronous on `stream` (default: torch's current stream); nothing crosses PCIe."""
        import torch
        dev = t.device
        if dev.type != "cuda" or dev.index != self.device or t.dtype != torch.float64 or not t.is_contiguous():
            raise ValueError(f"t must be a contiguous float64 tensor on cuda:{self.device}")
        B = t.shape[0]
        if phase is None:
            phase = torch.empty(B, dtype=torch.int32, device=dev)
        if contact is None:
            contact = torch.empty((B, self.h, 2), dtype=torch.uint8, device=dev)
        gait = self._gait(period, offset, duty)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(self._lib.bmpc_contact_sequence_device(self._h, B, t.data_ptr(), None if gait is None else C.byref(gait),
                                                          phase.data_ptr(), contact.data_ptr(), st))
        return phase, contact

    # ---- receding-horizon use --------------------------------------------------
    def set_warm_start(self, enable=True, shift=0, theta=0.5):
        """Later solves of the same batch size start from the state the previous solve left on the device
        (`bmpc_set_warm_start`): `shift` horizon steps later, penalties pulled back by rho0 (rho / rho0)^theta."""
        _lib.check(self._lib.bmpc_set_warm_start(self._h, 1 if enable else 0, int(shift), float(theta)))

    def reset_warm_start(self):
        _lib.check(self._lib.bmpc_reset_warm_start(self._h))

    def set_dispatch_order(self, order=None, longest_first_rollouts=True):
        """`bmpc_set_dispatch_order`: `order` is None or an int32 CUDA(HIP) tensor holding a permutation of 0 .. B-1
        (workgroup g solves instance order[g]; keep the tensor alive while it is set); `longest_first_rollouts`:
        roll-outs dispatch each period's instances by descending iteration count of the period before.  Results never
        depend on the order, only the time a batch takes."""
        import torch
        ptr = None
        if order is not None:
            if not (isinstance(order, torch.Tensor) and order.is_cuda and order.dtype == torch.int32 and order.is_contiguous()):
                raise ValueError("order must be a contiguous int32 tensor on the solver's device")
            ptr = order.data_ptr()
        self._order_keepalive = order
        _lib.check(self._lib.bmpc_set_dispatch_order(self._h, ptr, 1 if longest_first_rollouts else 0))

    def rollout_device(self, x_fb, foot, t, steps, x_cmd=None, mu=None, period=None, offset=None, duty=None,
                       want_iters=True, stream=None):
        """`steps` closed-loop control periods on device tensors (`bmpc_rollout_device`): x_fb (B,12) float32 and
        t (B,) float64 are advanced IN PLACE; returns dict(u0 (steps,B,12), x (steps,B,12), iters (steps,B) | None,
        status_any (B,)).  Asynchronous on `stream` (default: torch's current stream)."""
        import torch
        dev = x_fb.device
        B = x_fb.shape[0]
        if dev.type != "cuda" or dev.index != self.device:
            raise ValueError(f"tensors must live on cuda:{self.device}")

        def chk(tn, dtype, shape):
            if tn is None:
                return None
            if tn.device != dev or tn.dtype != dtype or not tn.is_contiguous() or tuple(tn.shape) != shape:
                raise ValueError(f"expected contiguous {dtype} tensor of shape {shape} on {dev}")
            return tn.data_ptr()

        gait = self._gait(period, offset, duty)
        u0 = torch.empty((steps, B, 12), dtype=torch.float32, device=dev)
        xt = torch.empty((steps, B, 12), dtype=torch.float32, device=dev)
        its = torch.empty((steps, B), dtype=torch.int32, device=dev) if want_iters else None
        st_any = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(self._lib.bmpc_rollout_device(
            self._h, B, int(steps), chk(x_fb, torch.float32, (B, 12)), chk(foot, torch.float32, (B, 6)),
            chk(t, torch.float64, (B,)), None if gait is None else C.byref(gait), chk(x_cmd, torch.float32, (B, 12)),
            chk(mu, torch.float32, (B, self.h, 2)), u0.data_ptr(), xt.data_ptr(), None if its is None else its.data_ptr(),
            st_any.data_ptr(), st))
        return dict(u0=u0, x=xt, iters=its, status_any=st_any)

    def last_kernel_ms(self):
        ms = C.c_float(-1.0)
        _lib.check(self._lib.bmpc_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def synchronize(self):
        _lib.check(self._lib.bmpc_synchronize(self._h))


class SolverStatusWarning(RuntimeWarning):
    """Some instances stopped at the iteration cap (status 1): their controls are the last iterate."""


def _check_status(info, where):
    """The reference never looks at its solver's status (REF:297-300); a drop-in that silently hands a
    non-converged or non-finite iterate to `lowLevelControl` would be worse than that.  Status 2 (NaN/Inf
    iterate: non-finite inputs) is an error, status 1 (iteration cap) a warning.  Callers who want to
    handle it themselves pass `return_info=True`."""
    import warnings
    status = info["status"]
    nbad = int((status == 2).sum())
    if nbad:
        raise FloatingPointError(f"{where}: {nbad} of {status.size} instances produced a NaN/Inf iterate "
                                 f"(status 2; first at index {int(np.flatnonzero(status == 2)[0])})")
    ncap = int((status == 1).sum())
    if ncap:
        warnings.warn(f"{where}: {ncap} of {status.size} instances stopped at the iteration cap "
                      f"(status 1); their controls are the last iterate", SolverStatusWarning, stacklevel=3)


_SOLVERS = {}          # (h, device) -> BatchSolver: ONE handle per horizon and device, whatever the parameters


def _cached_solver(mpc, biped, half, device, solver_options):
    """The drop-in wrappers keep one handle per (horizon, device) and push a changed parameter block
    through `bmpc_set_params`; a control loop that varies `mpc.x_cmd`, `Q`, `mu` ... from step to step
    therefore never creates a second handle (stream + events + staging buffers)."""
    cp = pack_params(mpc, biped, half, solver_options)
    key = (int(cp.h), int(device))
    s = _SOLVERS.get(key)
    if s is None:
        s = BatchSolver(cparams=cp, device=device)
        _SOLVERS[key] = s
    elif params_key(cp) != params_key(s.cparams):
        s.set_params(cp)
    return s


def close_cached_solvers():
    """Destroy the handles the drop-in wrappers keep (tests; orderly shutdown)."""
    for s in _SOLVERS.values():
        s.close()
    _SOLVERS.clear()


def solve_mpc_batch(x_fb, t, foot, contact, mpc=None, biped=None, x_cmd=None, mu=None, phase=None, half=None,
                    device=0, solver_options=None, return_info=False):
    """B instances of REF:187 `solve_mpc`.  x_fb (B,12), t (B,) seconds [or phase (B,) directly],
    foot (B,6), contact (B,h,2); optional per-instance x_cmd (B,12) and mu (B,h,2).
    Returns states (B,h,13), controls (B,h,12) [, info].  Without `return_info` a NaN/Inf instance raises
    FloatingPointError and instances stopped at the iteration cap raise a SolverStatusWarning."""
    from .params import MPC
    mpc = mpc if mpc is not None else MPC()
    solver = _cached_solver(mpc, biped, half, device, solver_options)
    if phase is None:
        phase = phase_indices(t, mpc.dt, mpc.h)
    states, controls, info = solver.solve(x_fb, foot, contact, phase, x_cmd=x_cmd, mu=mu)
    if return_info:
        return states, controls, info
    _check_status(info, "solve_mpc")
    return states, controls


def solve_mpc(x_fb, t, foot, mpc, biped, contact, half=None, device=0, solver_options=None):
    """Drop-in for REF:187-304: same arguments, same return shapes and dtypes (fp64 `states (h,13)`,
    `controls (h,12)`), inputs not mutated, silent."""
    h = int(mpc.h)
    contact = np.asarray(contact)
    if contact.ndim != 2 or contact.shape[1] != 2 or contact.shape[0] < h:
        raise ValueError(f"contact must have at least {h} rows of 2 (REF:239-249 indexes contact[k] for k < h)")
    x_fb = np.asarray(x_fb, float).reshape(12)
    foot = np.asarray(foot, float).reshape(6)
    states, controls = solve_mpc_batch(x_fb[None], [t], foot[None], contact[None, :h, :], mpc=mpc, biped=biped,
                                       half=half, device=device, solver_options=solver_options)
    return states[0], controls[0]


def reference_trajectories_batch(x_fb, t, foot, contact, mpc=None, biped=None, x_cmd=None, phase=None, half=None, device=0):
    """Batched REF:61-109 on the device (the generators phase 1 of the solve kernel runs, exposed through
    `bmpc_debug_assemble`): x_ref (B,13,h) and foot_ref (B,6,h), fp64, in the reference's row/column order."""
    from .params import MPC
    mpc = mpc if mpc is not None else MPC()
    solver = _cached_solver(mpc, biped, half, device, None)
    x_fb = np.asarray(x_fb, float).reshape(-1, 12)
    if phase is None:
        phase = phase_indices(t, mpc.dt, mpc.h)
    x_ref, foot_ref, _, _ = solver.assemble(x_fb, foot, contact, phase, x_cmd=x_cmd, want_matrices=False)
    B, h = x_ref.shape[0], x_ref.shape[1]
    xr = np.concatenate([x_ref.transpose(0, 2, 1), np.ones((B, 1, h))], axis=1)       # REF:62: 13th row of ones
    return xr, foot_ref.transpose(0, 2, 1)


def get_reference_trajectory(x_fb, mpc, device=0):
    """Drop-in for REF:61-70: returns x_ref (13,h) fp64."""
    h = int(mpc.h)
    xr, _ = reference_trajectories_batch(np.asarray(x_fb, float).reshape(1, 12), [0.0], np.zeros((1, 6)),
                                         np.ones((1, h, 2), np.uint8), mpc=mpc, device=device)
    return xr[0]


def get_reference_foot_trajectory(x_fb, t, foot, mpc, contact, half=None, device=0):
    """Drop-in for REF:72-109: returns foot_ref (6,h) fp64 (`contact` as REF:102 reads it: its first row decides)."""
    h = int(mpc.h)
    contact = np.asarray(contact)
    # REF:102 reads contact[0, :] only, and REF:58 hands over ten rows whatever mpc.h is: row 0 is repeated h times
    c = np.broadcast_to(contact.reshape(-1, 2)[0:1], (h, 2))
    _, fr = reference_trajectories_batch(np.asarray(x_fb, float).reshape(1, 12), [float(t)], np.asarray(foot, float).reshape(1, 6),
                                         c[None], mpc=mpc, half=half, device=device)
    return fr[0]


def getFootPositionWorld(x_fb, q, biped, mpc=None, device=0):
    """Drop-in for REF:406-424: returns pf_w (6,1) fp64 like the reference."""
    from .params import MPC
    solver = _cached_solver(mpc if mpc is not None else MPC(), biped, None, device, None)
    return solver.foot_position_world(np.asarray(x_fb, float).reshape(1, 12), np.asarray(q, float).reshape(1, 10)).reshape(6, 1)


def lowLevelControl(x_fb, t, pf_w, q, qd, mpc, biped, contact, u, device=0):
    """Drop-in for REF:444-470: same arguments (u is the (12,1) column REF:493 builds), returns tau (10,1) fp64."""
    solver = _cached_solver(mpc, biped, None, device, None)
    contact = np.asarray(contact)
    tau = solver.low_level_control(np.asarray(x_fb, float).reshape(1, 12), [float(t)], np.asarray(pf_w, float).reshape(1, 6),
                                   np.asarray(q, float).reshape(1, 10), np.asarray(qd, float).reshape(1, 10),
                                   contact[0, 0:2].reshape(1, 2), np.asarray(u, float).reshape(1, 12))
    return tau.reshape(10, 1)
