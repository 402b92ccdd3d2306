"""GPU parity: the HIP path through the C ABI against the oracle's certified optima (golden
fixtures) and against the oracle itself on fresh seeded inputs.  Tolerance: 1e-4 relative force
error (north_star), measured as SURVEY 8(d) defines it."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


PATH_AUTO, PATH_DENSE, PATH_STAGE = 0, 1, 2


def _solver(h, half, biped=None, **opts):
    import biped_mpc_py_amd as bm
    mpc = bm.MPC()
    mpc.h = h
    return bm.BatchSolver(mpc=mpc, biped=biped, half=half, solver_options=opts or None), mpc


def _solve_fixture(d, h, half, **opts):
    """A batch fixture through `BatchSolver.solve`, one handle per group of instances generated with the same Biped bounds
    (REF:45-48; one group -- the defaults -- for every fixture but cfg_bounds_h10).  Returns (states, controls, info, solvers)."""
    import biped_mpc_py_amd as bm
    n = d["x_fb"].shape[0]
    mu = d["mu_steps"] if "mu_steps" in d.files and d["mu_steps"].size else None
    states, controls = np.empty((n, h, 13)), np.empty((n, h, 12))
    info = {k: np.empty(n, np.int32) for k in ("iters", "status", "nfactor")}
    solvers = []
    for idx, biped in util.bounds_groups(d, bm):
        solver, mpc = _solver(h, half, biped=biped, **opts)
        st, u, inf = solver.solve(d["x_fb"][idx], d["foot"][idx], d["contact"][idx], util.phases(d["t"][idx], mpc.dt, h),
                                  x_cmd=d["x_cmd"][idx], mu=None if mu is None else mu[idx])
        states[idx], controls[idx] = st, u
        for k in info:
            info[k][idx] = inf[k]
        solvers.append(solver)
    return states, controls, info, solvers


@pytest.mark.parametrize("name", list(util.BATCH_FIXTURES))
def test_golden_batches(name):
    h, half = util.BATCH_FIXTURES[name]
    d = util.load(name)
    states, controls, info, _ = _solve_fixture(d, h, half)
    e = util.rel_err(controls, d["controls"])
    es = util.rel_err(states, d["states"])
    print(name, "ctrl err max %.2e  state err max %.2e  iters mean %.1f max %d  nfactor mean %.1f" %
          (e.max(), es.max(), info["iters"].mean(), info["iters"].max(), info["nfactor"].mean()))
    assert (info["status"] == 0).all(), info["status"]
    assert e.max() <= util.REL_TOL
    assert es.max() <= util.REL_TOL


@pytest.mark.parametrize("name", ["known_standing", "known_walking_t0"])
def test_known_answers_dropin(name):
    """The reference's own call surface (REF:487, 493-494) on its two default cases."""
    import biped_mpc_py_amd as bm
    d = util.load(name)
    mpc, biped = bm.MPC(), bm.Biped()
    states, controls = bm.solve_mpc(d["x_fb"], float(d["t"]), d["foot"], mpc, biped, d["contact"])
    assert states.shape == (10, 13) and controls.shape == (10, 12)
    assert states.dtype == np.float64 and controls.dtype == np.float64
    assert util.rel_err(controls[None], d["controls"][None]).max() <= util.REL_TOL
    assert np.abs(states - d["states"]).max() <= 1e-4 * max(1.0, np.abs(d["states"]).max())
    u0 = controls[0, :].reshape(-1, 1)
    assert u0.shape == (12, 1)


def test_assembly_matches_model():
    """x_ref / foot_ref against the reference-pinned fixtures; Gt, qt against the fp64 model."""
    from oracle import ws_model as ws
    d = util.load("cfg4_walking_h10")
    solver, mpc = _solver(10, 5)
    ph = util.phases(d["t"], mpc.dt, 10)
    x_ref, foot_ref, Gt, qt = solver.assemble(d["x_fb"], d["foot"], d["contact"], ph, x_cmd=d["x_cmd"])
    assert np.abs(x_ref.transpose(0, 2, 1) - d["x_ref"][:, :12]).max() < 1e-6
    assert np.abs(foot_ref.transpose(0, 2, 1) - d["foot_ref"]).max() < 1e-6
    P = ws.Params()
    _, _, info = ws.solve_batch(P, d["x_fb"], d["foot"], d["contact"], ph, x_cmd=d["x_cmd"], dtype=np.float64,
                                iters=1, return_debug=True)
    assert np.abs(Gt - info["Gt"]).max() <= 2e-6 * np.abs(info["Gt"]).max()
    assert np.abs(qt - info["qt"]).max() <= 2e-6 * np.abs(info["qt"]).max()


@pytest.mark.parametrize("h,gait,seed,kw", [
    (10, "standing", 1, {}),
    (10, "mixed", 3, dict(vx_cmd=True)),
    (16, "walking", 2, dict(vx_cmd=True)),
    (20, "walking", 4, dict(vx_cmd=True, per_step_mu=True)),
])
def test_fresh_seeded_vs_oracle(h, gait, seed, kw):
    """Same seeded inputs through the HIP path and through the oracle (sizes the oracle does in seconds)."""
    from oracle import bmpc_oracle as orc
    B = 24
    s = util.synth_batch(B, h, 1000 + seed, gait=gait, **kw)
    solver, mpc = _solver(h, s["half"])
    states, controls, info = solver.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"])
    ref_c = np.zeros_like(controls)
    for i in range(B):
        m, b = orc.MPC(), orc.Biped()
        m.h = h
        m.x_cmd = s["x_cmd"][i]
        t = s["phase"][i] * m.dt + 0.5 * m.dt
        x32 = s["x_fb"][i].astype(np.float32).astype(float)
        f32 = s["foot"][i].astype(np.float32).astype(float)
        mu_i = None if s["mu"] is None else s["mu"][i].astype(np.float32).astype(float)
        _, ref_c[i] = orc.solve_mpc(x32, t, f32, m, b, s["contact"][i], half=s["half"], mu_steps=mu_i)
    e = util.rel_err(controls, ref_c)
    print(h, gait, "err max %.2e  iters mean %.1f" % (e.max(), info["iters"].mean()))
    assert e.max() <= util.REL_TOL


def test_full_size_properties():
    """BASELINE config 2 size (B = 4096): size-independent properties -- feasibility of every
    constraint, consistency of states with the dynamics roll-out, determinism, batch invariance."""
    B, h = 4096, 10
    s = util.synth_batch(B, h, 1, gait="standing")
    solver, mpc = _solver(h, 5)
    states, controls, info = solver.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    assert (info["status"] == 0).all()
    f = controls.reshape(B, h, 4, 3)
    tol = 2e-3
    for j in range(2):
        fx, fy, fz = f[:, :, j, 0], f[:, :, j, 1], f[:, :, j, 2]
        assert (fz >= -tol).all() and (fz <= 500 + tol).all()
        assert (np.abs(fx) <= 0.5 * fz + tol).all() and (np.abs(fy) <= 0.5 * fz + tol).all()
        assert (np.abs(f[:, :, 2 + j, 0]) <= tol).all()                       # tau_max[0] = 0 (REF:47)
    # same inputs -> bitwise same outputs; a sub-batch gives the same rows
    _, c2, _ = solver.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    assert np.array_equal(controls, c2)
    _, c3, _ = solver.solve(s["x_fb"][100:164], s["foot"][100:164], s["contact"][100:164], s["phase"][100:164],
                            x_cmd=s["x_cmd"][100:164])
    assert np.array_equal(controls[100:164], c3)
    print("iters mean %.1f max %d, nfactor mean %.2f" % (info["iters"].mean(), info["iters"].max(), info["nfactor"].mean()))


@pytest.mark.parametrize("name,B,h,gait,seed,kw", [
    ("cfg3", 4096, 16, "walking", 2, dict(vx_cmd=True)),
    ("cfg4", 65536, 10, "mixed", 3, dict(vx_cmd=True)),
    ("cfg5_eighth", 8192, 20, "walking", 4, dict(vx_cmd=True, per_step_mu=True)),
])
def test_baseline_config_shapes_at_scale(name, B, h, gait, seed, kw):
    """BASELINE configs 3-5 at (or, for config 5, at one rank's share of) their full sizes: every
    instance converges and every constraint of REF:220-271 holds (size-independent properties)."""
    s = util.synth_batch(B, h, seed, gait=gait, **kw)
    solver, mpc = _solver(h, s["half"])
    solver.close()
    import biped_mpc_py_amd as bm
    solver = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    states, controls, info = solver.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"])
    assert (info["status"] == 0).all(), np.bincount(info["status"])
    assert not np.isnan(controls).any() and not np.isnan(states).any()
    mu = s["mu"] if s["mu"] is not None else np.full((B, h, 2), 0.5)
    f = controls.reshape(B, h, 4, 3)
    tol = 2e-3
    for j in range(2):
        c = s["contact"][:, :, j].astype(float)
        fx, fy, fz = f[:, :, j, 0], f[:, :, j, 1], f[:, :, j, 2]
        assert (fz >= -tol).all() and (fz <= 500 * c + tol).all()                      # REF:240-249
        assert (np.abs(fx) <= mu[:, :, j] * fz + tol).all() and (np.abs(fy) <= mu[:, :, j] * fz + tol).all()
        assert (np.abs(f[:, :, 2 + j, 0]) <= tol).all()                                 # tau_max[0] = 0
        assert (np.abs(f[:, :, 2 + j, 1]) <= 67 * c + tol).all() and (np.abs(f[:, :, 2 + j, 2]) <= 33.5 * c + tol).all()
    # dynamics consistency: v_z after step 1 from the first control row (REF:165-184, forward Euler)
    vz1 = s["x_fb"][:, 11].astype(np.float32) - 9.81 * 0.04 + 0.04 / 12 * (f[:, 0, 0, 2] + f[:, 0, 1, 2])
    assert np.abs(states[:, 0, 11] - vz1).max() <= 1e-4
    print(name, "iters mean %.1f max %d  nfactor mean %.2f max %d" %
          (info["iters"].mean(), info["iters"].max(), info["nfactor"].mean(), info["nfactor"].max()))
    solver.close()


def test_library_and_torch_share_one_hip_runtime():
    """libbmpc.so loaded BEFORE torch (the order build() -> smoke() produces) must not leave the process with
    two HIP runtimes (symptom: 'No HIP GPUs are available' / BMPC_ERR_NO_DEVICE).  Run in a child process."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from biped_mpc_py_amd import _lib; _lib.load()\n"
            "import torch; assert torch.cuda.is_available()\n"
            "import biped_mpc_py_amd as bm; s = bm.BatchSolver(max_batch=8)\n"
            "x = torch.zeros(4, device='cuda'); print('ok', float(x.sum()))\n") % util.ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_low_level_control_and_fk_on_device():
    """batched FK and force->torque map against the reference-pinned fixtures
    (unit_functions.npz: FK from the reference; known_*.npz: tau from the reference's lowLevelControl)
    and against the oracle restatement on random joint states.  Tolerance 2e-5 (fp32 I/O)."""
    import biped_mpc_py_amd as bm
    from oracle import bmpc_oracle as orc
    d = util.load("unit_functions")
    solver = bm.BatchSolver(max_batch=256)
    pf = solver.foot_position_world(d["xs"], d["qj"])
    assert np.abs(pf - d["fk"]).max() <= 2e-6
    for name in ("known_standing", "known_walking_t0"):
        k = util.load(name)
        mpc, biped = bm.MPC(), bm.Biped()
        pfw = bm.getFootPositionWorld(k["x_fb"], k["q_joint"], biped)
        assert pfw.shape == (6, 1) and np.abs(pfw - k["pf_w"]).max() <= 2e-6
        u0 = k["controls"][0, :].reshape(-1, 1)                         # REF:493
        tau = bm.lowLevelControl(k["x_fb"], float(k["t"]), k["pf_w"], k["q_joint"], k["qd_joint"], mpc, biped,
                                 k["contact"], u0)
        assert tau.shape == (10, 1)
        assert np.abs(tau - k["tau"]).max() <= 2e-5 * max(1.0, np.abs(k["tau"]).max())
    # random joint states, swing and stance legs, non-zero joint velocities and time
    rng = np.random.default_rng(7)
    B = 128
    xs = np.stack([util.synth_batch(1, 10, 100 + i)["x_fb"][0] for i in range(B)])
    q = rng.uniform(-1, 1, (B, 10)); qd = rng.uniform(-2, 2, (B, 10)); t = rng.uniform(0, 2, B)
    u0 = rng.uniform(-50, 150, (B, 12)); c0 = rng.integers(0, 2, (B, 2))
    pf = solver.foot_position_world(xs, q)
    tau = solver.low_level_control(xs, t, pf, q, qd, c0, u0)
    m, b = orc.MPC(), orc.Biped()
    for i in range(B):
        x32 = xs[i].astype(np.float32).astype(float)
        ref_pf = orc.getFootPositionWorld(x32, q[i].astype(np.float32).astype(float), b)
        assert np.abs(pf[i] - ref_pf.reshape(-1)).max() <= 2e-6
        ref = orc.lowLevelControl(x32, float(t[i]), pf[i].astype(np.float32).astype(float).reshape(6, 1),
                                  q[i].astype(np.float32).astype(float), qd[i].astype(np.float32).astype(float), m, b,
                                  np.tile(c0[i], (10, 1)), u0[i].astype(np.float32).astype(float).reshape(12, 1))
        assert np.abs(tau[i] - ref.reshape(-1)).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_edge_inputs_and_parameter_changes():
    """Empty batch, batch of one, NaN input (reported per instance, batch not failed), parameter update
    through bmpc_set_params semantics (a new solver with different mu / x_cmd changes the answer consistently)."""
    import biped_mpc_py_amd as bm
    from oracle import bmpc_oracle as orc
    d = util.load("cfg2_standing_h10")
    solver = bm.BatchSolver(max_batch=16)
    st, ct, info = solver.solve(np.zeros((0, 12)), np.zeros((0, 6)), np.zeros((0, 10, 2), np.uint8), np.zeros(0, np.int32))
    assert ct.shape == (0, 10, 12) and st.shape == (0, 10, 13)
    x = d["x_fb"][:4].copy()
    x[2, 5] = np.nan
    st, ct, info = solver.solve(x, d["foot"][:4], d["contact"][:4], np.zeros(4, np.int32))
    assert info["status"][2] == 2 and (info["status"][[0, 1, 3]] == 0).all()
    assert util.rel_err(ct[[0, 1, 3]], d["controls"][[0, 1, 3]]).max() <= util.REL_TOL
    with pytest.raises(ValueError):
        solver.solve(d["x_fb"][:2], d["foot"][:2], np.full((2, 10, 2), 2), np.zeros(2, np.int32))
    from biped_mpc_py_amd._lib import BmpcError
    with pytest.raises(BmpcError):
        solver.solve(d["x_fb"][:32], d["foot"][:32], d["contact"][:32], np.zeros(32, np.int32))   # > max_batch
    # different friction and command -> the oracle with the same parameters agrees
    mpc, biped = bm.MPC(), bm.Biped()
    biped.mu = 0.3
    mpc.x_cmd = np.array([0, 0, 0, 0.05, 0, 0.52, 0, 0, 0, 0.2, 0, 0], float)
    s2 = bm.BatchSolver(mpc=mpc, biped=biped, max_batch=4)
    _, c2, i2 = s2.solve(d["x_fb"][:2], d["foot"][:2], d["contact"][:2], np.zeros(2, np.int32))
    om, ob = orc.MPC(), orc.Biped()
    ob.mu, om.x_cmd = 0.3, mpc.x_cmd
    for i in range(2):
        _, ref = orc.solve_mpc(d["x_fb"][i].astype(np.float32).astype(float), 0.0, d["foot"][i].astype(np.float32).astype(float),
                               om, ob, d["contact"][i])
        assert util.rel_err(c2[i][None], ref[None]).max() <= util.REL_TOL


def test_gait_scheduler_on_device():
    """batched get_contact_sequence / phase index (REF:50-59, 99-100) on the device, against
    the tables captured from the reference, against the host mirror (Python's float floor division), and a
    general periodic schedule against its definition."""
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import api

    # 1. the reference's own tables (captured by oracle/gen_golden.py)
    d = util.load("unit_functions")
    mpc = bm.MPC()
    s = bm.BatchSolver(mpc=mpc, max_batch=8192)
    t_ref = np.asarray(d["t_list"], float)
    phase, contact = s.contact_sequence(t_ref)
    for i, t in enumerate(t_ref):
        assert np.array_equal(contact[i], np.asarray(d["contact_seq"][i], np.uint8))
        assert phase[i] == api.phase_index(float(t), mpc)
    # 2. times on and next to the step boundaries, where t // dt is decided by rounding
    rng = np.random.default_rng(5)
    k = rng.integers(0, 4000, 3000)
    t = np.concatenate([k * mpc.dt, np.nextafter(k * mpc.dt, np.inf), np.nextafter(k * mpc.dt, -np.inf),
                        (k * 0.01) * 4.0, rng.uniform(0, 200, 3000), [0.0, 0.12, 0.28, 0.36, 1e-300]])[:8192]
    phase, contact = s.contact_sequence(t)
    want_phase = np.array([api.phase_index(float(x), mpc) for x in t])
    assert np.array_equal(phase, want_phase)
    for i in rng.integers(0, len(t), 200):
        assert np.array_equal(contact[i], api.get_contact_sequence(float(t[i]), mpc).astype(np.uint8))
    s.close()
    # 3. other horizons / half periods (BASELINE configs 3, 5) against the host mirror
    for h, half in ((16, 8), (20, 5)):
        m2 = bm.MPC()
        m2.h = h
        s2 = bm.BatchSolver(mpc=m2, half=half, max_batch=512)
        t2 = rng.uniform(0, 50, 512)
        ph2, c2 = s2.contact_sequence(t2)
        for i in range(0, 512, 7):
            assert ph2[i] == api.phase_index(float(t2[i]), m2)
            assert np.array_equal(c2[i], api.get_contact_sequence(float(t2[i]), m2, half=half).astype(np.uint8))
        # 4. a general schedule: period 12, leg offsets 0 / 7, duties 9 / 4 (overlap and flight phases)
        ph3, c3 = s2.contact_sequence(t2, period=12, offset=(0, 7), duty=(9, 4))
        n = ph3[:, None] + np.arange(h)[None, :]
        want = np.stack([((n + 0) % 12) < 9, ((n + 7) % 12) < 4], axis=2).astype(np.uint8)
        assert np.array_equal(ph3, ph2) and np.array_equal(c3, want)
        with pytest.raises(bm.BmpcError):
            s2.contact_sequence(t2, period=12, duty=(13, 4))
        s2.close()


def test_device_resident_control_step():
    """One control step without host arithmetic: t -> (phase, contact) -> solve -> states[:, 0] on device tensors,
    against the host-pointer path on the same inputs; from rest the loop holds F_z = m g."""
    import torch
    import biped_mpc_py_amd as bm
    mpc, biped = bm.MPC(), bm.Biped()
    B = 64
    dev = torch.device("cuda", 0)
    s = bm.BatchSolver(mpc=mpc, biped=biped, max_batch=B)
    x0 = np.zeros((B, 12), np.float32)
    x0[:, 5] = 0.55
    foot = np.tile(np.array([-0.0195, 0.089, 0, -0.0195, -0.089, 0], np.float32), (B, 1))
    t_host = np.linspace(0.0, 2.0, B)
    t = torch.from_numpy(t_host).to(dev)
    x_fb, foot_t = torch.from_numpy(x0).to(dev), torch.from_numpy(foot).to(dev)
    states = torch.empty((B, mpc.h, 13), dtype=torch.float32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    phase, contact = s.contact_sequence_device(t, period=10, duty=(10, 10))       # both legs in stance
    assert contact.dtype == torch.uint8 and bool((contact == 1).all())
    ph_h, _ = s.contact_sequence(t_host, want_contact=False)
    assert np.array_equal(phase.cpu().numpy(), ph_h)
    for _ in range(8):
        controls, _ = s.solve_device(x_fb, foot_t, contact, phase, states=states, status=status)
        x_fb = states[:, 0, :12].contiguous()
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0
    u = controls.cpu().numpy()
    assert np.abs(u[:, 0, 2] + u[:, 0, 5] - biped.m * biped.g).max() < 0.3       # holds the weight (transient of the fp64 oracle loop: 0.1 N at this step)
    assert np.abs(x_fb.cpu().numpy()[:, 5] - 0.55).max() < 1e-3
    # same inputs through the host-pointer path
    _, u_h, info = s.solve(x_fb.cpu().numpy(), foot, contact.cpu().numpy(), phase.cpu().numpy(), want_states=False)
    c2, _ = s.solve_device(x_fb, foot_t, contact, phase)
    torch.cuda.synchronize()
    assert np.array_equal(c2.cpu().numpy().astype(np.float64), u_h)
    s.close()


@pytest.mark.parametrize("name", ["big_stand10", "big_mixed10", "big_walk16", "big_walk20"])
def test_oracle_solved_sets(name):
    """768 random instances of the BASELINE config shapes solved by the fp64 oracle (tests/gen_tuning_sets.py):
    every instance converges and matches to the north_star tolerance."""
    import os
    import biped_mpc_py_amd as bm
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tuning", name + ".npz"))
    mpc = bm.MPC()
    mpc.h = int(z["h"])
    s = bm.BatchSolver(mpc=mpc, half=int(z["half"]), max_batch=len(z["x_fb"]))
    mu = z["mu"] if z["mu"].size else None
    _, u, info = s.solve(z["x_fb"], z["foot"], z["contact"], z["phase"], x_cmd=z["x_cmd"], mu=mu, want_states=False)
    s.close()
    ref = z["ref"]
    rel = np.abs(u - ref).reshape(len(u), -1).max(1) / np.maximum(1.0, np.abs(ref).reshape(len(u), -1).max(1))
    assert int((info["status"] != 0).sum()) == 0
    assert rel.max() <= util.REL_TOL, rel.max()


@pytest.mark.parametrize("h,seed,kw", [(16, 306, dict(vx_cmd=True)), (20, 205, dict(vx_cmd=True, per_step_mu=True))])
def test_rare_active_set_cycles_are_damped(h, seed, kw):
    """Two batches in which one instance used to cycle between two active sets until max_iter (found by a
    2 M-instance soak at kappa = 20): with the damped late moves every instance converges in < 300 iterations."""
    import biped_mpc_py_amd as bm
    B = 16384
    s = util.synth_batch(B, h, seed, gait="walking", **kw)
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    sol.close()
    assert int((info["status"] != 0).sum()) == 0
    assert int(info["iters"].max()) < 300 and not np.isnan(u).any()


def _oracle_controls(s, idx, h, mpc_mod=None, biped_mod=None):
    """Oracle optimum for instances `idx` of a synth_batch dict, on the float32-rounded inputs the GPU sees."""
    from oracle import bmpc_oracle as orc
    out = []
    for i in idx:
        m, b = orc.MPC(), orc.Biped()
        m.h = h
        m.x_cmd = s["x_cmd"][i]
        if mpc_mod:
            mpc_mod(m)
        if biped_mod:
            biped_mod(b)
        mu_i = None if s["mu"] is None else s["mu"][i].astype(np.float32).astype(float)
        _, ct = orc.solve_mpc(s["x_fb"][i].astype(np.float32).astype(float), (s["phase"][i] + 0.5) * m.dt,
                              s["foot"][i].astype(np.float32).astype(float), m, b, s["contact"][i], half=s["half"], mu_steps=mu_i)
        out.append(ct)
    return np.stack(out)


@pytest.mark.parametrize("h,gait,seed,kw", [
    (10, "mixed", 501, dict(vx_cmd=True)), (10, "standing", 502, {}),
    (16, "walking", 503, dict(vx_cmd=True)), (20, "walking", 504, dict(vx_cmd=True, per_step_mu=True))])
def test_hardest_instances_vs_oracle(h, gait, seed, kw):
    """The HARDEST instances (most iterations) of a 16384 batch of every config shape against the fp64 oracle."""
    import biped_mpc_py_amd as bm
    B, n = 16384, 8
    s = util.synth_batch(B, h, seed, gait=gait, **kw)
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    sol.close()
    assert int((info["status"] != 0).sum()) == 0
    idx = np.argsort(-info["iters"], kind="stable")[:n]
    ref = _oracle_controls(s, idx, h)
    rel = util.rel_err(u[idx], ref)
    print("h=%d %s: hardest %d of %d (iterations %d..%d): max rel err %.2e" % (h, gait, n, B, info["iters"][idx].min(),
                                                                             info["iters"][idx].max(), rel.max()))
    assert rel.max() <= util.REL_TOL


@pytest.mark.parametrize("what", ["Q_x10", "Q_div10", "R_x10", "R_div10", "f_max_150", "R_mixed"])
def test_non_default_weights_and_bounds(what):
    """REF:278-286 / REF:45-48 with other numbers than the reference's defaults: the cost weights set the
    conditioning the penalty schedule was tuned around, so they are varied by a decade either way, and the
    force cap is lowered until it binds in double support.  Against the oracle with the same parameters."""
    import biped_mpc_py_amd as bm
    B = 12
    mods = {
        "Q_x10": (lambda m: setattr(m, "Q", np.asarray(m.Q, float) * 10.0), None),
        "Q_div10": (lambda m: setattr(m, "Q", np.asarray(m.Q, float) / 10.0), None),
        "R_x10": (lambda m: setattr(m, "R", np.asarray(m.R, float) * 10.0), None),
        "R_div10": (lambda m: setattr(m, "R", np.asarray(m.R, float) / 10.0), None),
        "R_mixed": (lambda m: setattr(m, "R", np.asarray(m.R, float) * np.array([1, 3, 10, 1, 3, 10, 30, 1, 3, 30, 1, 3.0])), None),
        "f_max_150": (None, lambda b: setattr(b, "f_max", np.array([[150.0], [150.0], [55.0]]))),
    }
    mpc_mod, biped_mod = mods[what]
    for gait, seed, kw in (("standing", 41, {}), ("walking", 42, dict(vx_cmd=True))):
        s = util.synth_batch(B, 10, seed, gait=gait, **kw)
        mpc, biped = bm.MPC(), bm.Biped()
        if mpc_mod:
            mpc_mod(mpc)
        if biped_mod:
            biped_mod(biped)
        sol = bm.BatchSolver(mpc=mpc, biped=biped, half=s["half"], max_batch=B)
        _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], want_states=False)
        sol.close()
        ref = _oracle_controls(s, range(B), 10, mpc_mod, biped_mod)
        rel = util.rel_err(u, ref)
        print(what, gait, "err max %.2e iters mean %.1f max %d nfac %.1f" % (rel.max(), info["iters"].mean(), info["iters"].max(),
                                                                           info["nfactor"].mean()))
        assert int((info["status"] != 0).sum()) == 0
        assert rel.max() <= util.REL_TOL
        if what == "f_max_150" and gait == "standing":
            assert np.abs(ref[:, :, 2] - 55.0).min() < 1e-6          # the lowered cap really binds


def test_solve_device_is_ordered_on_the_default_stream():
    """`solve_device` without a stream runs on torch's current stream.  On the DEFAULT stream that is HIP's null
    stream (cuda_stream == 0): the launch must be ordered after a long-running torch producer of x_fb on that
    stream and before the torch consumer of the controls (it used to go to the handle's own
    non-blocking stream, unordered against both)."""
    import torch
    import biped_mpc_py_amd as bm
    B, h = 512, 10
    dev = torch.device("cuda", 0)
    s = util.synth_batch(B, h, 7)
    sol = bm.BatchSolver(max_batch=B)
    assert torch.cuda.current_stream(dev).cuda_stream == 0
    x_ok = torch.from_numpy(s["x_fb"].astype(np.float32)).to(dev)
    foot = torch.from_numpy(s["foot"].astype(np.float32)).to(dev)
    contact = torch.from_numpy(s["contact"]).to(dev)
    phase = torch.from_numpy(s["phase"]).to(dev)
    want, _ = sol.solve_device(x_ok, foot, contact, phase)
    torch.cuda.synchronize()
    want = want.clone()
    big = torch.randn(8192, 8192, device=dev)
    for _ in range(3):
        x_in = torch.full((B, 12), float("nan"), device=dev)        # a solve that ran too early would see NaNs
        acc = big
        for _ in range(6):                                           # ~tens of ms of work queued on the null stream
            acc = acc @ big * 1e-4
        x_in.copy_(x_ok + 0.0 * acc[:B, :12].nan_to_num(0.0, 0.0, 0.0))
        status = torch.empty(B, dtype=torch.int32, device=dev)
        got, _ = sol.solve_device(x_in, foot, contact, phase, status=status)
        out = got.clone()                                            # consumer on the same stream
        got.zero_()                                                  # and a later writer: must come after the kernel's stores
        torch.cuda.synchronize()
        assert int((status != 0).sum()) == 0
        assert torch.equal(out, want)
    sol.close()


@pytest.mark.parametrize("B,h,kw", [(4099, 10, dict(vx_cmd=True)), (1023, 10, {}), (5, 10, {}), (2051, 20, dict(vx_cmd=True, per_step_mu=True))])
def test_host_pointer_path_is_chunked_and_bit_identical(B, h, kw):
    """`bmpc_solve_batch` / `bmpc_solve_batch_f64` (what REF:487 callers get): pinned staging, up to three chunks on streams of
    descending priority, each chunk's results written to HBM and followed on its stream by one packed device-to-host copy into
    pinned memory, a chunk's unpacking / fp64 widening overlapped with the later chunks' solves; and `bmpc_host_io` /
    `bmpc_solve_batch_io` (round 5): the handle's page-locked I/O block, one copy in, up to three chunked launches, fp64 controls
    and counters stored by the kernels straight into the block's host arrays, states by copy engine per chunk (the last chunk's
    by the kernel).  The
    results must not depend on any of that: every entry against ONE launch of `bmpc_solve_batch_device` over the whole batch,
    bit for bit -- controls, states, iteration counts, status, residuals -- for ragged sizes (chunk boundaries off any power
    of two, fewer instances than a chunk), optional inputs, `want_states = False`, caller-owned output arrays, and with a
    dispatch order set (one chunk)."""
    import ctypes as C
    import torch
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import _lib
    s = util.synth_batch(B, h, 900 + B, gait="mixed" if h == 10 else "walking", **kw)
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    dev = torch.device("cuda:0")
    t = {k: (None if s[k] is None else torch.from_numpy(np.ascontiguousarray(s[k].astype(np.float32) if s[k].dtype == np.float64 else s[k])).to(dev))
         for k in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu")}
    o_s = torch.empty((B, h, 13), dtype=torch.float32, device=dev)
    o_it, o_st, o_nf = (torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3))
    o_rs = torch.empty((B, 2), dtype=torch.float32, device=dev)
    u_dev, _ = sol.solve_device(t["x_fb"], t["foot"], t["contact"], t["phase"], t["x_cmd"], t["mu"], states=o_s, iters=o_it,
                                residuals=o_rs, status=o_st, nfactor=o_nf)
    torch.cuda.synchronize()
    u1, s1 = u_dev.cpu().numpy(), o_s.cpu().numpy()
    # fp64 entry (BatchSolver.solve)
    st, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"])
    assert u.dtype == np.float64 and st.dtype == np.float64
    assert np.array_equal(u, u1.astype(np.float64)) and np.array_equal(st, s1.astype(np.float64))
    assert np.array_equal(info["iters"], o_it.cpu().numpy()) and np.array_equal(info["status"], o_st.cpu().numpy())
    assert np.array_equal(info["nfactor"], o_nf.cpu().numpy()) and np.array_equal(info["residuals"], o_rs.cpu().numpy())
    # without states, into caller-owned arrays
    buf = np.full((B, h, 12), np.nan)
    st2, u2, _ = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False, out=(None, buf))
    assert st2 is None and u2 is buf and np.array_equal(buf, u)
    # fp32 entry through ctypes
    x32 = [np.ascontiguousarray(s[k].astype(np.float32)) for k in ("x_fb", "foot")]
    c8, ph = np.ascontiguousarray(s["contact"]), np.ascontiguousarray(s["phase"])
    xc = None if s["x_cmd"] is None else np.ascontiguousarray(s["x_cmd"].astype(np.float32))
    mu32 = None if s["mu"] is None else np.ascontiguousarray(s["mu"].astype(np.float32))
    uf, sf = np.empty((B, h, 12), np.float32), np.empty((B, h, 13), np.float32)
    itf = np.empty(B, np.int32)
    P = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    _lib.check(sol._lib.bmpc_solve_batch(sol._h, B, P(x32[0]), P(x32[1]), P(c8), P(ph), P(xc), P(mu32), P(uf), P(sf), P(itf), None, None, None))
    assert np.array_equal(uf, u1) and np.array_equal(sf, s1) and np.array_equal(itf, info["iters"])
    # the chunked call spans its kernels with the handle's timing events: first chunk's start to last chunk's end
    ms_host = sol.last_kernel_ms()
    sol.solve_device(t["x_fb"], t["foot"], t["contact"], t["phase"], t["x_cmd"], t["mu"])
    torch.cuda.synchronize()
    ms_dev = sol.last_kernel_ms()
    assert 0.5 * ms_dev < ms_host < 3.0 * ms_dev + 0.5, (ms_host, ms_dev)
    # the handle's page-locked I/O block: results read in place, fp64, the same bits (twice: the block is reused)
    for _ in range(2):
        st_i, u_i, i_i = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"])
        assert u_i.dtype == np.float64 and st_i.dtype == np.float64 and u_i.shape == (B, h, 12) and st_i.shape == (B, h, 13)
        assert np.array_equal(u_i, u) and np.array_equal(st_i, st)
        assert np.array_equal(i_i["iters"], info["iters"]) and np.array_equal(i_i["status"], info["status"])
        assert np.array_equal(i_i["nfactor"], info["nfactor"]) and np.array_equal(i_i["residuals"], info["residuals"])
    st_j, u_j, _ = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    assert st_j is None and np.array_equal(u_j, u)
    # ... and through raw ctypes, the way INTEGRATION.md binds it
    v = _lib.CHostViews()
    _lib.check(sol._lib.bmpc_host_io(sol._h, B, int(xc is not None), int(mu32 is not None), 1, C.byref(v)))
    C.memmove(v.x_fb, x32[0].ctypes.data, x32[0].nbytes); C.memmove(v.foot, x32[1].ctypes.data, x32[1].nbytes)
    C.memmove(v.contact, c8.ctypes.data, c8.nbytes); C.memmove(v.phase, ph.ctypes.data, ph.nbytes)
    if xc is not None:
        C.memmove(v.x_cmd, xc.ctypes.data, xc.nbytes)
    if mu32 is not None:
        C.memmove(v.mu, mu32.ctypes.data, mu32.nbytes)
    _lib.check(sol._lib.bmpc_solve_batch_io(sol._h, B))
    u_c = np.frombuffer((C.c_char * (B * h * 12 * 8)).from_address(v.controls), np.float64).reshape(B, h, 12)
    assert np.array_equal(u_c, u)
    assert sol._lib.bmpc_solve_batch_io(sol._h, B + 1) != 0                 # (laid out for B: anything else is refused)
    # (the raw call re-laid the block: the wrapper sees the layout generation move and re-reads its views by itself)
    # a dispatch order indexes the whole batch: one chunk, same results
    order = torch.arange(B - 1, -1, -1, dtype=torch.int32, device=dev)
    sol.set_dispatch_order(order)
    _, u3, i3 = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    st4, u4, i4 = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"])   # (one chunk: states straight
    sol.set_dispatch_order(None)                                                                                     #  from the kernel too)
    assert np.array_equal(u3, u) and np.array_equal(i3["iters"], info["iters"])
    assert np.array_equal(u4, u) and np.array_equal(st4, st) and np.array_equal(i4["iters"], info["iters"])
    sol.close()


def test_io_block_edges():
    """`bmpc_host_io` / `bmpc_solve_batch_io` at the edges: one instance (no chunking), a layout larger than `max_batch` refused,
    a solve before any layout refused, a re-layout for another batch size (the views move: the wrapper re-reads them), and results
    equal to `solve` every time."""
    import ctypes as C
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import _lib
    sol = bm.BatchSolver(max_batch=600)
    assert sol._lib.bmpc_solve_batch_io(sol._h, 1) != 0                    # no layout yet
    v = _lib.CHostViews()
    assert sol._lib.bmpc_host_io(sol._h, 601, 0, 0, 1, C.byref(v)) != 0    # beyond max_batch
    assert sol._lib.bmpc_host_io(sol._h, 0, 0, 0, 1, C.byref(v)) != 0
    for B in (1, 600, 37):
        s = util.synth_batch(B, 10, 4100 + B, gait="mixed", vx_cmd=True)
        st, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
        st_i, u_i, i_i = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
        assert u_i.shape == (B, 10, 12) and np.array_equal(u_i, u) and np.array_equal(st_i, st)
        assert np.array_equal(i_i["iters"], info["iters"]) and (i_i["status"] == 0).all()
    # the reference's own single step through the in-place path (REF:475-487 inputs of the standing known answer)
    k = util.load("known_standing")
    st1, u1, _ = sol.solve_inplace(k["x_fb"][None], k["foot"][None], k["contact"][None], np.array([util.phases(np.array([float(k["t"])]), 0.04, 10)[0]]))
    assert util.rel_err(u1, k["controls"][None]).max() <= util.REL_TOL
    sol.close()


def test_io_views_outlive_the_solver_and_a_foreign_layout_call_is_noticed():
    """ADVICE r5.  (medium) The arrays `solve_inplace` returns are views of the handle's page-locked block, which `bmpc_destroy`
    frees: they keep the native handle alive (`api._HandleOwner`), so results held past `close()` -- or past the solver object
    itself, `BatchSolver(...).solve_inplace(...)` -- still read the results, not freed memory; the handle goes with the last
    view.  (low) The wrapper's cached views belong to ONE layout: a raw `bmpc_host_io` on the same handle with the same B and
    other flags moves the offsets; the layout generation (`bmpc_host_io_generation`, ABI 11) shows it and the next
    `solve_inplace` re-reads the views instead of writing its inputs at stale offsets."""
    import ctypes as C
    import gc
    import weakref
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import _lib
    s = util.synth_batch(700, 10, 4242, gait="mixed", vx_cmd=True)
    ref_sol = bm.BatchSolver(max_batch=700)
    st, u, info = ref_sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    # results of a temporary solver
    st_t, u_t, i_t = bm.BatchSolver(max_batch=700).solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    gc.collect()
    assert np.array_equal(u_t, u) and np.array_equal(st_t, st) and np.array_equal(i_t["iters"], info["iters"])
    # results held past close(); the handle is destroyed when the last view goes
    sol = bm.BatchSolver(max_batch=700)
    owner = weakref.ref(sol._owner)
    st_i, u_i, i_i = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    sol.close()
    gc.collect()
    assert owner() is not None and np.array_equal(u_i, u) and np.array_equal(st_i, st)
    with pytest.raises(Exception):
        sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])          # a closed solver refuses
    del st_i, u_i, i_i
    gc.collect()
    assert owner() is None
    # a foreign layout call between two in-place solves
    sol = bm.BatchSolver(max_batch=700)
    _, u_a, _ = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    assert np.array_equal(u_a, u)
    g0 = sol._lib.bmpc_host_io_generation(sol._h)
    v = _lib.CHostViews()
    assert sol._lib.bmpc_host_io(sol._h, 700, 0, 1, 0, C.byref(v)) == 0          # same B, other flags: other offsets
    assert sol._lib.bmpc_host_io_generation(sol._h) == g0 + 1
    st_b, u_b, _ = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    assert np.array_equal(u_b, u) and np.array_equal(st_b, st)
    assert sol._lib.bmpc_host_io(sol._h, 701, 0, 0, 1, C.byref(v)) != 0          # refused before anything moved: the layout stands
    assert sol._lib.bmpc_host_io_generation(sol._h) == g0 + 2
    sol.close()
    ref_sol.close()


def test_dropin_reuses_one_handle_and_reports_status():
    """The drop-in wrappers keep ONE handle per (horizon, device) however often the command changes (ADVICE r1:
    one handle per distinct parameter block leaked streams and buffers), answers follow the changed
    parameters, and a non-converged / non-finite result is not handed back silently."""
    import warnings
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import api
    from oracle import bmpc_oracle as orc
    d = util.load("known_standing")
    api.close_cached_solvers()
    mpc, biped = bm.MPC(), bm.Biped()
    for k in range(6):
        mpc.x_cmd = np.array([0, 0, 0, 0, 0, 0.50 + 0.01 * k, 0, 0, 0, 0.05 * k, 0, 0], float)
        _, ctrl = bm.solve_mpc(d["x_fb"], float(d["t"]), d["foot"], mpc, biped, d["contact"])
        om = orc.MPC()
        om.x_cmd = mpc.x_cmd
        _, ref = orc.solve_mpc(np.asarray(d["x_fb"], np.float32).astype(float), float(d["t"]),
                               np.asarray(d["foot"], np.float32).astype(float), om, orc.Biped(), d["contact"])
        assert util.rel_err(ctrl[None], ref[None]).max() <= util.REL_TOL
        assert len(api._SOLVERS) == 1
    # iteration cap -> warning, NaN input -> error; return_info hands the status over instead
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        bm.solve_mpc(d["x_fb"], float(d["t"]), d["foot"], mpc, biped, d["contact"], solver_options=dict(max_iter=5))
    assert any(issubclass(x.category, bm.SolverStatusWarning) for x in w)
    bad = np.array(d["x_fb"], float).copy()
    bad[4] = np.nan
    with pytest.raises(FloatingPointError):
        bm.solve_mpc(bad, float(d["t"]), d["foot"], mpc, biped, d["contact"])
    _, _, info = bm.solve_mpc_batch(bad[None], [float(d["t"])], np.asarray(d["foot"], float)[None], d["contact"][None, :10],
                                    mpc=mpc, biped=biped, return_info=True)
    assert info["status"][0] == 2
    assert len(api._SOLVERS) == 1
    api.close_cached_solvers()


def test_default_half_follows_horizon_on_device():
    """h = 16 without an explicit `half`: the reference-foot generator must use the half period of the contact
    table (8), not the reference's hard-coded 5."""
    import biped_mpc_py_amd as bm
    d = util.load("cfg3_trot_h16")
    mpc = bm.MPC()
    mpc.h = 16
    sol = bm.BatchSolver(mpc=mpc, max_batch=len(d["x_fb"]))          # no half
    _, u, info = sol.solve(d["x_fb"], d["foot"], d["contact"], util.phases(d["t"], mpc.dt, 16), x_cmd=d["x_cmd"], want_states=False)
    sol.close()
    assert util.rel_err(u, d["controls"]).max() <= util.REL_TOL


def test_bench_two_ranks_strong_scaling():
    """The N > 1 path end to end with the real kernel: `python bench.py --gpus 2` started bare (it spawns its
    ranks itself), config-4 strong scaling of ONE seeded batch, parameter broadcast + all_gather of the controls
    inside the timed region, gathered result bit-identical to the single-GPU solve.  Both ranks share this box's
    one GPU, so the collectives run over gloo here (RCCL refuses two ranks on one device)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--config", "4", "--scaling", "strong", "--total", "4099", "--backend", "gloo", "--share-device"],
                       env=env, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads([x for x in p.stdout.decode().splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["total"] == 4099
    assert "bit-identical" in line["config"]["gather_check"]
    assert line["config"]["not_converged"] == 0
    assert line["ranks"]["world_size_backend"] == 2 and line["ranks"]["backend"] == "gloo" and len(line["ranks"]["devices"]) == 2


def test_bench_bare_n_gpus_reports_the_north_star_partition_and_proves_its_rank_count():
    """What the driver runs at N > 1 -- bare `bench.py --gpus N --steps K --warmup W`, nothing else -- must carry, on ONE
    line: `value` = the weak-scaling number (N = 1 equals BENCH), a `strong` record = the north_star partition (ONE batch of
    config 4 sharded over the ranks, broadcast + solve + all_gather per step, gathered controls bit-identical to the
    single-GPU solve, solves/s), and in `ranks` the world size the PROCESS GROUP reports after init_process_group together
    with one device identity per rank.  Two ranks on this box's one GPU, so over gloo (`--backend gloo --share-device` are
    rehearsal flags; the driver's run uses nccl = RCCL and one GPU per rank); `--total` only shrinks the batch of the record."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--total", "8193", "--backend", "gloo", "--share-device"], env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads([x for x in p.stdout.decode().splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["baseline_config"] == 2
    assert line["config"]["total"] == 2 * 4096 and line["config"]["collectives_in_step"] == ["all_gather(controls)"]
    sr = line["strong"]
    assert sr["baseline_config"] == 4 and sr["total"] == 8193 and sr["scaling"] == "strong" and sr["value"] > 0
    assert sr["collectives_in_step"] == ["broadcast(params)", "all_gather(controls)"]
    assert "bit-identical" in sr["gather_check"] and sr["not_converged_rank0"] == 0 and len(sr["kernel_ms_per_rank"]) == 2
    rk = line["ranks"]
    assert rk["world_size_backend"] == 2 and rk["backend"] == "gloo" and len(rk["devices"]) == 2
    assert all(d.startswith("rank %d:" % i) and "uuid" in d for i, d in enumerate(rk["devices"]))
    assert rk["distinct_devices"] == 1                     # (both ranks of this rehearsal sit on the one GPU)
    # the contract's roofline keys: `frac` is the SURVEY 8(d) number again, `frac_survey_formula` the same under the stable key
    rf = line["roofline"]
    assert abs(rf["frac"] - rf["frac_survey_formula"]) < 1e-12 and rf["frac_executed"] < rf["frac"]


def test_bench_bare_five_ranks_ragged_shards_and_a_killed_rank():
    """The driver's N = 8 command rehearsed as far as this box allows: a GPU box admits six processes on its card (the test
    process is one), so FIVE ranks over gloo on the one GPU (`--backend gloo --share-device`), started bare like the driver
    does, strong record on a batch of 65539 instances -- five ragged shards (13108 x 4 + 13107) --, one device string per rank,
    gathered controls bit-identical to rank 0's solve of the whole batch.  (The 8-rank partition itself -- bounds, padding,
    gather -- is held on the CPU over gloo: tests/test_sharding_gloo.py::test_eight_rank_shard_and_gather.)  Then the same
    launch with one rank killed (SIGKILL) right after the process group is up: the launcher must come back non-zero in
    bounded time, print no JSON line and leave no rank behind."""
    import json
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "5", "--steps", "2", "--warmup", "1",
           "--total", "65539", "--backend", "gloo", "--share-device", "--cpu-sample", "0", "--batch", "1024"]
    p = subprocess.run(cmd, env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads([x for x in p.stdout.decode().splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 5 and line["scaling"] == "weak" and line["config"]["total"] == 5 * 1024
    sr = line["strong"]
    assert sr["baseline_config"] == 4 and sr["total"] == 65539 and sr["scaling"] == "strong" and sr["value"] > 0
    assert "bit-identical" in sr["gather_check"] and sr["not_converged_rank0"] == 0 and len(sr["kernel_ms_per_rank"]) == 5
    rk = line["ranks"]
    assert rk["world_size_backend"] == 5 and rk["backend"] == "gloo" and len(rk["devices"]) == 5
    assert all(d.startswith("rank %d:" % i) and "uuid" in d for i, d in enumerate(rk["devices"]))
    # one rank dies: the launcher takes the rest down
    t0 = time.time()
    q = subprocess.run(cmd, env=dict(env, BMPC_BENCH_KILL_RANK="3", BMPC_BENCH_LAUNCH_TIMEOUT="200"), capture_output=True, timeout=400)
    took = time.time() - t0
    print("killed rank 3: launcher exit code %d after %.0f s" % (q.returncode, took))
    assert q.returncode != 0 and took < 300
    assert not any(ln.startswith("{") for ln in q.stdout.decode("utf-8", "replace").splitlines())
    import psutil
    me = os.getpid()
    ancestors = {a.pid for a in psutil.Process(me).parents()}
    left = [a for a in psutil.process_iter(["pid", "cmdline", "name"])
            if a.info["pid"] != me and a.info["pid"] not in ancestors and a.info["cmdline"]
            and "python" in (a.info["name"] or "") and any(x.endswith("bench.py") for x in a.info["cmdline"][1:3])]
    assert not left, [(a.info["pid"], a.info["cmdline"]) for a in left]


def test_bench_collectives_over_rccl_single_rank_rehearsal():
    """The N > 1 code path of bench.py over RCCL itself, as far as one GPU allows: `--force-dist` initialises the nccl process
    group with ONE rank and runs every collective of the line for real -- the parameter broadcast, the asynchronous
    all_gather_into_tensor of the controls inside the timed step, all_gather_object of the device identities, the all_reduce
    of the timings, and the `strong` record's broadcast + all_gather + bit-exact gather check.  (Two ranks need two GPUs over
    RCCL; the two-rank tests above run over gloo.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1",
                        "--total", "8193", "--cpu-sample", "0", "--skip-host-path"], env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads([x for x in p.stdout.decode().splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collectives_in_step"] == ["all_gather(controls)"]
    rk = line["ranks"]
    assert rk["world_size_backend"] == 1 and rk["backend"] == "nccl" and rk["distinct_devices"] == 1 and "uuid" in rk["devices"][0]
    sr = line["strong"]
    assert sr["total"] == 8193 and "bit-identical" in sr["gather_check"] and sr["not_converged_rank0"] == 0


def test_bench_default_line_carries_the_contract_and_the_secondary_records():
    """The driver's N = 1 run (`python bench.py`, shortened here): the contract's keys, `roofline` and `cpu_baseline`, the parity
    of the timed batch, and the two secondary rates -- the host-pointer path and two batches in flight -- each bit-identical to
    the timed device-resident solve."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--cpu-sample", "64"],
                       env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = [x for x in p.stdout.decode().splitlines() if x.strip()]
    assert len(out) == 1                                  # ONE JSON line on stdout
    line = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["config"]["baseline_config"] == 2 and line["config"]["not_converged"] == 0
    rf = line["roofline"]
    assert rf["frac"] == rf["frac_survey_formula"] and 0 < rf["frac"] < 1 and rf["kernel_ms"] <= line["ms_per_step"] * 1.02
    assert rf["mfma_instructions_per_solve"] == 24 and rf["mfma_util"] > 0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert line["parity"]["max_rel_err_vs_oracle"] <= 1e-5 and line["parity"]["u0_max_rel_err"] <= 2e-5
    assert line["value_incl_pcie"]["bit_identical_to_device_path"] and line["value_incl_pcie"]["value"] > 0
    two = line["two_batches_in_flight"]
    assert two["bit_identical_to_one_at_a_time"] and two["streams"] == 2 and two["value"] > 0.9 * line["value"]
    od = line["ordered_dispatch"]                         # (longest first by the previous solve's counts: same results, not slower)
    assert od["bit_identical_to_batch_order"] and od["value"] > 0.95 * line["value"]


@pytest.mark.parametrize("gait", ["standing", "walking"])
def test_closed_loop_rollout_and_warm_start(gait):
    """K = 20 control periods on the device (`bmpc_rollout_device`: schedule -> solve -> state
    feedback, one stream, no host arithmetic) against the fp64 oracle's OWN closed loop on the same start states:
    applied control and state agree per period to the north_star tolerance, with and without the receding-horizon
    warm start (same optimum); the warm start needs fewer iterations."""
    import torch
    import biped_mpc_py_amd as bm
    from oracle import bmpc_oracle as orc
    B, K, h = 6, 20, 10
    dev = torch.device("cuda", 0)
    mpc, biped = bm.MPC(), bm.Biped()
    rng = np.random.default_rng(5)
    x0 = np.zeros((B, 12), np.float32)
    x0[:, 5] = 0.55 + rng.uniform(-0.02, 0.02, B)
    x0[:, 0:3] = rng.uniform(-0.03, 0.03, (B, 3))
    x0[:, 9:12] = rng.uniform(-0.05, 0.05, (B, 3))
    foot = np.tile(np.array([-0.0195, 0.089, 0, -0.0195, -0.089, 0], np.float32), (B, 1))
    t0 = rng.uniform(0.0, 0.4, B)
    duty = (10, 10) if gait == "standing" else None              # both legs in stance / the reference's schedule
    # oracle closed loop
    ref_u, ref_x = np.zeros((K, B, 12)), np.zeros((K, B, 12))
    for b in range(B):
        x, t = x0[b].astype(float), float(t0[b])
        for k in range(K):
            contact = np.ones((h, 2), int) if gait == "standing" else orc.get_contact_sequence(t, orc.MPC())
            st, ct = orc.solve_mpc(x, t, foot[b].astype(float), orc.MPC(), orc.Biped(), contact)
            ref_u[k, b], x = ct[0], st[0, :12]
            ref_x[k, b] = x
            t += mpc.dt
    out = {}
    for warm in (False, True):
        s = bm.BatchSolver(mpc=mpc, biped=biped, max_batch=B)
        if warm:
            s.set_warm_start(True, shift=(0 if gait == "standing" else 1), theta=0.5)
        x = torch.from_numpy(x0.copy()).to(dev)
        t = torch.from_numpy(t0.copy()).to(dev)
        r = s.rollout_device(x, torch.from_numpy(foot).to(dev), t, K, period=(10 if duty else None), duty=duty)
        torch.cuda.synchronize()
        assert int((r["status_any"] != 0).sum()) == 0
        u, xs = r["u0"].cpu().numpy().astype(float), r["x"].cpu().numpy().astype(float)
        eu = np.abs(u - ref_u).max(2) / np.maximum(1.0, np.abs(ref_u).max(2))
        ex = np.abs(xs - ref_x).max(2) / np.maximum(1.0, np.abs(ref_x).max(2))
        out[warm] = r["iters"].cpu().numpy()
        print(gait, "warm" if warm else "cold", "u0 err max %.2e  x err max %.2e  mean iters %.1f (periods 2..K: %.1f)" %
              (eu.max(), ex.max(), out[warm].mean(), out[warm][1:].mean()))
        assert eu.max() <= util.REL_TOL and ex.max() <= util.REL_TOL
        assert np.allclose(t.cpu().numpy(), t0 + K * mpc.dt)
        s.close()
    assert np.array_equal(out[True][0], out[False][0])             # the first period starts cold either way
    # measured: 0.60x in double support, 0.84x while the contact schedule advances every period -- of the
    # cold counts of round 4; the round-5 schedule took 20 % off a COLD solve (35.6 instead of 44.2 iterations here) and 4 % off a
    # warm one (25.4 instead of 26.5): 0.71x
    # (iteration counts do not depend on the box: the bound is the measured ratio + 0.03; it was 0.78 in round 5)
    assert out[True][1:].mean() < (0.745 if gait == "standing" else 0.94) * out[False][1:].mean()


def test_reference_generators_dropin():
    """`get_reference_trajectory` / `get_reference_foot_trajectory` (REF:61-109) as first-class
    device-backed functions with the reference's return shapes, against the vectors captured from the reference."""
    import biped_mpc_py_amd as bm
    mpc = bm.MPC()
    for name in ("known_standing", "known_walking_t0"):
        d = util.load(name)
        xr = bm.get_reference_trajectory(d["x_fb"], mpc)
        fr = bm.get_reference_foot_trajectory(d["x_fb"], float(d["t"]), d["foot"], mpc, d["contact"])
        assert xr.shape == (13, 10) and fr.shape == (6, 10) and xr.dtype == np.float64
        assert np.abs(xr - d["x_ref"]).max() < 1e-6 and np.abs(fr - d["foot_ref"]).max() < 1e-6
    d = util.load("cfg4_walking_h10")
    xr, fr = bm.reference_trajectories_batch(d["x_fb"], d["t"], d["foot"], d["contact"], mpc=mpc, x_cmd=d["x_cmd"])
    assert np.abs(xr - d["x_ref"]).max() < 1e-6 and np.abs(fr - d["foot_ref"]).max() < 1e-6


@pytest.mark.parametrize("h,path", [(10, PATH_STAGE), (16, PATH_STAGE), (32, PATH_AUTO), (40, PATH_AUTO), (7, PATH_AUTO), (15, PATH_AUTO),
                                    (33, PATH_AUTO), (3, PATH_AUTO)])
def test_reference_generators_at_every_horizon_and_on_the_stage_family(h, path):
    """`assemble` / `reference_trajectories_batch` (REF:61-109 on the device) at the horizons only the stage family solves,
    and through the stage kernel's own reference branch at h <= 20 (path = STAGE), against the oracle's generators with the
    same half period.  (Round 3 always asked for the Gt / qt views, which exist for h <= 20 only: the call failed beyond,
    and at h <= 20 it silently ran the dense kernel whatever the path.)  The views themselves are refused beyond h = 20."""
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd._lib import BmpcError
    from oracle import bmpc_oracle as orc
    B = 48
    s = util.synth_batch(B, h, 40 + h, gait="walking", vx_cmd=True)
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(path=path))
    assert sol._lib.bmpc_solver_path(sol._h) == PATH_STAGE
    x_ref, foot_ref, Gt, qt = sol.assemble(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], want_matrices=False)
    assert Gt is None and qt is None
    om = orc.MPC()
    om.h = h
    for i in range(B):
        om.x_cmd = s["x_cmd"][i].astype(np.float32).astype(float)
        xf, ft = s["x_fb"][i].astype(np.float32).astype(float), s["foot"][i].astype(np.float32).astype(float)
        xr = orc.get_reference_trajectory(xf, om)
        fr = orc.get_reference_foot_trajectory(xf, (s["phase"][i] + 0.5) * om.dt, ft, om, s["contact"][i], half=s["half"])
        assert np.abs(x_ref[i].T - xr[:12]).max() < 1e-6 and np.abs(foot_ref[i].T - fr).max() < 1e-6, i
    if h > 20 or h % 2:                         # (no dense kernel at this horizon -- odd ones since round 5: half = h // 2, the second
        with pytest.raises(BmpcError):          #  touch-down point kept to the end of the horizon)
            sol.assemble(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], want_matrices=True)
    else:                                       # the views of the dense family, whatever the handle's path
        _, _, Gt, qt = sol.assemble(s["x_fb"][:2], s["foot"][:2], s["contact"][:2], s["phase"][:2], x_cmd=s["x_cmd"][:2])
        # (the rows are accumulated in f32 since round 4 -- preconditioner data --, each by its own lane: symmetric to f32 rounding)
        assert Gt.shape == (2, 6 * h, 6 * h) and np.abs(Gt - Gt.transpose(0, 2, 1)).max() <= 2e-6 * np.abs(Gt).max()
    sol.close()
    # the drop-in wrappers (one instance, the handle cache) at a long horizon
    bm.close_cached_solvers()
    xr1 = bm.get_reference_trajectory(s["x_fb"][0], mpc)
    assert xr1.shape == (13, h) and np.abs(xr1[:12, 0] - s["x_fb"][0].astype(np.float32)).max() < 1e-6
    bm.close_cached_solvers()


@pytest.mark.parametrize("cfg,B", [(2, 4096), (3, 2048), (5, 2048)])
def test_results_do_not_change_from_run_to_run(cfg, B):
    """The same batch solved six times gives bit-identical outputs, at a size that puts two waves on every SIMD (that is
    where a workgroup barrier reached with an LDS store in flight showed: ~1 % of the instances of a 4096 batch changed
    between runs, some to NaN, before the wait in front of every barrier was written out -- bmpc::sync_workgroup), and
    every instance converges in every run."""
    import torch
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import synth
    c = synth.CONFIGS[cfg]
    s = synth.synth_batch(B, c["h"], c["seed"], gait=c["gait"], **c["kw"])
    mpc = bm.MPC()
    mpc.h = c["h"]
    solver = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    dev = torch.device("cuda:0")
    t = {k: (None if s[k] is None else torch.from_numpy(np.ascontiguousarray(
        s[k].astype(np.float32) if s[k].dtype == np.float64 else s[k])).to(dev)) for k in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu")}
    ref = None
    for rep in range(6):
        status = torch.empty(B, dtype=torch.int32, device=dev)
        states = torch.empty((B, c["h"], 13), dtype=torch.float32, device=dev)
        controls, _ = solver.solve_device(t["x_fb"], t["foot"], t["contact"], t["phase"], t["x_cmd"], t["mu"], states=states,
                                          status=status)
        torch.cuda.synchronize()
        out = (controls.cpu().numpy(), states.cpu().numpy())
        assert (status.cpu().numpy() == 0).all(), (rep, np.flatnonzero(status.cpu().numpy()))
        if ref is None:
            ref = out
        else:
            for a, b in zip(ref, out):
                assert np.array_equal(a, b), (rep, np.flatnonzero((a != b).any(axis=(1, 2)))[:10])


def test_rollout_longest_first_dispatch_same_results_less_time():
    """`bmpc_set_dispatch_order`: a roll-out that dispatches every period's instances by descending iteration count of
    the period before gives bit-identical trajectories (the order only decides which workgroup solves which
    instance) and spends less time per period at 4096 instances, where the last workgroups of a launch otherwise
    decide when it ends; a user-supplied permutation does the same for a plain solve."""
    import time
    import torch
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import synth
    B, K = 4096, 12
    dev = torch.device("cuda", 0)
    s0 = synth.synth_batch(B, 10, 1)
    x0 = s0["x_fb"].astype(np.float32)
    x0[:, 0:3] *= 0.2                                  # small tilts and velocities: the closed loop stays near standing
    x0[:, 6:12] *= 0.2
    foot = torch.from_numpy(s0["foot"].astype(np.float32)).to(dev)
    res = {}
    for lf in (False, True):
        s = bm.BatchSolver(max_batch=B)
        s.set_warm_start(True, shift=0, theta=0.5)
        s.set_dispatch_order(None, longest_first_rollouts=lf)
        best = None
        for rep in range(3):
            s.reset_warm_start()
            x = torch.from_numpy(x0.copy()).to(dev)
            t = torch.zeros(B, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            t_0 = time.perf_counter()
            r = s.rollout_device(x, foot, t, K, period=10, duty=(10, 10))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t_0) / K
            best = dt if best is None else min(best, dt)
        assert int((r["status_any"] != 0).sum()) == 0
        res[lf] = (r["u0"].cpu().numpy(), r["x"].cpu().numpy(), r["iters"].cpu().numpy(), best)
        s.close()
    assert np.array_equal(res[False][0], res[True][0]) and np.array_equal(res[False][1], res[True][1])
    assert np.array_equal(res[False][2], res[True][2])
    print("roll-out of %d instances, ms per control period: index order %.3f, longest first %.3f (iterations per solve %.1f)" %
          (B, 1e3 * res[False][3], 1e3 * res[True][3], res[True][2][1:].mean()))
    assert res[True][3] < 1.10 * res[False][3]          # (measured -3 %; the bound only guards against a regression)
    # plain solve with a user-supplied order
    s = bm.BatchSolver(max_batch=B)
    t = {k: torch.from_numpy(np.ascontiguousarray(s0[k].astype(np.float32) if s0[k].dtype == np.float64 else s0[k])).to(dev)
         for k in ("x_fb", "foot", "contact", "phase")}
    it = torch.empty(B, dtype=torch.int32, device=dev)
    nf = torch.empty(B, dtype=torch.int32, device=dev)

    def timed():
        ms = []
        for _ in range(7):
            c, _ = s.solve_device(t["x_fb"], t["foot"], t["contact"], t["phase"], iters=it, nfactor=nf)
            torch.cuda.synchronize()
            ms.append(s.last_kernel_ms())
        return c, float(np.median(ms[2:]))

    c0, ms0 = timed()
    cost = 36.0 + 33.8 * nf.float() + 3.2 * it.float()          # k cycles (profiles/r02_cfg2_phase_cycles.txt)
    order = torch.argsort(cost, descending=True, stable=True).to(torch.int32)
    s.set_dispatch_order(order)
    c1, ms1 = timed()
    assert torch.equal(c0, c1)
    print("plain solve of the same batch: index order %.3f ms, longest first (by its own measured cost) %.3f ms" % (ms0, ms1))
    assert ms1 < 1.10 * ms0             # (measured -1 % ... -9 % depending on the box; the bound only guards against a regression)
    s.set_dispatch_order(None)
    s.close()


# ------------------------------------------------------------------------------------------------------------------
# the stage-structured kernel family (bmpc_stage.hip), the horizon as a launch parameter
# ------------------------------------------------------------------------------------------------------------------


def _hgen(h, name="cfg_hgen"):
    d = util.load(name)
    return {k: d["h%d_%s" % (h, k)] for k in ("x_fb", "t", "foot", "contact", "x_cmd", "mu_steps", "controls", "states", "half")}


def test_every_horizon_is_supported():
    """REF:24: `h` is a plain field of MPC.  Every horizon in [1, 40] has a kernel (round 5: odd and short ones on the stage
    family, whose lane map takes any number of steps -- the rest are phantoms); the dense family has the even ones in [8, 20]."""
    from biped_mpc_py_amd import _lib
    lib = _lib.load()
    for h in range(0, 48):
        want = 1 if 1 <= h <= 40 else 0
        assert lib.bmpc_supported_horizon(h) == want, h
        assert lib.bmpc_supported_horizon_path(h, PATH_STAGE) == want, h
        assert lib.bmpc_supported_horizon_path(h, PATH_DENSE) == (1 if (8 <= h <= 20 and h % 2 == 0) else 0), h


@pytest.mark.parametrize("h", [1, 2, 3, 4, 5, 7, 9, 15, 21, 33])
def test_odd_and_short_horizons(h):
    """REF:24 takes any int.  Odd and short horizons (oracle-solved extension fixtures, 4 instances each: walking with half
    period max(1, h // 2) -- the second touch-down point kept to the end of the horizon --, commanded v_x, per-step friction) run on
    the stage-structured family: its lane map takes any number of steps (h = 1: ONE step, nine of the smallest variant's ten
    slots are phantoms; h = 21: five steps per lane; h = 33: two waves).  AUTO resolves to it."""
    g = _hgen(h, "cfg_hodd")
    for path in (PATH_AUTO, PATH_STAGE):
        solver, mpc = _solver(h, int(g["half"][0]), path=path)
        assert solver._lib.bmpc_solver_path(solver._h) == PATH_STAGE
        states, controls, info = solver.solve(g["x_fb"], g["foot"], g["contact"], util.phases(g["t"], mpc.dt, h),
                                              x_cmd=g["x_cmd"], mu=g["mu_steps"])
        e, es = util.rel_err(controls, g["controls"]), util.rel_err(states, g["states"])
        print("h=%d path %d: err %.2e / %.2e iters %s" % (h, path, e.max(), es.max(), info["iters"]))
        assert (info["status"] == 0).all()
        assert e.max() <= util.REL_TOL and es.max() <= util.REL_TOL
        solver.close()


@pytest.mark.parametrize("h", [1, 2, 3])
def test_tiny_horizons_through_every_entry_point(h):
    """REF:24 takes any int, h = 1 included (one step: REF:195-216 has a single dynamics row).  The tiny horizons through everything a
    caller can reach: the host-pointer solve and the in-place I/O block (bit-identical), the oracle, the reference generators, a
    warm-started closed-loop roll-out on the device (shift = 1 needs two steps) and the reference's own call at h = 1."""
    import torch
    import biped_mpc_py_amd as bm
    from oracle import bmpc_oracle as orc
    B = 64
    dev = torch.device("cuda", 0)
    s = util.synth_batch(B, h, 70 + h, gait="walking", vx_cmd=True)
    assert s["half"] == 1
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    assert sol._lib.bmpc_solver_path(sol._h) == PATH_STAGE
    st, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    st2, u2, _ = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"])
    assert (info["status"] == 0).all() and st.shape == (B, h, 13) and u.shape == (B, h, 12)
    assert np.array_equal(u, u2) and np.array_equal(st, st2)
    r32 = lambda v: v.astype(np.float32).astype(float)
    om = orc.MPC()
    om.h = h
    for i in range(6):
        om.x_cmd = r32(s["x_cmd"][i])
        t = (s["phase"][i] + 0.5) * om.dt
        so, co = orc.solve_mpc(r32(s["x_fb"][i]), t, r32(s["foot"][i]), om, orc.Biped(), s["contact"][i], half=s["half"])
        assert util.rel_err(u[i:i + 1], co[None]).max() <= 1e-5 and util.rel_err(st[i:i + 1], so[None]).max() <= 1e-5, i
    x_ref, foot_ref, _, _ = sol.assemble(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], want_matrices=False)
    for i in range(6):
        om.x_cmd = r32(s["x_cmd"][i])
        xr = orc.get_reference_trajectory(r32(s["x_fb"][i]), om)
        fr = orc.get_reference_foot_trajectory(r32(s["x_fb"][i]), (s["phase"][i] + 0.5) * om.dt, r32(s["foot"][i]), om, s["contact"][i], half=s["half"])
        assert np.abs(x_ref[i].T - xr[:12]).max() < 1e-6 and np.abs(foot_ref[i].T - fr).max() < 1e-6, i
    sol.set_warm_start(True, shift=min(1, h - 1), theta=0.5)
    xt = torch.from_numpy(s["x_fb"].astype(np.float32)).to(dev)
    ft = torch.from_numpy(s["foot"].astype(np.float32)).to(dev)
    tt = torch.from_numpy((s["phase"] + 0.5) * mpc.dt).to(dev)
    out = sol.rollout_device(xt, ft, tt, steps=6)
    torch.cuda.synchronize()
    assert int(out["status_any"].sum().item()) == 0 and bool(torch.isfinite(out["u0"]).all().item()) and bool(torch.isfinite(out["x"]).all().item())
    sol.close()
    if h == 1:                                   # the reference's own call, one instance
        bm.close_cached_solvers()
        x = np.array([0.01, -0.02, 0.03, 0.0, 0.0, 0.5, 0, 0, 0, 0.1, 0, 0.0])
        foot = np.array([-0.02, 0.09, 0, -0.02, -0.09, 0])
        stt, ctl = bm.solve_mpc(x, 0.02, foot, mpc, bm.Biped(), bm.get_contact_sequence(0.02, mpc))
        _, cto = orc.solve_mpc(r32(x), 0.02, r32(foot), _mpc_h(orc, 1), orc.Biped(), orc.get_contact_sequence(0.02, _mpc_h(orc, 1)))
        assert stt.shape == (1, 13) and ctl.shape == (1, 12) and util.rel_err(ctl[None], cto[None]).max() <= 1e-5
        bm.close_cached_solvers()


def _mpc_h(mod, h):
    m = mod.MPC()
    m.h = h
    return m


@pytest.mark.parametrize("name", ["cfg2_standing_h10", "cfg4_walking_h10", "edge_cases_h10", "cfg3_trot_h16", "cfg5_mu_h20",
                                  "cfg_cmd_h10", "cfg_bounds_h10", "cfg_h32", "cfg_h40"])
def test_stage_path_golden_batches(name):
    """The stage-structured kernels against the same certified optima as the dense ones (and the long-horizon
    extension fixtures only they can solve): <= 1e-4, every instance converged, and -- where both families exist --
    the SAME iteration counts: the outer method is shared, only the application of K^-1 differs."""
    d = util.load(name)
    h = int(d["hor"][0]) if "hor" in d.files else 10
    half = int(d["half"][0]) if "half" in d.files else 5
    states, controls, info, solvers = _solve_fixture(d, h, half, path=PATH_STAGE)
    assert all(s._lib.bmpc_solver_path(s._h) == PATH_STAGE for s in solvers)
    e, es = util.rel_err(controls, d["controls"]), util.rel_err(states, d["states"])
    print(name, "stage path: ctrl err max %.2e state err max %.2e iters mean %.1f max %d nfactor %.2f" %
          (e.max(), es.max(), info["iters"].mean(), info["iters"].max(), info["nfactor"].mean()))
    assert (info["status"] == 0).all(), info["status"]
    assert e.max() <= util.REL_TOL and es.max() <= util.REL_TOL
    if h <= 20:
        # the plain method (secant extrapolation off: it amplifies last-bit differences into different paths) on both families
        _, _, info, _ = _solve_fixture(d, h, half, path=PATH_STAGE, accel=0)
        _, _, info_d, dense = _solve_fixture(d, h, half, path=PATH_DENSE, accel=0)
        assert all(s._lib.bmpc_solver_path(s._h) == PATH_DENSE for s in dense)
        assert np.abs(info["iters"].astype(int) - info_d["iters"]).max() <= 10          # (f32 preconditioners differ in the last bits)
        assert abs(info["iters"].mean() - info_d["iters"].mean()) <= 2.0
    else:
        auto, _ = _solver(h, half)
        assert auto._lib.bmpc_solver_path(auto._h) == PATH_STAGE                      # AUTO: the only family for h > 20


@pytest.mark.parametrize("h", [8, 12, 14, 18, 22, 24, 26, 28, 30, 34, 36, 38])
def test_horizon_is_a_launch_parameter(h):
    """Every other even horizon (oracle-solved extension fixtures, 4 instances each: walking, commanded v_x, per-step
    friction), on every kernel family that has it."""
    g = _hgen(h)
    for path in ((PATH_DENSE, PATH_STAGE) if h <= 20 else (PATH_STAGE,)):
        solver, mpc = _solver(h, int(g["half"][0]), path=path)
        states, controls, info = solver.solve(g["x_fb"], g["foot"], g["contact"], util.phases(g["t"], mpc.dt, h),
                                              x_cmd=g["x_cmd"], mu=g["mu_steps"])
        e, es = util.rel_err(controls, g["controls"]), util.rel_err(states, g["states"])
        print("h=%d path %d: err %.2e / %.2e iters %s" % (h, path, e.max(), es.max(), info["iters"]))
        assert (info["status"] == 0).all()
        assert e.max() <= util.REL_TOL and es.max() <= util.REL_TOL
        solver.close()


@pytest.mark.parametrize("h,B", [(32, 4096), (40, 4096), (9, 2048), (15, 2048), (25, 2048)])
def test_long_horizons_at_scale(h, B):
    """Full-size batches of the horizons only the stage family solves -- the long ones and (round 5) odd ones: every instance
    converges, every constraint holds, and the hardest instances agree with the oracle."""
    import biped_mpc_py_amd as bm
    s = util.synth_batch(B, h, 700 + h, gait="walking", vx_cmd=True, per_step_mu=True)
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    sol.close()
    print("h=%d B=%d: iters mean %.1f max %d nfactor %.2f unsolved %d" % (h, B, info["iters"].mean(), info["iters"].max(),
                                                                        info["nfactor"].mean(), int((info["status"] != 0).sum())))
    assert int((info["status"] != 0).sum()) == 0
    f = u.reshape(B, h, 4, 3)
    c = s["contact"].astype(float)
    assert (f[:, :, 0:2, :] >= -1e-3).all() and (f[:, :, 0:2, :] <= 500 * c[..., None] + 1e-3).all()      # REF:235-251
    mu = s["mu"]
    for k in range(2):                                                                                 # REF:220-232
        assert (np.abs(f[:, :, k, 0]) <= mu[:, :, k] * f[:, :, k, 2] + 2e-3).all()
        assert (np.abs(f[:, :, k, 1]) <= mu[:, :, k] * f[:, :, k, 2] + 2e-3).all()
    idx = np.argsort(-info["iters"], kind="stable")[:4]
    ref = _oracle_controls(s, idx, h)
    rel = util.rel_err(u[idx], ref)
    print("   hardest 4 (iterations %d..%d): max rel err %.2e" % (info["iters"][idx].min(), info["iters"][idx].max(), rel.max()))
    assert rel.max() <= util.REL_TOL


def test_stage_path_warm_start_and_rollout():
    """Receding-horizon use on the stage path (its own warm-start layout): a warm second solve of the shifted problem
    reaches the cold solve's optimum in fewer iterations on average; a closed-loop roll-out agrees with the dense path."""
    import torch
    import biped_mpc_py_amd as bm
    h, B = 10, 256
    s = util.synth_batch(B, h, 91, gait="standing")
    outs = {}
    for path in (PATH_DENSE, PATH_STAGE):
        mpc = bm.MPC()
        sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(path=path))
        _, u0, i0 = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], want_states=False)
        sol.set_warm_start(True, shift=0, theta=0.5)
        _, u1, i1 = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], want_states=False)       # stores
        x2 = s["x_fb"] + 0.01
        _, u2, i2 = sol.solve(x2, s["foot"], s["contact"], s["phase"], want_states=False)               # warm
        sol.set_warm_start(False)
        _, u3, i3 = sol.solve(x2, s["foot"], s["contact"], s["phase"], want_states=False)               # cold
        outs[path] = (u0, u2, u3, i2["iters"].mean(), i3["iters"].mean())
        assert (i2["status"] == 0).all() and (i3["status"] == 0).all()
        assert util.rel_err(u2, u3).max() <= util.REL_TOL
        print("path %d: warm %.1f cold %.1f iterations" % (path, i2["iters"].mean(), i3["iters"].mean()))
        assert i2["iters"].mean() < 0.85 * i3["iters"].mean()
        sol.close()
    assert util.rel_err(outs[PATH_STAGE][0], outs[PATH_DENSE][0]).max() <= util.REL_TOL


# ------------------------------------------------------------------------------------------------------------------
# the penalty schedule is scale-free -- weights, step length, mass, inertia, geometry, gravity
# ------------------------------------------------------------------------------------------------------------------
_NONDIAG_I = np.array([[0.932, 0.05, -0.03], [0.05, 0.942, 0.02], [-0.03, 0.02, 0.0711]])
_PARAM_CASES = {
    "default": (None, None),
    "R_div100": (lambda m: setattr(m, "R", np.asarray(m.R, float) / 100.0), None),
    "R_x100": (lambda m: setattr(m, "R", np.asarray(m.R, float) * 100.0), None),
    "Q_x100": (lambda m: setattr(m, "Q", np.asarray(m.Q, float) * 100.0), None),
    "dt_0.02": (lambda m: setattr(m, "dt", 0.02), None),
    "dt_0.05": (lambda m: setattr(m, "dt", 0.05), None),
    "m_8": (None, lambda b: setattr(b, "m", 8.0)),
    "m_20": (None, lambda b: setattr(b, "m", 20.0)),
    "I_nondiagonal": (None, lambda b: setattr(b, "I", _NONDIAG_I.copy())),
    "lt_lh": (None, lambda b: (setattr(b, "lt", 0.12), setattr(b, "lh", 0.07))),
    "g_3.7": (None, lambda b: setattr(b, "g", 3.7)),
    "kv_0.05": (lambda m: setattr(m, "kv", 0.05), None),
}
_param_iters = {}


@pytest.mark.parametrize("what", list(_PARAM_CASES))
def test_parameter_range_vs_oracle(what):
    """REF:22-48 are user-settable fields.  Each is moved away from the reference's default (weights by two decades) and
    the kernels are held to the oracle with the same parameters: all instances converge within max_iter, <= 1e-4, and
    the mean iteration count stays within 1.5x of the default-parameter case (the penalties scale with the curvature of
    the problem instead of being absolute numbers tuned at the reference's weights)."""
    import biped_mpc_py_amd as bm
    B = 16
    mpc_mod, biped_mod = _PARAM_CASES[what]
    tot = []
    for gait, seed, kw in (("standing", 61, {}), ("mixed", 62, dict(vx_cmd=True))):
        s = util.synth_batch(B, 10, seed, gait=gait, **kw)
        mpc, biped = bm.MPC(), bm.Biped()
        if mpc_mod:
            mpc_mod(mpc)
        if biped_mod:
            biped_mod(biped)
        # (phase follows t // dt: keep the schedule of the batch, whatever dt is)
        sol = bm.BatchSolver(mpc=mpc, biped=biped, half=s["half"], max_batch=B)
        _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], want_states=False)
        sol.close()
        ref = _oracle_controls(s, range(B), 10, mpc_mod, biped_mod)
        rel = util.rel_err(u, ref)
        print(what, gait, "err max %.2e iters mean %.1f max %d nfac %.1f" % (rel.max(), info["iters"].mean(), info["iters"].max(),
                                                                           info["nfactor"].mean()))
        assert int((info["status"] != 0).sum()) == 0
        assert rel.max() <= util.REL_TOL
        tot.append(info["iters"].mean())
    _param_iters[what] = float(np.mean(tot))
    if "default" in _param_iters:
        assert _param_iters[what] <= 1.5 * _param_iters["default"], (_param_iters[what], _param_iters["default"])


@pytest.mark.parametrize("name,h,half,idx", [("cfg4_walking_h10", 10, 5, [0, 5, 11, 23, 40, 63]), ("cfg3_trot_h16", 16, 8, [0, 7, 19]),
                                             ("cfg5_mu_h20", 20, 10, [1, 9, 15]),
                                             # round 6: commanded angular rates / attitudes -- Rot, R_inv, I_w differ at every step, so the prefix
                                             # sums P_i, the Me table and the Gram M'M on the matrix cores see non-identity steps
                                             ("cfg_cmd_h10", 10, 5, [0, 7, 19, 20, 33, 46, 58, 79])])
def test_assembly_is_the_condensed_qp_of_the_oracle(name, h, half, idx):
    """the kernel's OWN assembly output against the oracle directly, no model of the product in
    between: Hc = Wbar' Gt Wbar + 2 Rbar and gc = Wbar' qt formed from what bmpc_debug_assemble returns (Gt, qt and the
    lever arms foot_ref - com_ref) equal `build_condensed_qp` of the reference restatement (REF:203-216, 278-286)."""
    from oracle import bmpc_oracle as orc
    d = util.load(name)
    solver, mpc = _solver(h, half, path=PATH_DENSE)
    ph = util.phases(d["t"], mpc.dt, h)
    x_ref, foot_ref, Gt, qt = solver.assemble(d["x_fb"][idx], d["foot"][idx], d["contact"][idx], ph[idx], x_cmd=d["x_cmd"][idx])
    for n, i in enumerate(idx):
        m = orc.MPC()
        m.h = h
        m.x_cmd = d["x_cmd"][i]
        c = orc.build_condensed_qp(d["x_fb"][i].astype(np.float32).astype(float), float(d["t"][i]), d["foot"][i].astype(np.float32).astype(float),
                                   m, orc.Biped(), d["contact"][i], half=None if h == 10 else half)
        r = foot_ref[n].reshape(h, 2, 3) - x_ref[n][:, None, 3:6]
        W = util.wrench_map(r)
        Hc = W.T @ Gt[n] @ W + 2 * np.kron(np.eye(h), np.diag(np.asarray(m.R, float)))
        gc = W.T @ qt[n]
        eh = np.abs(Hc - c["Hc"]).max() / np.abs(c["Hc"]).max()
        eg = np.abs(gc - c["gc"]).max() / max(1.0, np.abs(c["gc"]).max())
        print(name, i, "Hc rel err %.2e gc rel err %.2e" % (eh, eg))
        assert eh <= 2e-6 and eg <= 2e-6          # Gt is formed in f64 from f32 step data (Me table)


def test_bench_two_ranks_on_one_device_over_rccl_fails_cleanly():
    """the N-rank RCCL program cannot run on a one-GPU box (RCCL refuses two ranks on one device),
    but its failure mode can be held: `bench.py --gpus 2 --backend nccl` started bare with both ranks on cuda:0 must come
    back with a non-zero exit code in bounded time, print no JSON line, and leave no rank behind holding the GPU."""
    import os
    import subprocess
    import sys
    import time
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("more than one GPU visible: the refusal this test expects needs two ranks on ONE device")
    env = dict(os.environ, BMPC_BENCH_LAUNCH_TIMEOUT="150", NCCL_DEBUG="WARN")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", "--share-device",
                        "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--batch", "256"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    took = time.time() - t0
    print("exit code", p.returncode, "after %.0f s" % took)
    print(p.stderr.decode("utf-8", "replace")[-1500:])
    assert p.returncode != 0
    assert not any(ln.startswith("{") for ln in p.stdout.decode("utf-8", "replace").splitlines())
    import psutil
    me = os.getpid()
    # (a Python process running bench.py as its script -- not a shell whose command line merely mentions it, and not an
    #  ancestor of this test, e.g. a wrapper that runs the test suite and the bench in one command)
    ancestors = {q.pid for q in psutil.Process(me).parents()}
    left = [q for q in psutil.process_iter(["pid", "cmdline", "ppid", "name"])
            if q.info["pid"] != me and q.info["pid"] not in ancestors and q.info["cmdline"]
            and "python" in (q.info["name"] or "") and any(a.endswith("bench.py") for a in q.info["cmdline"][1:3])]
    assert not left, [(q.info["pid"], q.info["cmdline"]) for q in left]
    # and the GPU still works for this process
    assert float(torch.ones(4, device="cuda").sum().item()) == 4.0


def _oracle_worker(a):
    from threadpoolctl import threadpool_limits
    from oracle import bmpc_oracle as orc
    with threadpool_limits(limits=1):
        x, f, c, xc, mu, h, half, ph = a
        mpc = orc.MPC()
        mpc.h = h
        mpc.x_cmd = xc
        _, ct, info = orc.solve_mpc(x, (ph + 0.5) * mpc.dt, f, mpc, orc.Biped(), c, half=half, mu_steps=mu, return_info=True)
        k = info["kkt"]
        return ct, bool(info["polished"]) and max(k["stationarity"], k["primal_ineq"], k["complementarity"]) <= 1e-7


# every BASELINE shape, instance by instance: (label, B, h, gait, seed, generator options)
_AT_SCALE = [("config2_standing_h10", 8192, 10, "standing", 31, {}),
             ("config4_mixed_h10", 8192, 10, "mixed", 3, dict(vx_cmd=True)),
             ("config3_trot_h16", 4096, 16, "walking", 2, dict(vx_cmd=True)),
             ("config5_mu_h20", 4096, 20, "walking", 4, dict(vx_cmd=True, per_step_mu=True)),
             # round 6: turning / attitude / lateral commands -- REF:64-69's rate branches for the Euler angles, so the
             # linearisation REF:148-185 differs at every step of the horizon -- at the three BASELINE horizons
             ("turning_mixed_h10", 4096, 10, "mixed", 61, dict(vx_cmd=True, turn=True)),
             ("turning_trot_h16", 4096, 16, "walking", 62, dict(vx_cmd=True, turn=True)),
             ("turning_mu_h20", 4096, 20, "walking", 63, dict(vx_cmd=True, per_step_mu=True, turn=True))]


# regression bounds (all controls, u0) of batches that do not sit at the common 1e-5 / 2e-5: none.  (The turning batches did not
# when they were added: the dense family ended at 1.1e-5 / 1.9e-5 on them -- the drift of its carried gradient between exact rebuilds,
# REFRESH_ITERS in bmpc_kernels.hip; rebuilt every 10 iterations instead of 20 it ends at 1.1e-6 / 1.6e-6 like the stage family.)
_AT_SCALE_BOUNDS = {}


@pytest.mark.parametrize("label,B,h,gait,seed,kw", _AT_SCALE, ids=[a[0] for a in _AT_SCALE])
def test_parity_against_the_oracle_at_scale(label, B, h, gait, seed, kw):
    """EVERY instance of a batch of every BASELINE config shape (2: standing, 4: mixed gaits, 3: h = 16 trot, 5: h = 20 with
    per-step friction) against the certified fp64 oracle on the fp32-rounded inputs the GPU sees, on both kernel families,
    on both metrics of SURVEY 8(d): all h x 12 controls, and the row the reference applies (`u0`, REF:493).  (The fixtures
    hold 16..64 instances per shape; a 1-in-10^4 failure only shows here.  The standing batch contains the two instances
    that used to stop early on small residuals, fixed by the third stopping test.)  Round 4: the dense family forms its
    gradient in state space like the stage family (no f32 copy of the Hessian in the fixed point) -- before that it sat at
    4e-5 on config 2 and, on the u0 metric, OUTSIDE the tolerance on single instances of configs 3 and 5 (1.0e-4, 6.2e-4 in
    512).  The log of this test is kept as profiles/r04_parity_at_scale.txt."""
    import multiprocessing as mp
    import os
    import biped_mpc_py_amd as bm
    s = util.synth_batch(B, h, seed, gait=gait, **kw)
    r32 = lambda v: v.astype(np.float32).astype(float)
    args = [(r32(s["x_fb"][i]), r32(s["foot"][i]), s["contact"][i], r32(s["x_cmd"][i]),
             None if s["mu"] is None else r32(s["mu"][i]), h, s["half"], int(s["phase"][i])) for i in range(B)]
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:
        res = pool.map(_oracle_worker, args, chunksize=16)
    ref = np.stack([r[0] for r in res])
    ok = np.array([r[1] for r in res])
    assert ok.mean() > 0.999                                    # (an uncertified reference is no yardstick)
    mpc = bm.MPC()
    mpc.h = h
    for path in (PATH_DENSE, PATH_STAGE):
        sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(path=path))
        _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
        sol.close()
        e, e0 = util.rel_err(u, ref)[ok], util.u0_err(u, ref)[ok]
        print("%s path %s (%d of %d references certified): all controls max %.2e p99.9 %.2e above 5e-5: %d | u0 max %.2e p99.9 %.2e "
              "above 5e-5: %d | iterations %.1f (max %d), not converged %d" % (
                  label, "dense" if path == PATH_DENSE else "stage", int(ok.sum()), B, e.max(), np.quantile(e, 0.999), int((e > 5e-5).sum()),
                  e0.max(), np.quantile(e0, 0.999), int((e0 > 5e-5).sum()), info["iters"].mean(), info["iters"].max(),
                  int((info["status"] != 0).sum())))
        assert (info["status"] == 0).all()
        assert e.max() <= util.REL_TOL and e0.max() <= util.REL_TOL
        # regression bounds well inside the tolerance (measured on MI355X: profiles/r04_parity_at_scale.txt; the turning batches of
        # round 6: profiles/r06_parity_at_scale.txt)
        b_all, b_u0 = _AT_SCALE_BOUNDS.get(label, (5e-6, 1e-5))        # (round 6: measured maxima 2.0e-6 / 2.8e-6 over the seven shapes, both families;
                                                                       #  1e-5 / 2e-5 until round 5)
        assert e.max() <= b_all and e0.max() <= b_u0, (e.max(), e0.max())


# BASELINE's full sizes (configs[3], configs[4]): (label, config number, instances checked against the oracle)
_FULL_SIZE = [("config4_65536_mixed_h10", 4, 2048), ("config5_65536_mu_h20", 5, 1024)]


@pytest.mark.parametrize("label,cfg,nref", _FULL_SIZE, ids=[a[0] for a in _FULL_SIZE])
def test_full_size_baseline_batches_against_the_oracle(label, cfg, nref):
    """BASELINE configs 4 and 5 at their FULL sizes -- the very 65536-instance batches `bench.py --config 4 | 5` solves (SURVEY 8(d)
    generator and seeds), product default path: every instance converges, every constraint of REF:220-271 holds, and the
    instances that took the most iterations (the hard end of the batch: half of the sample) plus a random half are held to the
    certified fp64 oracle on the fp32-rounded inputs the GPU saw, on both metrics of SURVEY 8(d) -- all h x 12 controls and the
    row the reference applies (`u0`, REF:493).  Round 4 measured all 65536 of each by hand (tools/full_size_parity.py,
    profiles/r04_parity_full_size.txt: config 5's worst u0 3.8e-5, three instances above 2e-5); this puts the full sizes under the
    driver's eyes.  Bounds: tolerance 1e-4; regression bounds 1e-5 (all controls) and 5e-5 (u0)."""
    import multiprocessing as mp
    import os
    import time
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import synth
    cf = synth.CONFIGS[cfg]
    B, h = cf["batch"], cf["h"]
    assert B == 65536
    s = synth.synth_batch(B, h, cf["seed"], gait=cf["gait"], **cf["kw"])
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    t0 = time.time()
    states, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"])
    t_gpu = time.time() - t0
    sol.close()
    assert (info["status"] == 0).all(), np.bincount(info["status"])
    assert np.isfinite(u).all() and np.isfinite(states).all()
    # every constraint, every instance (REF:220-251; the line-foot rows are held through the oracle sample)
    mu = s["mu"] if s["mu"] is not None else np.full((B, h, 2), 0.5)
    f = u.reshape(B, h, 4, 3)
    tol = 2e-3
    for j in range(2):
        c = s["contact"][:, :, j].astype(float)
        fx, fy, fz = f[:, :, j, 0], f[:, :, j, 1], f[:, :, j, 2]
        assert (fz >= -tol).all() and (fz <= 500 * c + tol).all()
        assert (fx >= -tol).all() and (fy >= -tol).all()                              # f_min = 0 on all three axes (REF:46)
        assert (fx <= mu[:, :, j] * fz + tol).all() and (fy <= mu[:, :, j] * fz + tol).all()
        assert (np.abs(f[:, :, 2 + j, 0]) <= tol).all()                               # tau_max[0] = 0 (REF:47)
        assert (np.abs(f[:, :, 2 + j, 1]) <= 67 * c + tol).all() and (np.abs(f[:, :, 2 + j, 2]) <= 33.5 * c + tol).all()
    # the sample: the hardest half by iteration count (ties: by factorisations), a random half of the rest
    order = np.lexsort((-info["nfactor"], -info["iters"]))
    hard = order[:nref // 2]
    rest = np.random.default_rng(cfg).permutation(order[nref // 2:])[:nref - nref // 2]
    idx = np.concatenate([hard, rest])
    r32 = lambda v: v.astype(np.float32).astype(float)
    args = [(r32(s["x_fb"][i]), r32(s["foot"][i]), s["contact"][i], r32(s["x_cmd"][i]),
             None if s["mu"] is None else r32(s["mu"][i]), h, s["half"], int(s["phase"][i])) for i in idx]
    t0 = time.time()
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:
        res = pool.map(_oracle_worker, args, chunksize=8)
    t_orc = time.time() - t0
    ref = np.stack([r[0] for r in res])
    ok = np.array([r[1] for r in res])
    assert ok.mean() > 0.995                                   # (an uncertified reference is no yardstick)
    e, e0 = util.rel_err(u[idx], ref)[ok], util.u0_err(u[idx], ref)[ok]
    nh = int(ok[:len(hard)].sum())
    print("%s: %d instances, all converged; iterations %.1f (max %d), factorisations %.2f (max %d); GPU call %.2f s, oracle %.1f s for %d" % (
        label, B, info["iters"].mean(), info["iters"].max(), info["nfactor"].mean(), info["nfactor"].max(), t_gpu, t_orc, len(idx)))
    print("  hardest %d (iterations >= %d): all controls max %.2e, u0 max %.2e | random %d: all controls max %.2e, u0 max %.2e | p99 %.2e / %.2e" % (
        nh, info["iters"][hard].min(), e[:nh].max(), e0[:nh].max(), len(e) - nh, e[nh:].max(), e0[nh:].max(),
        np.quantile(e, 0.99), np.quantile(e0, 0.99)))
    assert e.max() <= util.REL_TOL and e0.max() <= util.REL_TOL
    assert e.max() <= 1e-5 and e0.max() <= 5e-5, (e.max(), e0.max())


_OFF_REFERENCE = {"Q_x10": lambda m: setattr(m, "Q", np.asarray(m.Q, float) * 10.0),
                  "R_div100": lambda m: setattr(m, "R", np.asarray(m.R, float) / 100.0),
                  "dt_0.02": lambda m: setattr(m, "dt", 0.02)}


@pytest.mark.parametrize("h", [10, 20])
def test_dense_family_away_from_the_reference_weights_and_the_rescue_pass(h):
    """REF:27-28 are user fields.  Away from the reference's weights the dense family used to lose ~1 instance in 10^3..10^4
    (Q x 10: 5 of 16384 at h = 10, 8 at h = 20; R / 100 at h = 20: 55, on BOTH families) and leaned on the rescue pass.
    Round 4, at the root -- four separate causes: (i) a sweep that meets a pivot lost to f32 rounding (cond ~ 1e7 at Q x 10) is
    repeated once on the regularised matrix (the inverse only preconditions the residual form: the fixed point is untouched);
    (ii) instances that keep re-classifying get the factorisations and iterations they need (caps 60 / 1000-1500 instead of
    24 / 400-600: the 55 of R / 100 need up to 60 and 1025); (iii) such an instance never comes near the optimum, so the carried
    products are also rebuilt every 100 iterations wherever it is (drift made 1 of 16384 cycle for good); (iv) the dense family's
    penalty ceilings are capped at 4e5 (2 min R + rho_lo) instead of 1e6: its f32 explicit inverse preconditions every iteration,
    and with the moment ceiling at 500 two h = 20 instances re-classified until the cap.  With the rescue pass OFF, 16384
    standing instances per case: NOTHING lost.  The rescue pass stays as the safety net (default AUTO = on away from the
    reference's weights, off at them): exercised below on the penalties that used to fail."""
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd.params import RESCUE_AUTO, RESCUE_OFF, RESCUE_ON
    B = 16384
    s = util.synth_batch(B, h, 77 + h, gait="standing", per_step_mu=(h >= 20))

    def run(mod, mode, extra=None, n=B):
        mpc = bm.MPC()
        mpc.h = h
        mod(mpc)
        sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=n, solver_options=dict(rescue=mode, **(extra or {})))
        assert sol._lib.bmpc_solver_path(sol._h) == PATH_DENSE
        on = sol._lib.bmpc_rescue_enabled(sol._h)
        _, u, info = sol.solve(s["x_fb"][:n], s["foot"][:n], s["contact"][:n], s["phase"][:n],
                               mu=None if s["mu"] is None else s["mu"][:n], want_states=False)
        sol.close()
        return u, info, on

    for name, mod in _OFF_REFERENCE.items():
        u0, i0, on0 = run(mod, RESCUE_OFF)
        u1, i1, on1 = run(mod, RESCUE_AUTO)
        assert (on0, on1) == (0, 1)
        lost = np.nonzero(i0["status"])[0]
        print("h=%d %s: dense path alone: %d of %d not converged %s; iterations %.1f (max %d)" % (
            h, name, len(lost), B, lost[:10], i0["iters"].mean(), i0["iters"].max()))
        assert len(lost) == 0
        assert np.array_equal(u1, u0) and np.array_equal(i1["iters"], i0["iters"])     # (the pass found nothing to do)
    if h == 20:
        # the penalties Q x 10 resolved to before the dense family's cap, given as absolute values: the two instances that
        # re-classify until the cap are lost without the pass, solved by it (= the oracle's optimum), nobody else is touched
        bad = dict(penalty_mode=1, rho=0.1423, rho_eq_scale=300.0 / 0.1423, rho_hi_f=10.0, rho_hi_m=500.0)
        mod = _OFF_REFERENCE["Q_x10"]
        u0, i0, _ = run(mod, RESCUE_OFF, bad)
        u1, i1, on = run(mod, RESCUE_ON, bad)
        lost = np.nonzero(i0["status"])[0]
        print("h=20 Q_x10 with the uncapped ceilings: dense path alone loses %d %s" % (len(lost), lost[:6]))
        assert on == 1 and 1 <= len(lost) <= 8
        assert int((i1["status"] != 0).sum()) == 0
        keep = np.setdiff1d(np.arange(B), lost)
        assert np.array_equal(u1[keep], u0[keep]) and np.array_equal(i1["iters"][keep], i0["iters"][keep])
        # ... and through the handle's page-locked I/O block (round 5): the rescue launch stores its fp64 results into the same host
        # arrays -- straight from its epilogue (controls) and by the chunks' copies (states)
        mpc = bm.MPC()
        mpc.h = h
        mod(mpc)
        sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(rescue=RESCUE_ON, **bad))
        st_a, u_a, i_a = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], mu=s["mu"])
        st_b, u_b, i_b = sol.solve_inplace(s["x_fb"], s["foot"], s["contact"], s["phase"], mu=s["mu"])
        assert np.array_equal(u_a, u1) and np.array_equal(u_b, u_a) and np.array_equal(st_b, st_a)
        assert np.array_equal(i_b["iters"], i_a["iters"]) and int((i_b["status"] != 0).sum()) == 0
        sol.close()
        ref = _oracle_controls(s, lost[:4], h, mod, None)
        rel = util.rel_err(u1[lost[:4]], ref)
        print("rescued instances: err max %.2e" % rel.max())
        assert rel.max() <= util.REL_TOL
    # the reference's own model and weights: AUTO leaves the pass off (nothing to rescue in 6 M soaked instances), ON still works
    ident = lambda m: None
    ua, _, ona = run(ident, RESCUE_AUTO, n=256)
    ub, ib, onb = run(ident, RESCUE_ON, n=256)
    assert (ona, onb) == (0, 1) and int((ib["status"] != 0).sum()) == 0 and np.array_equal(ua, ub)
    # no rescue on the stage path itself
    mpc = bm.MPC()
    mpc.h = h
    _OFF_REFERENCE["Q_x10"](mpc)
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=8, solver_options=dict(path=PATH_STAGE, rescue=RESCUE_ON))
    assert sol._lib.bmpc_rescue_enabled(sol._h) == 0
    sol.close()


@pytest.mark.parametrize("h,path,B", [(10, PATH_DENSE, 2048), (20, PATH_DENSE, 1024), (24, PATH_STAGE, 1024), (40, PATH_STAGE, 1024)])
def test_runs_are_bitwise_reproducible(h, path, B):
    """The same batch solved four times gives the same bits (controls, states, iteration counts): enough instances to put
    more than one wave on a SIMD / several workgroups on a CU, which is where an exchange without its barrier, or a
    register shared by mistake, shows as a run-to-run difference (`profiles/r03_determinism.txt`)."""
    import biped_mpc_py_amd as bm
    s = util.synth_batch(B, h, 3, gait="walking", vx_cmd=True, per_step_mu=(h >= 20))
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(path=path))
    first = None
    for _ in range(4):
        st, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"])
        assert int((info["status"] != 0).sum()) == 0
        if first is None:
            first = (st, u, info["iters"].copy())
        else:
            assert np.array_equal(u, first[1]) and np.array_equal(st, first[0]) and np.array_equal(info["iters"], first[2])
    sol.close()


@pytest.mark.parametrize("h,gait,seed,kw,path,gain", [(10, "standing", 1, {}, PATH_DENSE, 0.97), (10, "mixed", 3, dict(vx_cmd=True), PATH_DENSE, 0.97),
                                                       (20, "walking", 4, dict(vx_cmd=True, per_step_mu=True), PATH_DENSE, 0.97),
                                                       (10, "mixed", 3, dict(vx_cmd=True), PATH_STAGE, 0.97),
                                                       (32, "walking", 5, dict(vx_cmd=True, per_step_mu=True), PATH_STAGE, 0.99)])
def test_secant_extrapolation_saves_iterations_and_keeps_the_fixed_point(h, gait, seed, kw, path, gain):
    """bmpc_params.accel (default on; both families -- the stage family with the x part of the state as the secant's metric): at a stopping test the iterate is extrapolated along its last state
    change, w <- T(w) - gamma (T(w) - w) with gamma from the last two changes (Anderson acceleration, memory one).  It is a
    different path to the SAME fixed point: every instance converges, the controls agree with the plain run to the
    solver's tolerance, and a batch needs >= 3 % fewer iterations (1 % at the long horizons, which re-classify twice as often)
    and no more factorisations (model and GPU: ~ -6 % at h = 10 .. 20)."""
    import biped_mpc_py_amd as bm
    B = 4096 if h <= 20 else 2048
    s = util.synth_batch(B, h, seed, gait=gait, **kw)
    out = {}
    for acc in (0, 1):
        mpc = bm.MPC()
        mpc.h = h
        sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(path=path, accel=acc))
        _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
        sol.close()
        assert int((info["status"] != 0).sum()) == 0
        out[acc] = (u, info["iters"].mean(), info["nfactor"].mean(), info["iters"].max())
    print("h=%d %s: iterations %.1f -> %.1f, factorisations %.2f -> %.2f, worst %d -> %d" % (
        h, gait, out[0][1], out[1][1], out[0][2], out[1][2], out[0][3], out[1][3]))
    assert out[1][1] <= gain * out[0][1] and out[1][2] <= out[0][2]
    assert util.rel_err(out[1][0], out[0][0]).max() <= util.REL_TOL
