"""Shared helpers of the parity tests: run a synthetic batch through the C-ABI (HIP library, or
the CPU wave emulator when `lib` is given) and through the CPU oracle."""
from __future__ import annotations

import numpy as np

import mpcqp
from mpcqp import synth
from oracle import condense as cd, estim as es, qp


def constraint_kwargs(cfg, oracle=False):
    names = (("umin", "umin"), ("umax", "umax"), ("dumin", "Δumin"), ("dumax", "Δumax"),
             ("ymin", "ymin"), ("ymax", "ymax"))
    kw = {}
    for k, kk in names:
        v = getattr(cfg, k)
        if np.isfinite(v):
            kw[k if oracle else kk] = np.full(cfg.nu if "u" in k else cfg.ny, v)
    return kw


def make_controller(cfg, bt, lib=None, **kw):
    mpc = mpcqp.BatchLinMPC(bt["Ahat"], bt["Bhu"], bt["Chat"], Hp=cfg.Hp, Hc=cfg.Hc, Cwt=cfg.Cwt,
                            Mwt=np.full(cfg.ny, cfg.Mwt), Nwt=np.full(cfg.nu, cfg.Nwt),
                            Lwt=np.full(cfg.nu, cfg.Lwt), lib=lib, **kw)
    mpc.setconstraint(**constraint_kwargs(cfg))
    return mpc


def run_batch(cfg, bt, lib=None, **kw):
    mpc = make_controller(cfg, bt, lib=lib, **kw)
    mpc.lastu0 = bt["lastu0"].copy()
    u = mpc.moveinput(bt["xhat0"], bt["ry"], want_info=True)
    info = mpc.getinfo()
    return {"Z": mpc.Z.copy(), "u": u, "status": mpc.status.copy(), "iters": mpc.iters.copy(),
            "Yhat": info["Ŷ"], "mpc": mpc}


def make_oracle(cfg, bt, i):
    m = cd.LinMPCOracle(bt["Ahat"][i], bt["Bhu"][i], bt["Chat"][i], Hp=cfg.Hp, Hc=cfg.Hc,
                        Cwt=cfg.Cwt, Mwt=np.full(cfg.ny, cfg.Mwt), Nwt=np.full(cfg.nu, cfg.Nwt),
                        Lwt=np.full(cfg.nu, cfg.Lwt))
    m.setconstraint(**constraint_kwargs(cfg, oracle=True))
    return m


def oracle_batch(cfg, bt):
    B = bt["xhat0"].shape[0]
    Z, U, cert, F, Q, H = [], [], [], [], [], []
    for i in range(B):
        m = make_oracle(cfg, bt, i)
        m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i])
        m.linconstraint()
        z, st, info = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
        Z.append(z)
        U.append(z[:cfg.nu] + bt["lastu0"][i])
        cert.append(info["certificate"] == "active-set")
        F.append(m.F)
        Q.append(m.qt)
        H.append(m.Ht)
    return {"Z": np.array(Z), "u": np.array(U), "certified": np.array(cert), "F": np.array(F),
            "q": np.array(Q), "H": np.array(H)}


def fused_loop_vs_separate_steps(lib=None, B=3, periods=4, torch_device=None, multiple_shooting=False):
    """mpcqp_loop_device (preparestate! + moveinput! + updatestate! in one launch) against the three
    separate entry points on the same resident data: returns max |difference| of x̂0, u0, Z̃ over the
    periods (the arithmetic is the same instruction for instruction: expected 0).  Arrays are torch
    tensors on `torch_device`, or NumPy arrays when the library is the CPU emulator."""
    cfg = synth.Config("loop", nx=3, nu=2, ny=2, Hp=8, Hc=3, umin=-0.6, umax=0.7, ymax=0.9)
    bt = synth.make_batch(cfg, B, seed=12)
    rng = np.random.default_rng(5)
    K = mpcqp.steady_kalman_gain(bt["Ahat"], bt["Chat"], np.eye(cfg.nxh), np.eye(cfg.ny))

    def make():
        # (FLAG_KEEP_QP: a step that keeps q̃ / F runs on the one-controller-per-wavefront kernel like the fused loop does; without
        #  it the separate step of this nZ̃ = 7 controller takes the small-problem kernel -- output-bound rows included since round
        #  4 -- whose arithmetic differs in the last bits)
        # (multiple_shooting: the stage-structured kernel, which fuses the Kalman steps since round 6; it has no q̃ to keep)
        hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=1,
                          flags=mpcqp.FLAG_RY_CONSTANT | (0 if multiple_shooting else mpcqp.FLAG_KEEP_QP), lib=lib)
        if multiple_shooting:
            hd.set_transcription(mpcqp.api.MULTIPLE_SHOOTING)
        hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
        hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt))
        hd.set_bounds(U0min=np.full((B, hd.nU), cfg.umin), U0max=np.full((B, hd.nU), cfg.umax), Y0max=np.full((B, hd.nY), cfg.ymax))
        hd.kf_set(mpcqp.colmajor(K), np.arange(cfg.ny))
        hd.prepare()
        return hd

    if torch_device is None:
        new = lambda a: np.ascontiguousarray(a).copy()
        ptr = lambda a: a.ctypes.data
        host = lambda a: a
        sync = lambda: None
    else:
        import torch
        new = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch_device)
        ptr = lambda a: a.data_ptr()
        host = lambda a: a.cpu().numpy()
        sync = torch.cuda.synchronize
    runs = []
    for fused in (False, True):
        hd = make()
        x = new(bt["xhat0"]); lu = new(bt["lastu0"]); ry = new(bt["ry"])
        Z = new(np.zeros((B, hd.nZ))); u0 = new(np.zeros((B, cfg.nu)))
        st = new(np.zeros(B, np.int32)); it = new(np.zeros(B, np.int32))
        rg = np.random.default_rng(7)
        out = []
        for k in range(periods):
            y = new(0.3 * rg.standard_normal((B, cfg.ny)))
            if fused:
                hd.loop_device(ptr(x), ptr(y), ptr(lu), ptr(ry), ptr(Z), ptr(u0), ptr(st), iters=ptr(it))
            else:
                hd.kf_correct_device(ptr(x), ptr(y))
                hd.step_device(ptr(x), ptr(lu), ptr(ry), ptr(Z), ptr(u0), ptr(st), iters=ptr(it))
                hd.kf_predict_device(ptr(x), ptr(u0))
            sync()
            assert np.all(host(st) == 0)
            out.append((host(x).copy(), host(u0).copy(), host(Z).copy()))
            lu, u0 = u0, lu                      # u0 of this period is lastu0 of the next
        runs.append(out)
    return max(float(np.abs(a - b).max()) for pa, pb in zip(*runs) for a, b in zip(pa, pb))


def write_c_fixture(path, name="C2", n=8):
    """Raw fixture for tests/abi_c_client.c `run`: the first n instances of a golden file
    (tests/golden/<name>_seed0-3.npz) in the C-ABI layout + their expected optimum."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{name}_seed0-3.npz"))
    cfg = synth.CONFIGS[name]
    bt = {k[3:]: g[k][:n] for k in g.files if k.startswith("in_")}
    B, nu, ny, Hp, Hc = n, cfg.nu, cfg.ny, cfg.Hp, cfg.Hc
    neps = 0 if np.isinf(cfg.Cwt) else 1
    fin = np.isfinite
    hdr = np.array([B, cfg.nxh, nu, ny, Hp, Hc, neps, int(fin(cfg.umin) or fin(cfg.umax)),
                    int(fin(cfg.dumin) or fin(cfg.dumax)), int(fin(cfg.ymax)), 0, 0], np.int32)
    full = lambda v, m: np.full((B, m), float(v))
    arrs = [mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]),
            full(cfg.Mwt, ny * Hp), full(cfg.Nwt, nu * Hc), full(cfg.Lwt, nu * Hp), np.full(B, cfg.Cwt if neps else 0.0)]
    if hdr[7]:
        arrs += [full(cfg.umin, nu * Hp), full(cfg.umax, nu * Hp)]
    if hdr[8]:
        arrs += [full(cfg.dumin, nu * Hc), full(cfg.dumax, nu * Hc)]
    if hdr[9]:
        arrs += [full(cfg.ymax, ny * Hp)]
    arrs += [bt["xhat0"], bt["lastu0"], bt["ry"], g["out_Z"][:n]]
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        for a in arrs:
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())


def build_c_client(exe, lib, root):
    import os, subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "abi_c_client.c"), lib, "-o", exe, "-lm",
                           "-Wl,-rpath," + os.path.dirname(os.path.abspath(lib)), "-Wl,-rpath,/opt/rocm/lib"])


def rel_err(Zg, Zo, nDU):
    """max_b ‖ΔU_gpu − ΔU_oracle‖∞ / max(1, ‖ΔU_oracle‖∞)  (BASELINE.md §4 'Parity')."""
    return np.max(np.abs(Zg[:, :nDU] - Zo[:, :nDU]), axis=1) / np.maximum(
        1.0, np.max(np.abs(Zo[:, :nDU]), axis=1))


def lqr_terminal_cost_case():
    """T6 (test/3_test_predictive_control.jl:498-527): terminal cost = DARE solution => LQR."""
    from scipy.linalg import solve_discrete_are
    A = np.array([[0.5, -0.4], [0.6, 0.5]]); Bu = np.eye(2); C = np.eye(2)
    Q, R = np.eye(2), 0.5 * np.eye(2)
    P = solve_discrete_are(A, Bu, Q, R)
    K = np.linalg.solve(R + Bu.T @ P @ Bu, Bu.T @ P @ A)
    M_Hp = np.block([[np.eye(4), np.zeros((4, 2))], [np.zeros((2, 4)), P]])
    return A, Bu, C, K, M_Hp


def run_lqr_terminal_cost(lib=None, B=3, steps=20, transcription="SingleShooting", kinds=None):
    """Closed loop of T6 through the C-ABI (nint_ym = 0: the state is measured); returns the MPC
    and the LQR state trajectories, (2, steps) each."""
    A, Bu, C, K, M_Hp = lqr_terminal_cost_case()
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    mpc = mpcqp.BatchLinMPC(rep(A), rep(Bu), rep(C), Hp=3, Hc=3, M_Hp=M_Hp, Nwt=[0, 0], Lwt=[0.5, 0.5], lib=lib,
                            transcription=transcription)
    X_mpc, X_lqr = np.zeros((2, steps)), np.zeros((2, steps))
    x = np.array([1.0, 1.0])
    for i in range(steps):
        u = mpc.moveinput(np.tile(x, (B, 1)), [0.0, 0.0])
        assert np.all(mpc.status == 0) and np.abs(u - u[0]).max() <= 1e-13
        if kinds is not None and i == 0:
            kinds.append(mpc.kernel)
        X_mpc[:, i] = x
        x = A @ x + Bu @ u[B - 1]
    x = np.array([1.0, 1.0])
    for i in range(steps):
        X_lqr[:, i] = x
        x = A @ x + Bu @ (-K @ x)
    return X_mpc, X_lqr


def custom_constraint_cases():
    """T9 (test/3_test_predictive_control.jl:466-495): custom linear constraints on
    model2 = LinModel([tf(2,[10,1]) tf(0.1,[7,1])], 3.0, i_d=[2]), uop=25, dop=30, yop=50, default
    SteadyKalmanFilter, LinMPC(Nwt=[0], Cwt=Inf, Hp=50, Hc=50, W..).  Returns the estimator and the
    list of (keywords W.., wmin, wmax, [(ry, which info, expected value), ...])."""
    from oracle import estim as es
    (a1, b1, c1), (a2, b2, c2) = ([float(np.squeeze(v)) for v in es.tf1_zoh(g, tau, 3.0)]
                                  for g, tau in ((2.0, 10.0), (0.1, 7.0)))
    model = es.LinModelOracle(np.diag([a1, a2]), [[b1], [0.0]], [[c1, c2]], [[0.0], [b2]], [[0.0]], Ts=3.0)
    model.setop(uop=[25], yop=[50], dop=[30])
    kf = es.SteadyKalmanFilterOracle(model)
    cases = [
        (dict(Wy=[[1.0]]), [36.0], [75.0], [(0.0, "Ŷ", 36.0), (100.0, "Ŷ", 75.0)]),
        (dict(Wu=[[1.0]]), [4.0], [20.0], [(0.0, "U", 4.0), (100.0, "U", 20.0)]),
        (dict(Wd=[[1.0]], Wy=[[1.0]]), [56.0], [95.0], [(0.0, "Ŷ", 56.0 - 30.0), (100.0, "Ŷ", 95.0 - 30.0)]),
        (dict(Wr=[[1.0]], Wy=[[1.0]]), [52.0], [175.0], [(21.0, "Ŷ", 52.0 - 21.0), (100.0, "Ŷ", 175.0 - 100.0)]),
    ]
    return model, kf, cases


def run_custom_constraint_cases(lib=None, B=2, Hp=50, which=(0, 1, 2, 3)):
    """T9 through the C-ABI against the oracle and (at the reference's horizon Hp = Hc = 50) the
    reference's expected values."""
    model, kf, cases = custom_constraint_cases()
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    worst = 0.0
    for kwW, wmin, wmax, checks in [cases[i] for i in which]:
        kw = dict(Hp=Hp, Hc=Hp, Nwt=[0], Cwt=np.inf, uop=model.uop, yop=model.yop, dop=model.dop,
                  xhop=kf.xhop, fhop=kf.fhop)
        orc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, **kw, **kwW)
        orc.setconstraint(wmin=wmin, wmax=wmax)
        gpu = mpcqp.BatchLinMPC(rep(kf.Ah), rep(kf.Bhu), rep(kf.Ch), rep(kf.Bhd), rep(kf.Dhd), lib=lib, **kw, **kwW)
        gpu.setconstraint(wmin=wmin, wmax=wmax)
        x0 = np.zeros(kf.nxh)
        gpu.initstate([25.0]); orc.lastu0 = np.zeros(1)
        for ry, key, want in checks:
            ug = gpu.moveinput(np.tile(x0, (B, 1)), [ry], [30.0], want_info=True)
            uo = orc.moveinput(x0, [ry], [30.0])
            assert np.all(gpu.status == 0)
            ig, io = gpu.getinfo(), orc.getinfo()
            if Hp == 50:
                assert np.all(np.abs(ig[key][B - 1] - want) < 1e-1), (kwW, ry)
            worst = max(worst, np.abs(gpu.Z[B - 1] - orc.Zt).max() / max(1.0, np.abs(orc.Zt).max()),
                        np.abs(ig["W"][B - 1] - io["W"]).max() / max(1.0, np.abs(io["W"]).max()))
    return worst


def run_soft_custom_constraints(lib=None, B=2, seed=4, kinds=None, Hp=8, Hc=(1, 2, 2), terminal=False, periods=3):
    """Two soft custom rows mixing outputs, inputs, a measured disturbance and the set point, on top
    of ordinary u / y constraints, against the oracle (exercises the ϵ row of the custom block).
    Hp, Hc: the horizons (round 6: Hp = Hc = 60, nZ̃ = 121 puts the handle on a team of wavefronts, whose helpers take the
    custom-row and terminal-row parts of the Newton matrix); terminal: a soft bound on one terminal state on top."""
    from oracle import estim as es
    rng = np.random.default_rng(seed)
    A = np.diag([0.85, 0.6, 0.3]); Bu = rng.standard_normal((3, 2)); C = rng.standard_normal((2, 3))
    Bd = rng.standard_normal((3, 1)); Dd = rng.standard_normal((2, 1))
    model = es.LinModelOracle(A, Bu, C, Bd, Dd).setop(uop=[0.5, -0.2], yop=[2.0, 1.0], dop=[0.3])
    kf = es.SteadyKalmanFilterOracle(model)
    Wy, Wu = rng.standard_normal((2, 2)), rng.standard_normal((2, 2))
    Wd, Wr = rng.standard_normal((2, 1)), 0.3 * rng.standard_normal((2, 2))
    kw = dict(Hp=Hp, Hc=list(Hc) if not np.isscalar(Hc) else int(Hc), Lwt=[0.05, 0.02], uop=model.uop, yop=model.yop, dop=model.dop,
              xhop=kf.xhop, fhop=kf.fhop, Wy=Wy, Wu=Wu, Wd=Wd, Wr=Wr)
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    orc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, **kw)
    gpu = mpcqp.BatchLinMPC(rep(kf.Ah), rep(kf.Bhu), rep(kf.Ch), rep(kf.Bhd), rep(kf.Dhd), lib=lib, **kw)
    con = dict(umin=[-0.6, -1.0], umax=[1.4, 0.9], ymax=[2.6, 1.8], wmin=[0.2, -np.inf], wmax=[1.5, 0.9],
               c_wmin=[0.7, 1.0], c_wmax=[1.3, 0.4])
    if terminal:
        xm = np.full(kf.nxh, np.inf); xm[0] = 0.4
        con["xhatmax"] = kf.xhop + xm
    orc.setconstraint(**con); gpu.setconstraint(**{("x̂max" if k == "xhatmax" else k): v for k, v in con.items()})
    x0 = 0.3 * rng.standard_normal(kf.nxh)
    gpu.initstate([0.6, 0.0]); orc.lastu0 = np.array([0.6, 0.0]) - model.uop
    worst = 0.0
    for k in range(periods):
        ry, d = [2.5 + 0.2 * k, 0.4], [0.5 - 0.1 * k]
        ug = gpu.moveinput(np.tile(x0, (B, 1)), ry, d, want_info=True)
        uo = orc.moveinput(x0, ry, d)
        assert np.all(gpu.status == 0)
        ig, io = gpu.getinfo(), orc.getinfo()
        worst = max(worst, np.abs(gpu.Z[B - 1] - orc.Zt).max() / max(1.0, np.abs(orc.Zt).max()),
                    np.abs(ug[B - 1] - uo).max(), np.abs(ig["W"][B - 1] - io["W"]).max())
        x0 = kf.Ah @ x0 + kf.Bhu @ (uo - model.uop) * 0.5
    if kinds is not None:
        kinds.append(gpu.hd.kernel_kind())
    return worst


def closed_loop_pair(cfg, bt, steps, lib=None, noise=0.02, seed=1, **kw):
    """The same noisy closed loop (plant = model + state noise) run by two controllers of one
    configuration, the second with keyword overrides `kw`; returns per-step (Z_a, Z_b, it_a, it_b)."""
    a = make_controller(cfg, bt, lib=lib)
    b_ = make_controller(cfg, bt, lib=lib, **kw)
    for c in (a, b_):
        c.lastu0 = bt["lastu0"].copy()
    x = bt["xhat0"].copy()
    rg = np.random.default_rng(seed)
    out = []
    for k in range(steps):
        ua = a.moveinput(x, bt["ry"])
        ub = b_.moveinput(x, bt["ry"])
        assert np.all(a.status == 0) and np.all(b_.status == 0)
        out.append((a.Z.copy(), b_.Z.copy(), a.iters.copy(), b_.iters.copy()))
        x = (np.einsum("bij,bj->bi", bt["Ahat"], x) + np.einsum("bij,bj->bi", bt["Bhu"], ua)
             + noise * rg.standard_normal(x.shape))
        b_.lastu0 = a.lastu0.copy()      # keep the two loops on the same trajectory
    return out


EXTRA_KW = None      # (experiments: extra BatchLinMPC keywords for run_random_case)


def run_random_case(seed, lib=None, B=3, small=False, large=False, huge=False, kinds=None, transcription="SingleShooting", huge2=False, ny4=False):
    """One randomly drawn controller family (dimensions, move blocking, which bounds exist, hard /
    soft mix, terminal bounds, measured disturbance, Cwt finite or Inf) as a batch of B DIFFERENT
    controllers of that family -- every member has its own model, weights, operating points, bound
    values, pattern of +-Inf holes, state, set point and disturbance -- stepped twice through the
    C-ABI and, member by member, through the oracle.  Returns the worst relative ΔU error over all
    members and steps whose oracle optimum carries an exact certificate (None if none did).
    Member 0 is drawn from the family's own generator (the instances the earlier single-controller
    form of this test compared), the others from generators of their own."""
    from oracle import estim as es
    rng = np.random.default_rng(1000 + seed)
    nx = int(rng.integers(2, 4 if small else 7)); nu = int(rng.integers(1, 3 if small else 5))
    ny = int(rng.integers(1, 3 if small else 4)); nd = int(rng.integers(0, 2))
    Hp = int(rng.integers(4, 9 if small else 24))
    if ny4:                                 # four outputs, nu a divisor of 16: the shapes whose E'DE takes its operands from registers (round 6)
        ny = 4; nu = int(rng.choice([1, 2, 4])); Hp = int(rng.integers(8, 31))
    if huge and not huge2:                  # beyond one row per lane: 64 < nZ~ <= ~130
        nu = int(rng.integers(2, 5)); Hp = int(rng.integers(32, 46))
        Hc = min(Hp, int(rng.integers(66, 130)) // nu)
    elif huge2:                             # beyond two rows per lane: 130 < nZ~ <= 165 (round 6; three row slots per lane)
        nu = int(rng.integers(3, 5)); Hc = int(rng.integers(131, 165)) // nu; Hp = Hc + int(rng.integers(0, 5))
    elif large:                             # close to the one-wavefront limit nZ~ = 64
        nu = int(rng.integers(2, 5)); Hp = int(rng.integers(16, 31))
        Hc = min(Hp, 63 // nu) - int(rng.integers(0, 3))
    elif rng.random() < 0.5:
        Hc = int(rng.integers(1, min(Hp, (60 // nu)) + 1))
    else:                                   # a move-blocking vector (sums to <= Hp, construct.jl:629-660)
        parts = []
        while sum(parts) < Hp - 1 and len(parts) * nu < 56 and len(parts) < 6:
            parts.append(int(rng.integers(1, 4)))
            if sum(parts) > Hp:
                parts[-1] -= sum(parts) - Hp
        Hc = [p for p in parts if p > 0] or 1
    fam = {}                                # structural decisions of the family, set by member 0

    def decide(key, value):                 # member 0 decides, the others follow
        return fam.setdefault(key, value)

    def member(rg, idx=0):
        lam = rg.uniform(0.3, 0.97, nx)
        Q, _ = np.linalg.qr(rg.standard_normal((nx, nx)))
        A = Q @ np.diag(lam) @ Q.T
        Bu = rg.standard_normal((nx, nu)) / np.sqrt(nx); C = rg.standard_normal((ny, nx)) / np.sqrt(nx)
        Bd = rg.standard_normal((nx, nd)); Dd = 0.3 * rg.standard_normal((ny, nd))
        model = es.LinModelOracle(A, Bu, C, Bd, Dd).setop(uop=0.2 * rg.standard_normal(nu),
                                                          yop=rg.standard_normal(ny), dop=0.3 * rg.standard_normal(nd))
        if seed % 2 == 0:
            # a linearisation point that is NOT an equilibrium: f̂op ≠ x̂op, per member (the successive-linearisation use,
            # docs/src/manual/nonlinmpc.md:501) -- B = [Ĉ S(t)](f̂op − x̂op) and bx̂ of init_predmat (transcription.jl:184-192)
            # are then non-zero.  Drawn from a generator of its own: the other draws of the family stay what they were.
            r2 = np.random.default_rng([777, seed, idx])
            model.setop(xop=0.3 * r2.standard_normal(nx), fop=0.3 * r2.standard_normal(nx))
        kf = es.SteadyKalmanFilterOracle(model)
        soft = decide("soft", bool(rg.random() < 0.75))
        Mw, Nw = rg.uniform(0.5, 2.0, ny), rg.uniform(0.02, 0.3, nu)
        Lw = rg.uniform(0.0, 0.1, nu) * (rg.random() < 0.5)
        cw = 10 ** rg.uniform(3, 5.5)
        kw = dict(Hp=Hp, Hc=Hc, Mwt=Mw, Nwt=Nw, Lwt=Lw, Cwt=cw if soft else np.inf,
                  uop=model.uop, yop=model.yop, dop=model.dop, xhop=kf.xhop, fhop=kf.fhop)
        orc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, **kw)
        inf_some = lambda v: np.where(rg.random(v.shape) < 0.25, np.inf * np.sign(v), v)
        con = {}
        if decide("u", bool(rg.random() < 0.8)):
            con["umin"] = inf_some(model.uop - rg.uniform(0.3, 1.2, nu))
            con["umax"] = inf_some(model.uop + rg.uniform(0.3, 1.2, nu))
        if decide("du", bool(rg.random() < 0.6)):
            con["dumin"] = inf_some(-rg.uniform(0.1, 0.6, nu))
            con["dumax"] = inf_some(rg.uniform(0.1, 0.6, nu))
        if decide("y", bool(soft and rg.random() < 0.8)):
            con["ymin"] = inf_some(model.yop - rg.uniform(0.2, 1.5, ny))
            con["ymax"] = inf_some(model.yop + rg.uniform(0.2, 1.5, ny))
        if decide("x", bool(soft and rg.random() < 0.4)):
            xm = np.full(kf.nxh, np.inf); xm[int(rg.integers(0, kf.nxh))] = 0.6
            con["xhatmax"] = kf.xhop + xm
        if soft:                                # softness: some rows hard (0), some soft
            for base, n in (("c_umin", nu), ("c_umax", nu), ("c_dumin", nu), ("c_dumax", nu),
                            ("c_ymin", ny), ("c_ymax", ny)):
                if base[2:] in con and decide(base, bool(rg.random() < 0.5)):
                    con[base] = rg.uniform(0.2, 1.5, n) * (rg.random(n) < 0.6 if base[2] != "y" else 1.0)
        orc.setconstraint(**con)
        x0 = 0.5 * rg.standard_normal(kf.nxh)
        u_prev = model.uop + 0.2 * rg.standard_normal(nu)
        orc.lastu0 = u_prev - model.uop
        return dict(model=model, kf=kf, orc=orc, kw=kw, con=con, x0=x0, u_prev=u_prev, rg=rg)

    mem = [member(rng)] + [member(np.random.default_rng([1000 + seed, i]), i) for i in range(1, B)]
    st = lambda f: np.stack([f(m) for m in mem])
    nxh = mem[0]["kf"].nxh
    gpu = mpcqp.BatchLinMPC(st(lambda m: m["kf"].Ah), st(lambda m: m["kf"].Bhu), st(lambda m: m["kf"].Ch),
                            st(lambda m: m["kf"].Bhd) if nd else None, st(lambda m: m["kf"].Dhd) if nd else None,
                            lib=lib, Hp=Hp, Hc=Hc, Mwt=st(lambda m: m["kw"]["Mwt"]), Nwt=st(lambda m: m["kw"]["Nwt"]),
                            Lwt=st(lambda m: m["kw"]["Lwt"]), Cwt=st(lambda m: m["kw"]["Cwt"]),
                            uop=st(lambda m: m["model"].uop), yop=st(lambda m: m["model"].yop),
                            dop=st(lambda m: m["model"].dop), xhop=st(lambda m: m["kf"].xhop),
                            fhop=st(lambda m: m["kf"].fhop), transcription=transcription, **(EXTRA_KW or {}))
    gname = dict(dumin="Δumin", dumax="Δumax", c_dumin="c_Δumin", c_dumax="c_Δumax", xhatmax="x̂max")
    gpu.setconstraint(**{gname.get(k, k): st(lambda m: m["con"][k]) for k in mem[0]["con"]})
    gpu.initstate(st(lambda m: m["u_prev"]))
    worst = None
    for k in range(2):
        if kinds is not None and k == 1:          # (the first moveinput prepared the kernel of the handle)
            kinds.append((gpu.hd.kernel_kind(), gpu.hd.nZ))
        for m in mem:
            rg, model = m["rg"], m["model"]
            m["ry"] = model.yop + rg.standard_normal(ny) * (1.5 if k == 0 else 0.5)
            m["d"] = model.dop + 0.3 * rg.standard_normal(nd) if nd else None
            m["Dhat"] = (np.tile(m["d"], Hp) + 0.05 * rg.standard_normal(nd * Hp)) if nd else None
        gpu.moveinput(st(lambda m: m["x0"]), st(lambda m: m["ry"]), st(lambda m: m["d"]) if nd else None,
                      Dhat=st(lambda m: m["Dhat"]) if nd else None)
        stop = False
        for i, m in enumerate(mem):
            orc, model = m["orc"], m["model"]
            orc.initpred(m["x0"], orc.lastu0 + model.uop, m["ry"], m["d"], m["Dhat"]); orc.linconstraint()
            z, sto, info = qp.solve_qp(*orc.qp_data(), orc.warmstart(), return_info=True)
            if sto != 0:                  # the oracle gave up on this one (it would take its error
                stop = True               # branch and the two loops would no longer see the same inputs)
                break
            assert gpu.status[i] == 0, (seed, i, gpu.status)
            e = rel_err(gpu.Z[i:i + 1], z[None, :], orc.nDU).max()
            if info["certificate"] == "active-set":
                worst = e if worst is None else max(worst, e)
            elif e > 1e-4:                # uncertified oracle point that differs: nobody to compare
                stop = True               # with, and the two loops would part ways from here on
                break
            uo = orc.moveinput(m["x0"], m["ry"], m["d"], Dhat=m["Dhat"])
            m["x0"] = m["kf"].Ah @ m["x0"] + m["kf"].Bhu @ (uo - model.uop) + m["kf"].fhop - m["kf"].xhop
        if stop:
            break
    return worst


def offset_tables_case(lib=None, nx=3, nu=2, ny=2, Hp=7, Hc=3, B=3, seed=11):
    """init_predmat with f̂op ≠ x̂op (transcription.jl:184-192: B = [Ĉ S(t)](f̂op − x̂op), bx̂ = S(Hp−1)(f̂op − x̂op)), one
    offset per member: the B table read back through MPCQP_GET_BVEC, the free response F (execute.jl:248-255) through
    MPCQP_GET_FVEC, and -- with a terminal bound, whose right-hand side carries bx̂ -- the optimum, each against the dense
    restatement.  Returns (worst |B − B_oracle|, worst |F − F_oracle|, worst relative ΔU error, worst |B_oracle|)."""
    from oracle import estim as es
    mem = []
    for i in range(B):
        rg = np.random.default_rng([seed, i])
        lam = rg.uniform(0.4, 0.95, nx)
        Q, _ = np.linalg.qr(rg.standard_normal((nx, nx)))
        A = Q @ np.diag(lam) @ Q.T
        model = es.LinModelOracle(A, rg.standard_normal((nx, nu)) / np.sqrt(nx), rg.standard_normal((ny, nx)) / np.sqrt(nx),
                                  np.zeros((nx, 0)), np.zeros((ny, 0)))
        model.setop(uop=0.2 * rg.standard_normal(nu), yop=rg.standard_normal(ny),
                    xop=0.5 * rg.standard_normal(nx), fop=0.5 * rg.standard_normal(nx))
        kf = es.SteadyKalmanFilterOracle(model)
        kw = dict(Hp=Hp, Hc=Hc, Mwt=rg.uniform(0.5, 2.0, ny), Nwt=rg.uniform(0.05, 0.3, nu), Lwt=np.zeros(nu), Cwt=1e4,
                  uop=model.uop, yop=model.yop, dop=model.dop, xhop=kf.xhop, fhop=kf.fhop)
        orc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, **kw)
        xm = np.full(kf.nxh, np.inf); xm[0] = 0.3
        con = dict(umin=model.uop - 0.8, umax=model.uop + 0.8, xhatmax=kf.xhop + xm)
        orc.setconstraint(**con)
        u_prev = model.uop + 0.1 * rg.standard_normal(nu)
        orc.lastu0 = u_prev - model.uop
        mem.append(dict(model=model, kf=kf, orc=orc, kw=kw, con=con, x0=0.5 * rg.standard_normal(kf.nxh), u_prev=u_prev,
                        ry=model.yop + rg.standard_normal(ny)))
    st = lambda f: np.stack([f(m) for m in mem])
    gpu = mpcqp.BatchLinMPC(st(lambda m: m["kf"].Ah), st(lambda m: m["kf"].Bhu), st(lambda m: m["kf"].Ch), lib=lib, Hp=Hp, Hc=Hc,
                            Mwt=st(lambda m: m["kw"]["Mwt"]), Nwt=st(lambda m: m["kw"]["Nwt"]), Lwt=st(lambda m: m["kw"]["Lwt"]),
                            Cwt=1e4, uop=st(lambda m: m["model"].uop), yop=st(lambda m: m["model"].yop),
                            xhop=st(lambda m: m["kf"].xhop), fhop=st(lambda m: m["kf"].fhop), keep_qp=True)
    gpu.setconstraint(umin=st(lambda m: m["con"]["umin"]), umax=st(lambda m: m["con"]["umax"]),
                      **{"x̂max": st(lambda m: m["con"]["xhatmax"])})
    gpu.initstate(st(lambda m: m["u_prev"]))
    gpu.moveinput(st(lambda m: m["x0"]), st(lambda m: m["ry"]))
    Bv, F = gpu.hd.get(mpcqp.GET_BVEC), gpu.hd.get(mpcqp.GET_FVEC)
    eB = eF = eZ = bmax = 0.0
    for i, m in enumerate(mem):
        orc, model = m["orc"], m["model"]
        orc.initpred(m["x0"], orc.lastu0 + model.uop, m["ry"], None, None); orc.linconstraint()
        z, sto, info = qp.solve_qp(*orc.qp_data(), orc.warmstart(), return_info=True)
        assert sto == 0 and gpu.status[i] == 0, (i, sto, gpu.status)
        eB = max(eB, np.abs(Bv[i] - orc.B).max()); bmax = max(bmax, np.abs(orc.B).max())
        eF = max(eF, np.abs(F[i] - orc.F).max())
        eZ = max(eZ, rel_err(gpu.Z[i:i + 1], z[None, :], orc.nDU).max())
    return eB, eF, eZ, bmax


def run_random_case2(seed, lib=None, B=2, small=False, kinds=None):
    """Like run_random_case, for the horizon-wide forms: time-varying Umin/Umax/Ymin/Ymax vectors
    (with ±Inf holes), R̂y / R̂u / D̂ trajectories, a block-diagonal M_Hp, and (every other seed)
    custom linear constraints Wy/Wu/Wd/Wr.  Returns the worst relative ΔU error over certified steps."""
    from oracle import estim as es
    rng = np.random.default_rng(5000 + seed)
    nx = int(rng.integers(2, 4 if small else 6)); nu = int(rng.integers(1, 3 if small else 4))
    ny = int(rng.integers(1, 3 if small else 4)); nd = int(rng.integers(0, 2))
    Hp = int(rng.integers(4, 8 if small else 16)); Hc = int(rng.integers(1, min(Hp, 6) + 1))
    lam = rng.uniform(0.3, 0.95, nx)
    Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    A = Q @ np.diag(lam) @ Q.T
    Bu = rng.standard_normal((nx, nu)) / np.sqrt(nx); C = rng.standard_normal((ny, nx)) / np.sqrt(nx)
    Bd = rng.standard_normal((nx, nd)); Dd = 0.3 * rng.standard_normal((ny, nd))
    model = es.LinModelOracle(A, Bu, C, Bd, Dd).setop(uop=0.2 * rng.standard_normal(nu),
                                                      yop=rng.standard_normal(ny), dop=0.3 * rng.standard_normal(nd))
    kf = es.SteadyKalmanFilterOracle(model)
    kw = dict(Hp=Hp, Hc=Hc, Nwt=rng.uniform(0.02, 0.3, nu), Lwt=rng.uniform(0.0, 0.1, nu),
              Cwt=10 ** rng.uniform(3, 5), uop=model.uop, yop=model.yop, dop=model.dop, xhop=kf.xhop, fhop=kf.fhop)
    if rng.random() < 0.5:                        # block-diagonal M_Hp with a dense terminal block
        M = np.zeros((ny * Hp, ny * Hp))
        for t in range(Hp):
            R = rng.standard_normal((ny, ny)) * (0.3 if t < Hp - 1 else 1.0)
            M[t * ny:(t + 1) * ny, t * ny:(t + 1) * ny] = np.eye(ny) * rng.uniform(0.5, 2.0) + (R @ R.T if t == Hp - 1 else 0.0)
        kw["M_Hp"] = M
    else:
        kw["Mwt"] = rng.uniform(0.5, 2.0, ny)
    if seed % 2 == 1:
        nw = int(rng.integers(1, 3))
        kw.update(Wy=rng.standard_normal((nw, ny)), Wu=rng.standard_normal((nw, nu)),
                  Wr=0.3 * rng.standard_normal((nw, ny)))
        if nd:
            kw["Wd"] = rng.standard_normal((nw, nd))
    orc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, **kw)
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    gpu = mpcqp.BatchLinMPC(rep(kf.Ah), rep(kf.Bhu), rep(kf.Ch), rep(kf.Bhd) if nd else None,
                            rep(kf.Dhd) if nd else None, lib=lib, **kw)
    holes = lambda v, sgn: np.where(rng.random(v.shape) < 0.3, sgn * np.inf, v)
    con = dict(Umin=holes(np.tile(model.uop, Hp) - rng.uniform(0.3, 1.2, nu * Hp), -1),
               Umax=holes(np.tile(model.uop, Hp) + rng.uniform(0.3, 1.2, nu * Hp), +1),
               Ymin=holes(np.tile(model.yop, Hp) - rng.uniform(0.3, 1.5, ny * Hp), -1),
               Ymax=holes(np.tile(model.yop, Hp) + rng.uniform(0.3, 1.5, ny * Hp), +1))
    if seed % 2 == 1:
        con.update(wmin=np.where(rng.random(nw) < 0.3, -np.inf, -rng.uniform(0.5, 2.0, nw)),
                   wmax=rng.uniform(0.5, 2.0, nw), c_wmax=rng.uniform(0.3, 1.5, nw))
    orc.setconstraint(**con); gpu.setconstraint(**con)
    x0 = 0.5 * rng.standard_normal(kf.nxh)
    u_prev = model.uop + 0.2 * rng.standard_normal(nu)
    gpu.initstate(u_prev); orc.lastu0 = u_prev - model.uop
    worst = None
    for k in range(2):
        ry = model.yop + rng.standard_normal(ny)
        # (odd seeds: custom constraints with a Wr term AND a set point trajectory whose first block is not ry(k))
        Rhaty = np.tile(ry, Hp) + 0.2 * rng.standard_normal(ny * Hp) if seed % 4 != 2 else None
        Rhatu = np.tile(model.uop, Hp) + 0.1 * rng.standard_normal(nu * Hp)
        d = model.dop + 0.3 * rng.standard_normal(nd) if nd else None
        Dhat = (np.tile(d, Hp) + 0.05 * rng.standard_normal(nd * Hp)) if nd else None
        gpu.moveinput(np.tile(x0, (B, 1)), ry, d, Dhat=Dhat, Rhaty=Rhaty, Rhatu=Rhatu, want_info=True)
        orc.initpred(x0, orc.lastu0 + model.uop, ry, d, Dhat, Rhaty, Rhatu); orc.linconstraint()
        z, st, info = qp.solve_qp(*orc.qp_data(), orc.warmstart(), return_info=True)
        if st != 0:
            break
        assert np.all(gpu.status == 0), (seed, gpu.status)
        e = rel_err(gpu.Z[B - 1:B], z[None, :], orc.nDU).max()
        if info["certificate"] == "active-set":
            worst = e if worst is None else max(worst, e)
        elif e > 1e-4:                    # uncertified oracle point that differs: nobody to compare
            break                         # with, and the two loops would part ways from here on
        if kinds is not None and k == 0:
            kinds.append((gpu.hd.kernel_kind(), gpu.hd.nZ))
        uo = orc.moveinput(x0, ry, d, Dhat=Dhat, Rhaty=Rhaty, Rhatu=Rhatu)
        if seed % 2 == 1:
            assert np.abs(gpu.getinfo()["W"][B - 1] - orc.getinfo()["W"]).max() <= 1e-5
        x0 = kf.Ah @ x0 + kf.Bhu @ (uo - model.uop)
    return worst


def readme_example(lib=None, B=2, steps=40):
    """BASELINE config 0, the reference's README example (README.md:47-74): G(s) = [2 e^{-20s}/(10s+1);
    10/(4s+1)], Ts = 1, `LinMPC(model, Mwt=[1, 0], Nwt=[0.1])` (defaults Hp = 10 + 20 delays,
    construct.jl:569-591, Hc = 2, Cwt = 1e5, SteadyKalmanFilter), `setconstraint!(ymax=[Inf, 35])`,
    `sim!(mpc, 40, [5, 0])` (plot_sim.jl:291-311: evaloutput, preparestate!, moveinput!, both
    updatestate!).  The oracle loop drives the plant; the batch goes through the C-ABI (estimator
    steps included) on the same measurements.  Returns the worst |u_abi - u_oracle| and the
    trajectories (u, y)."""
    from oracle import estim as es
    (a1, b1, c1), (a2, b2, c2) = ([float(np.squeeze(v)) for v in es.tf1_zoh(g, tau, 1.0)]
                                  for g, tau in ((2.0, 10.0), (10.0, 4.0)))
    nk = 20
    nx = nk + 2
    A = np.zeros((nx, nx)); Bu = np.zeros((nx, 1)); C = np.zeros((2, nx))
    A[0, 0] = a1; Bu[0, 0] = b1
    A[1, 0] = c1                                  # w_1(k+1) = c1 x1(k); w_{i+1}(k+1) = w_i(k); y1 = w_20
    for i in range(2, nk + 1):
        A[i, i - 1] = 1.0
    C[0, nk] = 1.0
    A[nk + 1, nk + 1] = a2; Bu[nk + 1, 0] = b2; C[1, nk + 1] = c2
    model = es.LinModelOracle(A, Bu, C, Ts=1.0)
    plant = es.LinModelOracle(A, Bu, C, Ts=1.0)
    kf = es.SteadyKalmanFilterOracle(model)
    Hp = 10 + int(np.sum(np.abs(np.linalg.eigvals(A)) < 1e-3))
    kw = dict(Hp=Hp, Hc=2, Mwt=[1.0, 0.0], Nwt=[0.1])
    orc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, **kw)
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    gpu = mpcqp.BatchLinMPC(rep(kf.Ah), rep(kf.Bhu), rep(kf.Ch), lib=lib, **kw)
    orc.setconstraint(ymax=[np.inf, 35.0]); gpu.setconstraint(ymax=[np.inf, 35.0])
    gpu.setestimator(rep(kf.Khat))
    ry = np.array([5.0, 0.0])
    worst, U, Y = 0.0, [], []
    for k in range(steps):
        y = plant.evaloutput()
        xh = kf.preparestate(y)
        gpu.preparestate(y)
        uo = orc.moveinput(xh, ry)
        ug = gpu.moveinput(None, ry)
        worst = max(worst, float(np.abs(ug - uo).max()))
        assert np.all(gpu.status == 0), gpu.status
        plant.updatestate(uo)
        kf.updatestate(uo, y)
        gpu.updatestate(np.tile(uo, (B, 1)), y)
        U.append(uo.copy()); Y.append(y.copy())
    return worst, np.array(U), np.array(Y), Hp, kf.nxh


def setmodel_after_first_step(lib=None, B=4, cfg=None):
    """`setmodel!` between two periods (src/controller/execute.jl:684-790): a controller that has already
    stepped gets a new plant model AND new weights; its next `moveinput!` must equal, bit for bit, the one
    of a controller constructed with the new model (same warm start), and match the oracle."""
    cfg = cfg or synth.C2
    bt1 = synth.make_batch(cfg, B, seed=21)
    bt2 = synth.make_batch(cfg, B, seed=22)
    a = make_controller(cfg, bt1, lib=lib)
    a.lastu0 = bt1["lastu0"].copy()
    a.moveinput(bt1["xhat0"], bt1["ry"])
    assert np.all(a.status == 0)
    Z1, lu1 = a.Z.copy(), a.lastu0.copy()
    a.setmodel(bt2["Ahat"], bt2["Bhu"], bt2["Chat"])
    a.setweights(Mwt=np.full(cfg.ny, 2.0 * cfg.Mwt), Nwt=np.full(cfg.nu, 0.5 * cfg.Nwt))
    ua = a.moveinput(bt2["xhat0"], bt2["ry"])
    b = mpcqp.BatchLinMPC(bt2["Ahat"], bt2["Bhu"], bt2["Chat"], Hp=cfg.Hp, Hc=cfg.Hc, Cwt=cfg.Cwt,
                          Mwt=np.full(cfg.ny, 2.0 * cfg.Mwt), Nwt=np.full(cfg.nu, 0.5 * cfg.Nwt),
                          Lwt=np.full(cfg.nu, cfg.Lwt), lib=lib)
    b.setconstraint(**constraint_kwargs(cfg))
    b.Z[:] = Z1                    # same warm start and last input as the controller that has history
    b.lastu0 = lu1.copy()
    ub = b.moveinput(bt2["xhat0"], bt2["ry"])
    assert np.all(a.status == 0) and np.array_equal(a.Z, b.Z) and np.array_equal(ua, ub)
    worst = 0.0
    nDU = cfg.nu * cfg.Hc
    for i in range(B):
        m = cd.LinMPCOracle(bt2["Ahat"][i], bt2["Bhu"][i], bt2["Chat"][i], Hp=cfg.Hp, Hc=cfg.Hc, Cwt=cfg.Cwt,
                            Mwt=np.full(cfg.ny, 2.0 * cfg.Mwt), Nwt=np.full(cfg.nu, 0.5 * cfg.Nwt),
                            Lwt=np.full(cfg.nu, cfg.Lwt))
        m.setconstraint(**constraint_kwargs(cfg, oracle=True))
        m.initpred(bt2["xhat0"][i], lu1[i], bt2["ry"][i])
        m.linconstraint()
        z, st, info = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
        worst = max(worst, np.abs(a.Z[i, :nDU] - z[:nDU]).max() / max(1.0, np.abs(z[:nDU]).max()))
    return worst


def multiple_shooting_known_answers(lib=None, B=2, Hp=1000):
    """The reference's MultipleShooting LinMPC tests (test/3_test_predictive_control.jl:120-127, 570-579):
    Hp = 1000, Hc = 1; u ≈ 1 and Ŷ[end] ≈ 15 for `linmodel` (gain 5 after the operating points); a second plant
    tf(5,[2,1]): u ≈ 3 for r = 15, and after setmodel!(tf(10,[2,1])) u ≈ 4 for r = 40 (atol 1e-2 there).  Also
    checks the returned MultipleShooting decision vector [ΔU; X̂0; ϵ]: X̂0 obeys the model equality constraints
    and reproduces Ŷ.  (`Hp`: the CPU emulator test runs a quarter of the horizon -- the plant settles within a few
    periods, the answers are the same and the horizon-long data still live in the HBM scratch placement; the GPU test runs
    the reference's 1000.)"""
    rep = lambda M: np.repeat(np.asarray(M, float)[None], B, 0)
    out = {}
    model = es.LinModelOracle(*es.tf1_zoh(5.0, 2.0, 3.0), Ts=3.0)
    kf = es.SteadyKalmanFilterOracle(model)
    mpc = mpcqp.BatchLinMPC(rep(kf.Ah), rep(kf.Bhu), rep(kf.Ch), Hp=Hp, Hc=1, Nwt=[0], transcription="MultipleShooting", lib=lib)
    u = mpc.moveinput(np.zeros((B, kf.nxh)), [15.0], want_info=True)
    info = mpc.getinfo()
    out["u3"] = u.copy()
    Z = info["Z̃"]
    nxh = kf.nxh
    assert Z.shape == (B, 1 + nxh * Hp + 1)
    X0 = Z[:, 1:1 + nxh * Hp].reshape(B, Hp, nxh)
    U0 = info["U"].reshape(B, Hp, 1)
    xprev = np.concatenate([np.zeros((B, 1, nxh)), X0[:, :-1]], axis=1)
    defect = X0 - (np.einsum("ij,btj->bti", kf.Ah, xprev) + np.einsum("ij,btj->bti", kf.Bhu, U0))
    out["defect"] = np.abs(defect).max()
    out["yerr"] = np.abs(np.einsum("ij,btj->bti", kf.Ch, X0).reshape(B, -1) - info["Ŷ"]).max()
    out["yend"] = info["Ŷ"][:, -1].copy()
    model2 = es.LinModelOracle(*es.tf1_zoh(10.0, 2.0, 3.0), Ts=3.0)
    kf2 = es.SteadyKalmanFilterOracle(model2)
    mpc.setmodel(rep(kf2.Ah), rep(kf2.Bhu), rep(kf2.Ch))
    out["u4"] = mpc.moveinput(np.zeros((B, kf.nxh)), [40.0]).copy()
    return out


def dense_weight_case(lib=None, B=3, which=("M", "N", "L"), seed=0):
    """Full Hermitian weight matrices (M_Hp coupling prediction steps, N_Hc coupling moves, L_Hp coupling
    inputs/steps: construct.jl:45-93, 837-845) against the oracle, with input bounds active and a soft output
    bound, a time-varying R̂u, different plants per controller.  Returns (worst rel ΔU error, kernel kind)."""
    cfg = synth.Config("dense-w", nx=3, nu=2, ny=2, Hp=7, Hc=3, umin=-0.7, umax=0.7, ymax=0.9)
    bt = synth.make_batch(cfg, B, seed=30 + seed)
    rng = np.random.default_rng(seed)
    nY, nDU, nU = cfg.ny * cfg.Hp, cfg.nu * cfg.Hc, cfg.nu * cfg.Hp

    def spd(n, scale, diag):
        R = rng.standard_normal((n, n))
        return scale * (R @ R.T) / n + np.diag(np.full(n, diag))

    kw = {}
    if "M" in which:
        kw["M_Hp"] = spd(nY, 0.8, 0.5)
    if "N" in which:
        kw["N_Hc"] = spd(nDU, 0.2, 0.1)
    if "L" in which:
        kw["L_Hp"] = spd(nU, 0.1, 0.05)
    mpc = mpcqp.BatchLinMPC(bt["Ahat"], bt["Bhu"], bt["Chat"], Hp=cfg.Hp, Hc=cfg.Hc, Cwt=cfg.Cwt, lib=lib, **kw)
    mpc.setconstraint(**constraint_kwargs(cfg))
    mpc.lastu0 = bt["lastu0"].copy()
    Ru = 0.3 * rng.standard_normal((B, nU))
    mpc.moveinput(bt["xhat0"], bt["ry"], Rhatu=Ru, want_info=True)
    info = mpc.getinfo()
    assert np.all(mpc.status == 0), mpc.status
    worst = 0.0
    for i in range(B):
        m = cd.LinMPCOracle(bt["Ahat"][i], bt["Bhu"][i], bt["Chat"][i], Hp=cfg.Hp, Hc=cfg.Hc, Cwt=cfg.Cwt, **kw)
        m.setconstraint(**constraint_kwargs(cfg, oracle=True))
        m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i], Rhatu=Ru[i])
        m.linconstraint()
        z, st, oinfo = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
        assert st == 0
        worst = max(worst, np.abs(mpc.Z[i, :nDU] - z[:nDU]).max() / max(1.0, np.abs(z[:nDU]).max()))
        Jo = 0.5 * z @ m.Ht @ z + m.qt @ z + m.r
        assert abs(info["J"][i] - Jo) <= 1e-6 * max(1.0, abs(Jo)), (info["J"][i], Jo)
    return worst, mpc.hd.kernel_kind()


def small_kernel_cases(lib=None, B=6, with_y=False):
    """The small-problem step kernel (nZ̃ <= 16, box + input-bound rows, four controllers per wavefront:
    csrc/mpcqp_small_bodies.h) against the oracle, member by member, over two periods (cold, then warm started):
    hard u / Δu box (C2), soft input bounds with an active ϵ, measured disturbances with a D̂ preview, a non-default
    move-blocking vector with time-varying R̂u, and a controller without any finite bound (ExplicitMPC closed form).
    Returns the worst relative ΔU error and the kernel kinds."""
    rng = np.random.default_rng(0)
    worst, kinds = 0.0, []
    yact = []                 # (with_y) per case: output-bound rows on their bound at the oracle's optima, largest slack ϵ
    cases = [
        dict(cfg=synth.C2, kw={}, con=constraint_kwargs(synth.C2)),
        dict(cfg=synth.Config("soft-u", nx=3, nu=2, ny=2, Hp=12, Hc=4, Cwt=1e3), kw={},
             con=dict(umin=[-0.25, -0.3], umax=[0.25, 0.3], c_umin=[1.0, 0.5], c_umax=[0.5, 1.0], Δumax=[0.2, 0.2])),
        dict(cfg=synth.Config("nb", nx=3, nu=2, ny=2, Hp=9, Hc=[2, 1, 3, 3], Cwt=np.inf), kw={}, con=dict(umin=[-0.5, -0.4], umax=[0.5, 0.6])),
        dict(cfg=synth.Config("free", nx=3, nu=3, ny=2, Hp=8, Hc=4, Cwt=np.inf), kw={}, con={}),
    ]
    if with_y:
        # output-bound rows on the small-problem kernel (variant HASY): soft band on C2 shapes with an active ϵ, a hard
        # horizon-long upper bound with +-Inf holes next to hard input bounds, soft y + soft u sharing the slack, ymin only
        inf = np.inf
        Ymax_holes = np.tile([0.8, inf], 10); Ymax_holes[[4, 5, 12]] = inf
        cases = [
            dict(cfg=synth.Config("C2-soft-y", nx=4, nu=2, ny=2, Hp=20, Hc=5, Cwt=1e5), kw={},
                 con=dict(umin=[-1.0, -1.0], umax=[1.0, 1.0], Δumin=[-0.5, -0.5], Δumax=[0.5, 0.5], ymin=[-0.15, -0.2], ymax=[0.15, 0.2])),
            dict(cfg=synth.Config("hard-Y-holes", nx=3, nu=2, ny=2, Hp=10, Hc=3, Cwt=np.inf), kw={},
                 con=dict(umin=[-2.0, -2.0], umax=[2.0, 2.0], Ymax=Ymax_holes)),
            dict(cfg=synth.Config("soft-y-soft-u", nx=3, nu=2, ny=3, Hp=12, Hc=4, Cwt=1e3), kw={},
                 con=dict(umin=[-0.25, -0.3], umax=[0.25, 0.3], c_umin=[1.0, 0.5], c_umax=[0.5, 1.0], ymax=[0.1, 0.2, inf],
                          c_ymax=[1.0, 0.3, 1.0], ymin=[-inf, -0.3, -0.2], c_ymin=[1.0, 2.0, 0.0])),
            dict(cfg=synth.Config("ymin-nb", nx=3, nu=1, ny=1, Hp=16, Hc=[2, 2, 4, 8], Cwt=1e4), kw={}, con=dict(ymin=[-0.05], Δumax=[0.3])),
            # terminal rows (x̂min / x̂max on x̂(k+Hp), setconstraint!, construct.jl:324-509): soft, next to a soft output bound;
            # and hard terminal rows alone (nx̂ = 3 + 1 states, the output integrator unbounded)
            dict(cfg=synth.Config("soft-terminal", nx=3, nu=2, ny=2, Hp=10, Hc=4, Cwt=1e4), kw={},
                 con=dict(umin=[-1.0, -1.0], umax=[1.0, 1.0], ymax=[0.3, inf], x̂min=[-0.05, -0.05, -inf, -inf, -inf],
                          x̂max=[0.05, inf, 0.08, inf, inf], c_x̂min=[1.0, 0.5, 1.0, 1.0, 1.0], c_x̂max=[2.0, 1.0, 1.0, 1.0, 1.0])),
            dict(cfg=synth.Config("hard-terminal", nx=3, nu=2, ny=1, Hp=8, Hc=3, Cwt=np.inf), kw={},
                 con=dict(umin=[-2.0, -2.0], umax=[2.0, 2.0], x̂min=[-0.1, -inf, -0.1, -inf], x̂max=[0.1, 0.1, inf, inf])),
        ]
    for case in cases:
        cfg, con = case["cfg"], case["con"]
        Hc = cfg.Hc
        bt = synth.make_batch(synth.Config(cfg.name, nx=cfg.nx, nu=cfg.nu, ny=cfg.ny, Hp=cfg.Hp, Hc=len(Hc) if isinstance(Hc, list) else Hc), B, seed=41)
        mk = dict(Hp=cfg.Hp, Hc=Hc, Cwt=cfg.Cwt, Mwt=np.full(cfg.ny, cfg.Mwt), Nwt=np.full(cfg.nu, cfg.Nwt), Lwt=np.full(cfg.nu, 0.05))
        mpc = mpcqp.BatchLinMPC(bt["Ahat"], bt["Bhu"], bt["Chat"], lib=lib, **mk)
        mpc.setconstraint(**con)
        ors = []
        for i in range(B):
            m = cd.LinMPCOracle(bt["Ahat"][i], bt["Bhu"][i], bt["Chat"][i], **mk)
            m.setconstraint(**{k.replace("Δ", "d").replace("x̂", "xhat"): v for k, v in con.items()})
            ors.append(m)
        lu = bt["lastu0"].copy()
        mpc.lastu0 = lu.copy()
        nDU = mpc.nDU
        nact, epsmax = 0, 0.0
        for period in range(2):
            x0 = bt["xhat0"] * (1.0 - 0.2 * period)
            Ru = 0.2 * rng.standard_normal((B, mpc.nU))
            mpc.moveinput(x0, bt["ry"], Rhatu=Ru)
            assert np.all(mpc.status == 0), (cfg.name, mpc.status)
            for i, m in enumerate(ors):
                m.initpred(x0[i], lu[i], bt["ry"][i], Rhatu=Ru[i])
                m.linconstraint()
                z, st, _ = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
                assert st == 0
                m.Zt = z
                if with_y:
                    ze = z[-1] if m.neps else 0.0
                    Y0 = m.Et @ z + m.F
                    lo = np.isfinite(m.Y0min) & (np.abs(Y0 - (m.Y0min - m.C_ymin * ze)) <= 1e-7)
                    hi = np.isfinite(m.Y0max) & (np.abs(Y0 - (m.Y0max + m.C_ymax * ze)) <= 1e-7)
                    xe = m.ext @ z + m.fx
                    lo_x = np.isfinite(m.x0min) & (np.abs(xe - (m.x0min - m.c_xmin * ze)) <= 1e-7)
                    hi_x = np.isfinite(m.x0max) & (np.abs(xe - (m.x0max + m.c_xmax * ze)) <= 1e-7)
                    nact += int(lo.sum() + hi.sum() + lo_x.sum() + hi_x.sum()); epsmax = max(epsmax, ze)
                worst = max(worst, np.abs(mpc.Z[i, :nDU] - z[:nDU]).max() / max(1.0, np.abs(z[:nDU]).max()))
                if mpc.neps:
                    worst = max(worst, abs(mpc.Z[i, -1] - z[-1]) / max(1.0, abs(z[-1])))
            lu = mpc.lastu0.copy()
        kinds.append(mpc.hd.kernel_kind())
        yact.append((nact, epsmax))
    if with_y:
        return worst, kinds, yact
    return worst, kinds


def shape_vs_cport(cfg, B=256, seed=11, lib=None):
    """One synthetic workload (modelpredictivecontrol.jl_amd/synth.Config: dimensions + constraint pattern) at B
    controllers through the C-ABI against the oracle's C port (same interior-point iteration on a dense Newton matrix):
    returns dict(kind, optimal fraction of both, iteration means, relative dU differences).  A step kernel with a wrong
    Newton matrix converges on its exact residuals -- slowly and with a few per cent of failed solves -- which is what
    the iteration means and the optimal fraction over a few hundred instances show (scripts/shape_sweep.py)."""
    from oracle import cport
    bt = synth.make_batch(cfg, B, seed=seed)
    hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=0 if np.isinf(cfg.Cwt) else 1,
                      flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START, lib=lib)
    hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt),
                   np.full(B, cfg.Cwt) if np.isfinite(cfg.Cwt) else None)
    full = lambda v, n: None if not np.isfinite(v) else np.full((B, n), float(v))
    hd.set_bounds(U0min=full(cfg.umin, hd.nU), U0max=full(cfg.umax, hd.nU), DUmin=full(cfg.dumin, hd.nDU),
                  DUmax=full(cfg.dumax, hd.nDU), Y0min=full(cfg.ymin, hd.nY), Y0max=full(cfg.ymax, hd.nY))
    kind = hd.prepare()
    Z = np.zeros((B, hd.nZ))
    _, st, it = hd.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
    Zc, _, stc, itc = cport.from_synth(cfg, bt).step(bt["xhat0"], bt["lastu0"], bt["ry"])
    nDU = hd.nDU
    err = np.max(np.abs(Z[:, :nDU] - Zc[:, :nDU]), axis=1) / np.maximum(1.0, np.max(np.abs(Zc[:, :nDU]), axis=1))
    out = dict(kind=kind, nZ=hd.nZ, ms=hd.last_step_ms(), optimal=float(np.mean(st == 0)), optimal_cport=float(np.mean(stc == 0)),
               iters=float(it.mean()), iters_cport=float(itc.mean()), err99=float(np.quantile(err, 0.99)), errmax=float(err.max()))
    hd.close()
    return out


def unstable_plant_members(B, rho=(1.12, 1.05), Hp=50, Hc=50, seed=77, nx=4, nu=2, ny=2):
    """B different UNSTABLE plants (two eigenvalues outside the unit circle, `rho`) with output integrators, long
    horizons Hp = Hc = 50: cond(H̃) of the condensed QP is 1e8 and beyond -- the case the reference's documentation sends to
    MultipleShooting (src/controller/construct.jl:855-866).  Returns per-member dicts with the augmented model, the
    constructor keywords, the constraints and one (x̂0, ry)."""
    mem = []
    for i in range(B):
        rg = np.random.default_rng([seed, i])
        lam = np.concatenate([np.array(rho), rg.uniform(0.5, 0.9, nx - len(rho))])
        Q, _ = np.linalg.qr(rg.standard_normal((nx, nx)))
        A = Q @ np.diag(lam) @ Q.T
        Bu = rg.standard_normal((nx, nu)) / np.sqrt(nx); C = rg.standard_normal((ny, nx)) / np.sqrt(nx)
        Ah = np.block([[A, np.zeros((nx, ny))], [np.zeros((ny, nx)), np.eye(ny)]])
        Bhu = np.vstack([Bu, np.zeros((ny, nu))]); Ch = np.hstack([C, np.eye(ny)])
        kw = dict(Hp=Hp, Hc=Hc, Mwt=np.ones(ny), Nwt=np.full(nu, 0.1), Lwt=np.zeros(nu), Cwt=1e5)
        con = dict(umin=[-2.0] * nu, umax=[2.0] * nu, dumin=[-0.5] * nu, dumax=[0.5] * nu, ymax=[1.5] * ny)
        mem.append(dict(Ah=Ah, Bhu=Bhu, Ch=Ch, kw=kw, con=con, x0=0.05 * rg.standard_normal(nx + ny),
                        ry=0.5 * rg.standard_normal(ny)))
    return mem


def run_unstable_plant(lib=None, B=4, transcription="MultipleShooting", rho=(1.12, 1.05), check=None, hp=False):
    """The unstable plants of `unstable_plant_members` through the C-ABI with the given transcription against the dense
    MultipleShooting oracle (oracle/ms.py; `hp`: its 60-digit solve).  Returns dict(err per member, cond of H̃, kernel kind,
    statuses, defect of the model equations of the returned X̂0)."""
    from oracle import ms as oms
    mem = unstable_plant_members(B, rho=rho)
    st = lambda f: np.stack([f(m) for m in mem])
    g = mpcqp.BatchLinMPC(st(lambda m: m["Ah"]), st(lambda m: m["Bhu"]), st(lambda m: m["Ch"]), lib=lib,
                          transcription=transcription, **mem[0]["kw"])
    c = mem[0]["con"]
    g.setconstraint(umin=c["umin"], umax=c["umax"], Δumin=c["dumin"], Δumax=c["dumax"], ymax=c["ymax"])
    g.moveinput(st(lambda m: m["x0"]), st(lambda m: m["ry"]))
    nDU = g.nDU
    errs, conds, certs = [], [], []
    for i in (range(B) if check is None else check):
        m = mem[i]
        o = oms.LinMPCOracleMS(m["Ah"], m["Bhu"], m["Ch"], **m["kw"]).setconstraint(**m["con"])
        o.moveinput(m["x0"], m["ry"], hp=hp)
        assert o.status == 0 and max(o.info["kkt_full"].values()) <= 1e-9, o.info["kkt_full"]
        certs.append(o.info["certificate"])
        z = o.Zt[:nDU]
        errs.append(float(np.abs(g.Z[i, :nDU] - z).max() / max(1.0, np.abs(z).max())))
        conds.append(float(np.linalg.cond(o.ss.Ht)))
    out = dict(err=np.array(errs), cond=np.array(conds), kind=g.kernel, status=g.status.copy(), iters=g.iters.copy(), cert=certs)
    if g.kernel == mpcqp.api.KERNEL_MS:
        out["defect"] = g.hd.get(mpcqp.api.GET_MS_DEFECT)
    return out


def large_problem_case(lib=None, B=2, nu=8, Hp=32):
    """A SingleShooting controller with nZ̃ = nu Hc + 1 > 256 (nu = 8, Hp = Hc = 32: 257 variables, 512 + 512 + 32 rows): no
    condensed kernel exists for it, the stage-structured kernel takes the handle whatever its transcription.  Returns
    (worst relative ΔU error vs the dense SingleShooting oracle over two periods, kernel kind, statuses)."""
    rng = np.random.default_rng(11)
    nx, ny = 1, 1
    A = np.array([[0.85]]); Bu = rng.standard_normal((nx, nu)) / np.sqrt(nu); C = np.array([[1.2]])
    Ah = np.block([[A, np.zeros((nx, ny))], [np.zeros((ny, nx)), np.eye(ny)]])
    Bhu = np.vstack([Bu, np.zeros((ny, nu))]); Ch = np.hstack([C, np.eye(ny)])
    kw = dict(Hp=Hp, Hc=Hp, Mwt=[1.0], Nwt=rng.uniform(0.05, 0.3, nu), Cwt=1e5)
    con = dict(umin=[-0.3] * nu, umax=[0.3] * nu, ymax=[0.5], dumin=[-0.1] * nu, dumax=[0.1] * nu)
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    g = mpcqp.BatchLinMPC(rep(Ah), rep(Bhu), rep(Ch), lib=lib, **kw)
    g.setconstraint(umin=con["umin"], umax=con["umax"], ymax=con["ymax"], Δumin=con["dumin"], Δumax=con["dumax"])
    assert g.nZ > 256
    o = cd.LinMPCOracle(Ah, Bhu, Ch, **kw).setconstraint(**con)
    x0 = rng.standard_normal(nx + ny)
    worst, sts = 0.0, []
    for k in range(2):
        ry = [1.0 if k == 0 else -0.4]
        g.moveinput(rep(x0), ry)
        uo = o.moveinput(x0, ry)
        assert o.status == 0
        worst = max(worst, float(rel_err(g.Z[:1], o.Zt[None, :], o.nDU).max()))
        sts.append(g.status.copy())
        x0 = Ah @ x0 + Bhu @ uo
    return worst, g.kernel, np.concatenate(sts)


def varying_softness_case(lib=None, B=2):
    """Horizon-long softness vectors (`C_umax`, `C_umin`, `C_ymax`, `C_Δumax`; construct.jl:446-483) on a controller with a
    move-blocking vector: `C_umax` VARIES inside the blocking intervals, so the input rows of an interval cannot be merged into
    their tightest one -- the handle is served by the stage-structured kernel (one input row per step), whatever its
    transcription.  The set point drives the inputs into their (soft) bounds.  Returns (worst relative ΔU error vs the dense
    oracle over two periods, kernel kind, statuses, ϵ of the first period)."""
    rng = np.random.default_rng(23)
    nx, nu, ny, Hp, Hc = 3, 2, 2, 10, [1, 2, 3, 4]
    lam = rng.uniform(0.5, 0.95, nx)
    Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    A = Q @ np.diag(lam) @ Q.T
    Bu = rng.standard_normal((nx, nu)); C = rng.standard_normal((ny, nx))
    Ah = np.block([[A, np.zeros((nx, ny))], [np.zeros((ny, nx)), np.eye(ny)]])
    Bhu = np.vstack([Bu, np.zeros((ny, nu))]); Ch = np.hstack([C, np.eye(ny)])
    kw = dict(Hp=Hp, Hc=Hc, Mwt=[1.0, 1.5], Nwt=[0.05, 0.1], Cwt=2e3)
    C_umax = rng.uniform(0.0, 1.5, nu * Hp) * (rng.random(nu * Hp) < 0.8)       # some steps hard, the others of different softness
    C_umin = rng.uniform(0.2, 1.0, nu * Hp)
    C_ymax = rng.uniform(0.5, 2.0, ny * Hp)
    C_dumax = rng.uniform(0.0, 1.0, nu * len(Hc))
    con = dict(umin=[-0.4, -0.5], umax=[0.35, 0.45], ymax=[0.8, 0.9], dumax=[0.3, 0.3])
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    g = mpcqp.BatchLinMPC(rep(Ah), rep(Bhu), rep(Ch), lib=lib, **kw)
    g.setconstraint(umin=con["umin"], umax=con["umax"], ymax=con["ymax"], Δumax=con["dumax"],
                    C_umax=C_umax, C_umin=C_umin, C_ymax=C_ymax, C_Δumax=C_dumax)
    o = cd.LinMPCOracle(Ah, Bhu, Ch, **kw).setconstraint(**con, C_umax=C_umax, C_umin=C_umin, C_ymax=C_ymax, C_dumax=C_dumax)
    x0 = 0.3 * rng.standard_normal(nx + ny)
    worst, sts, eps0 = 0.0, [], None
    for k in range(2):
        ry = [3.0, -2.5] if k == 0 else [-2.0, 2.0]
        g.moveinput(rep(x0), ry)
        uo = o.moveinput(x0, ry)
        assert o.status == 0
        worst = max(worst, float(rel_err(g.Z[:1], o.Zt[None, :], o.nDU).max()))
        sts.append(g.status.copy())
        eps0 = o.Zt[-1] if eps0 is None else eps0
        x0 = Ah @ x0 + Bhu @ uo
    return worst, g.kernel, np.concatenate(sts), eps0


def hessian_after_refit_case(lib=None, B=2, nu=4, ny=2, nx=3, Hp=40, seed=5):
    """K2 is skipped while a handle's problem does not fit the LDS of a CU, and whether it fits depends
    on the row groups.  The handle here gets bounds on every group (does not fit: stage-structured kernel), then its weights
    (K2 skipped), then the bounds are reduced to the input bounds (fits: condensed kernel).  nΔU > 64, so the step reads the
    packed H̃ -- which must have been computed by then.  Returns (kinds seen, lds bytes seen, worst relative ΔU difference to a
    handle that was given the final bounds straight away, max |H̃ - H̃_fresh|)."""
    rng = np.random.default_rng(seed)
    nxh = nx + ny
    def one():
        lam = rng.uniform(0.5, 0.95, nx)
        Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
        A = Q @ np.diag(lam) @ Q.T
        Bu = rng.standard_normal((nx, nu)) / np.sqrt(nx); Cm = rng.standard_normal((ny, nx)) / np.sqrt(nx)
        Ah = np.block([[A, np.zeros((nx, ny))], [np.zeros((ny, nx)), np.eye(ny)]])
        return Ah, np.vstack([Bu, np.zeros((ny, nu))]), np.hstack([Cm, np.eye(ny)])
    ms = [one() for _ in range(B)]
    Ah, Bh, Ch = (np.stack([m[i] for m in ms]) for i in range(3))
    Hc = Hp
    x0 = 0.5 * rng.standard_normal((B, nxh)); lu = 0.1 * rng.standard_normal((B, nu)); ry = rng.standard_normal((B, ny)) * 2.0

    def handle():
        hd = mpcqp.Handle(B, nxh, nu, ny, 0, Hp, Hc, neps=1, flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START, lib=lib)
        hd.set_model(mpcqp.colmajor(Ah), mpcqp.colmajor(Bh), mpcqp.colmajor(Ch))
        return hd
    full = lambda v, n: np.full((B, n), float(v))
    weights = lambda hd: hd.set_weights(np.full((B, hd.nY), 1.0), np.full((B, hd.nDU), 0.1), np.full((B, hd.nU), 0.05), np.full(B, 1e5))
    final = lambda hd: hd.set_bounds(U0min=full(-0.4, hd.nU), U0max=full(0.4, hd.nU))
    # the handle under test
    hd = handle()
    hd.set_bounds(U0min=full(-0.4, hd.nU), U0max=full(0.4, hd.nU), DUmin=full(-0.2, hd.nDU), DUmax=full(0.2, hd.nDU),
                  Y0min=full(-3.0, hd.nY), Y0max=full(3.0, hd.nY), C_dumin=full(1.0, hd.nDU), C_dumax=full(1.0, hd.nDU))
    lds, kinds = [hd.lds_bytes()], [hd.kernel_kind()]
    weights(hd)
    final(hd)
    lds.append(hd.lds_bytes()); kinds.append(hd.kernel_kind())
    Z = np.zeros((B, hd.nZ))
    _, st, _ = hd.step(x0, lu, ry, Z)
    H = hd.get(mpcqp.api.GET_HESSIAN)
    hd.close()
    # the same controllers, final bounds from the start
    h2 = handle()
    final(h2); weights(h2)
    Z2 = np.zeros((B, h2.nZ))
    _, st2, _ = h2.step(x0, lu, ry, Z2)
    H2 = h2.get(mpcqp.api.GET_HESSIAN)
    h2.close()
    assert (st == 0).all() and (st2 == 0).all(), (st, st2)
    return kinds, lds, float(rel_err(Z, Z2, nu * Hc).max()), float(np.abs(H - H2).max())
