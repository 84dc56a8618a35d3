"""Pin the CPU oracle on the reference's own known-answer tests.

The reference is Julia and cannot run in this image; these are the analytic answers and the one
printed golden its test-suite asserts for the LinMPC moveinput! path
(/root/reference/test/3_test_predictive_control.jl, ext/LinearMPCext.jl doctest).
"""
import numpy as np
import pytest
from scipy.linalg import solve_discrete_are

from oracle import condense as cd, estim as es, qp


def _mpc(model, **kw):
    skw = {k: kw.pop(k) for k in ("sigmaQ", "sigmaR", "sigmaQint_ym", "nint_ym") if k in kw}
    kf = es.SteadyKalmanFilterOracle(model, **skw)
    mpc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, uop=model.uop, yop=model.yop,
                          dop=model.dop, xhop=kf.xhop, fhop=kf.fhop, **kw)
    return kf, mpc


def test_T1_steady_state_gain():
    # test/3_test_predictive_control.jl:95-106
    A, B, C = es.tf1_zoh(5.0, 2.0, 3.0)
    model = es.LinModelOracle(A, B, C, Ts=3.0).setop(yop=[10])
    kf, mpc = _mpc(model, Nwt=[0], Hp=1000, Hc=1)
    kf.preparestate([10])
    u = mpc.moveinput(kf.x0, [15])
    assert u == pytest.approx([1], abs=1e-2)
    u = mpc.moveinput(kf.x0, [15], lastu=[-1])
    assert u == pytest.approx([1], abs=1e-2)
    info = mpc.getinfo()
    assert info["Ŷ"][-1] == pytest.approx(15, abs=1e-2)
    assert info["ΔU"] == pytest.approx([2.0], abs=1e-2)
    # Cwt = Inf variant (:107-110)
    kf2, mpc2 = _mpc(model, Nwt=[0], Cwt=np.inf, Hp=1000, Hc=1)
    kf2.preparestate([10])
    assert mpc2.moveinput(kf2.x0, [15]) == pytest.approx([1], abs=1e-2)


def test_T2_input_setpoint_tracking():
    # :111-114  Lwt=1, Mwt=Nwt=0, R̂u = 12  =>  u = 12
    A, B, C = es.tf1_zoh(5.0, 2.0, 3.0)
    model = es.LinModelOracle(A, B, C, Ts=3.0).setop(yop=[10])
    kf, mpc = _mpc(model, Mwt=[0], Nwt=[0], Lwt=[1], Hp=10, Hc=2)
    kf.preparestate([10])
    u = mpc.moveinput(kf.x0, [0], Rhatu=np.full(mpc.Hp, 12.0))
    assert u == pytest.approx([12], abs=1e-2)


def test_T3_move_blocking_pattern():
    # :135-140  Hc=[1,2,3,4], Hp=10  =>  diff(U)[[2,4,5,7,8,9]] == 0 (1-based), atol 1e-9
    A, B, C = es.tf1_zoh(5.0, 2.0, 3.0)
    model = es.LinModelOracle(A, B, C, Ts=3.0).setop(yop=[10])
    kf, mpc = _mpc(model, Hp=10, Hc=[1, 2, 3, 4], Nwt=[10])
    assert mpc.nb == [1, 2, 3, 4] and mpc.Hc == 4
    kf.preparestate([10])
    mpc.moveinput(kf.x0, [15])
    dU = np.diff(mpc.getinfo()["U"])
    assert dU[[1, 3, 4, 6, 7, 8]] == pytest.approx(np.zeros(6), abs=1e-9)
    assert np.all(np.abs(dU[[0, 2, 5]]) > 1e-6)
    # truncation rule of move_blocking (construct.jl:625-627)
    assert cd.move_blocking(10, [1, 2, 3, 6, 7]) == [1, 2, 3, 4]
    assert cd.move_blocking(10, 3) == [1, 1, 8]


def test_T4_infeasible_returns_shifted_warm_start():
    # :143-150  umin=+1 > umax=-1, Cwt=Inf  =>  error status, last solution shifted
    A1, B1, C1 = es.tf1_zoh(5.0, 2000.0, 3000.0)
    model = es.LinModelOracle(A1, B1, C1, Ts=3000.0)
    kf, mpc = _mpc(model, Hp=1, Hc=1, Cwt=np.inf)
    mpc.setconstraint(umin=[1.0], umax=[-1.0])
    kf.preparestate([0])
    mpc.Zt[:] = 0.25
    u = mpc.moveinput(kf.x0, [0])
    assert mpc.status == qp.INFEASIBLE
    assert np.all(mpc.Zt == 0.0)          # [Z̃[nu+1:end]; 0] with Hc = 1
    assert u == pytest.approx([0.0])


@pytest.mark.parametrize("soft", [True, False])
def test_T5_constraint_activation(soft):
    # :391-464  tf(2,[10,1]), Ts=3, Hp=50, Hc=5
    A, B, C = es.tf1_zoh(2.0, 10.0, 3.0)
    model = es.LinModelOracle(A, B, C, Ts=3.0)
    kf, mpc = _mpc(model, Hp=50, Hc=5, Cwt=1e5 if soft else np.inf)
    mpc.setconstraint(xhatmin=[-1e6, -np.inf], xhatmax=[1e6, np.inf])
    mpc.setconstraint(umin=[-10], umax=[10], dumin=[-15], dumax=[15], ymin=[-100], ymax=[100])
    if soft:
        mpc.setconstraint(c_xhatmin=[1, 1], c_xhatmax=[1, 1], c_umin=[0.1], c_umax=[0.1],
                          c_dumin=[0.1], c_dumax=[0.1], c_ymin=[1], c_ymax=[1])
    kf.preparestate([0])
    x0 = kf.x0
    mpc.setconstraint(umin=[-3], umax=[4])
    mpc.moveinput(x0, [-100]); assert mpc.getinfo()["U"] == pytest.approx(np.full(50, -3), abs=1e-1)
    mpc.moveinput(x0, [100]); assert mpc.getinfo()["U"] == pytest.approx(np.full(50, 4), abs=1e-1)
    mpc.setconstraint(umin=[-10], umax=[10])
    mpc.setconstraint(dumin=[-1.5], dumax=[1.25])
    mpc.moveinput(x0, [-100]); assert mpc.getinfo()["ΔU"] == pytest.approx(np.full(5, -1.5), abs=1e-1)
    mpc.moveinput(x0, [100]); assert mpc.getinfo()["ΔU"] == pytest.approx(np.full(5, 1.25), abs=1e-1)
    mpc.setconstraint(dumin=[-15], dumax=[15])
    mpc.setconstraint(ymin=[-0.5], ymax=[0.9])
    mpc.moveinput(x0, [-100]); assert mpc.getinfo()["Ŷ"] == pytest.approx(np.full(50, -0.5), abs=1e-1)
    mpc.moveinput(x0, [100]); assert mpc.getinfo()["Ŷ"] == pytest.approx(np.full(50, 0.9), abs=1e-1)
    mpc.setconstraint(ymin=[-100], ymax=[100])
    mpc.setconstraint(Ymin=np.r_[-0.5, np.full(49, -100.0)], Ymax=np.r_[0.9, np.full(49, 100.0)])
    mpc.moveinput(x0, [-10]); Y = mpc.getinfo()["Ŷ"]
    assert Y[0] == pytest.approx(-0.5, abs=1e-1) and Y[-1] == pytest.approx(-10, abs=1e-1)
    mpc.moveinput(x0, [10]); Y = mpc.getinfo()["Ŷ"]
    assert Y[0] == pytest.approx(0.9, abs=1e-1) and Y[-1] == pytest.approx(10, abs=1e-1)
    mpc.setconstraint(ymin=[-100], ymax=[100])
    mpc.setconstraint(xhatmin=[-1e-6, -np.inf], xhatmax=[1e-6, np.inf])
    mpc.moveinput(x0, [-100]); assert mpc.getinfo()["x̂end"][0] == pytest.approx(0, abs=1e-1)
    mpc.moveinput(x0, [100]); assert mpc.getinfo()["x̂end"][0] == pytest.approx(0, abs=1e-1)
    # ±Inf pattern is frozen after the first solve (construct.jl:549-551)
    with pytest.raises(RuntimeError):
        mpc.setconstraint(umin=[-np.inf])


def test_T6_terminal_cost_is_lqr():
    # :498-527  the tight (1e-5) analytic pin of condense + solve, dense M_Hp
    A = np.array([[0.5, -0.4], [0.6, 0.5]]); Bu = np.eye(2); C = np.eye(2)
    Q, R = np.eye(2), 0.5 * np.eye(2)
    P = solve_discrete_are(A, Bu, Q, R)
    K = np.linalg.solve(R + Bu.T @ P @ Bu, Bu.T @ P @ A)
    M_Hp = np.block([[np.eye(4), np.zeros((4, 2))], [np.zeros((2, 4)), P]])
    mpc = cd.LinMPCOracle(A, Bu, C, Hp=3, Hc=3, M_Hp=M_Hp, Nwt=[0, 0], Lwt=[0.5, 0.5])  # nint_ym=0
    X_mpc, X_lqr = np.zeros((2, 20)), np.zeros((2, 20))
    x = np.array([1.0, 1.0])
    for i in range(20):
        u = mpc.moveinput(x, [0, 0])
        X_mpc[:, i] = x
        x = A @ x + Bu @ u
    x = np.array([1.0, 1.0])
    for i in range(20):
        X_lqr[:, i] = x
        x = A @ x + Bu @ (-K @ x)
    assert np.abs(X_mpc - X_lqr).max() < 1e-5      # reference asserts atol 1e-5; oracle is ~1e-15
    assert np.abs(X_mpc - X_lqr).max() < 1e-12


def test_T7_unconstrained_is_explicit_mpc():
    # :1593-1630 / explicitmpc.jl:216: without finite bounds Z̃ = -H̃⁻¹ q̃, also under move blocking
    rng = np.random.default_rng(0)
    A = np.diag([0.9, 0.5, 0.2]); Bu = rng.standard_normal((3, 2)); C = rng.standard_normal((2, 3))
    model = es.LinModelOracle(A, Bu, C)
    kf, mpc = _mpc(model, Hp=30, Hc=[2, 3, 4, 21], Cwt=np.inf)
    x0 = rng.standard_normal(kf.nxh)
    mpc.moveinput(x0, [1.0, -2.0])
    assert mpc.Zt == pytest.approx(-np.linalg.solve(mpc.Ht, mpc.qt), rel=1e-12, abs=1e-12)


def test_T8_doctest_golden_17_577311():
    # ext/LinearMPCext.jl:255-269: the only printed golden of the path, 6 digits, end-to-end through
    # SKF correction + condensation + solve.
    A, B, C = es.tf1_zoh(2.0, 10.0, 1.0)
    assert (B[0, 0] / ((1 - A[0, 0]) / 0.1), C[0, 0]) == pytest.approx((0.5, 0.4))
    model = es.LinModelOracle(A, B, C, Ts=1.0)
    kf, mpc = _mpc(model, Hp=10, Hc=2, sigmaQ=[1], sigmaR=[1], sigmaQint_ym=[1])
    kf.preparestate([1.0])
    u = mpc.moveinput(kf.x0, [10.0])
    assert round(float(u[0]), 6) == 17.577311


def test_qp_certificate_and_bound():
    """The oracle's QP answer is certified: active-set KKT check or a rigorous error bound."""
    rng = np.random.default_rng(1)
    n, m = 6, 14
    L = rng.standard_normal((n, n)); H = L @ L.T + 0.1 * np.eye(n); q = 3 * rng.standard_normal(n)
    A = rng.standard_normal((m, n)); b = np.abs(rng.standard_normal(m)) * 0.3
    z, st, info = qp.solve_qp(H, q, A, b, np.full(n, -0.5), np.full(n, np.inf), return_info=True)
    assert st == qp.OPTIMAL and info["certificate"] == "active-set"
    G, h = qp.stack_constraints(A, b, np.full(n, -0.5), np.full(n, np.inf))
    assert max(info["kkt"].values()) < 1e-9
    # independent check: SLSQP from scipy lands on the same point
    from scipy.optimize import minimize
    res = minimize(lambda x: 0.5 * x @ H @ x + q @ x, np.zeros(n), jac=lambda x: H @ x + q,
                   constraints=[{"type": "ineq", "fun": lambda x: h - G @ x, "jac": lambda x: -G}],
                   method="SLSQP", options={"ftol": 1e-14, "maxiter": 500})
    assert np.abs(res.x - z).max() < 1e-6


def test_T9_custom_linear_constraints():
    # :466-495  Wy / Wu / Wd / Wr rows drive the whole horizon onto the custom bound (atol 1e-1 there)
    from tests.parity_util import custom_constraint_cases
    model, kf, cases = custom_constraint_cases()
    for kwW, wmin, wmax, checks in cases:
        mpc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, Hp=50, Hc=50, Nwt=[0], Cwt=np.inf,
                              uop=model.uop, yop=model.yop, dop=model.dop, xhop=kf.xhop, fhop=kf.fhop, **kwW)
        mpc.setconstraint(wmin=wmin, wmax=wmax)
        assert mpc.Wby.shape == (51, 51) and mpc.nW == 51             # repeatdiag(W, Hp+1), :55
        x0 = np.zeros(kf.nxh)                                         # preparestate!(mpc, [50], [30])
        for ry, key, want in checks:
            mpc.moveinput(x0, [ry], [30.0], lastu=[25.0])
            info = mpc.getinfo()
            assert np.all(np.abs(info[key] - want) < 1e-1), (kwW, ry, info[key][:5])
            assert np.all(info["W"] >= wmin[0] - 1e-6) and np.all(info["W"] <= wmax[0] + 1e-6)
    with pytest.raises(ValueError):
        mpc.setconstraint(wmin=[0, 0, 0])                            # DimensionMismatch, :358
    with pytest.raises(ValueError):
        cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, Hp=5, Wy=np.ones((2, 2)))   # :85
