"""Pins of the MHE oracle (oracle/mhe.py) on the reference's own tests of the linear
MovingHorizonEstimator:

* "MHE v.s. Kalman filters" (test/2_test_state_estim.jl:1750-1777): an unconstrained MHE with the
  KalmanFilter arrival covariance IS the Kalman filter -- He = 3, nint_ym = 0, six noisy periods,
  prediction form (direct = false) and current form (direct = true), atol 1e-6 in the reference;
* "MHE estimation and getinfo (LinModel)" (:1034-1075): estimates stay at the operating point, and
  the estimated outputs follow a step in the plant to 1e-3 / 1e-2 after 40 periods;
* "MHE constraint violation (LinModel)" (:1491-1539): the constraint rows (relaxX̂/Ŵ/V̂, linconstraint!) -- a bound that
  excludes the operating point on x̂, then ŵ, then v̂ puts the estimate / Ŵ / V̂ on the bound, hard (Cwt = Inf) and soft
  (Cwt = 1e5, every softness parameter on), He = 1, atol 5e-2 in the reference;
* "MHE estimation with unfilled window" (:1313-1337): integrator on the input, He = 3, both forms, ŷ = y to 1e-6;
* hard and soft bounds (setconstraint!, :1192-1260 checks activation of x̂, ŵ, v̂ bounds).
The plant of those tests (`sys`, test/0_test_module.jl) is rebuilt in a minimal realisation: the
equivalences do not depend on the state coordinates."""
import numpy as np
import pytest

from oracle import estim as es
from oracle import mhe


def _plant(Ts=400.0):
    """sys = [1.9/(1800s+1) x3 ; -0.74/(800s+1), 0.74/(800s+1), -0.74/(800s+1)], inputs (u1, u2, d)."""
    a1, a2 = np.exp(-Ts / 1800.0), np.exp(-Ts / 800.0)
    A = np.diag([a1, a2])
    B = np.array([[1 - a1, 1 - a1, 1 - a1], [-(1 - a2), 1 - a2, -(1 - a2)]])
    C = np.diag([1.90, 0.74])
    return es.LinModelOracle(A, B[:, :2], C, B[:, 2:], np.zeros((2, 1)), Ts=Ts)


@pytest.mark.parametrize("direct", [False, True])
def test_mhe_is_the_kalman_filter(direct):
    model = _plant().setop(uop=[10, 50], yop=[50, 30], dop=[20])
    rng = np.random.default_rng(3)
    kw = dict(nint_ym=[0, 0])
    if direct:
        kf = mhe.make_kalman_filter(model, direct=True, **kw)
        kf.preparestate([50, 30], [20])                    # P̂(-1|-1): the a-posteriori covariance
        est = mhe.MHEOracle(model, He=3, direct=True, P_0=kf.P, **kw)
        kf.updatestate([10, 50], [50, 30], [20])
    else:
        kf = mhe.make_kalman_filter(model, direct=False, **kw)
        est = mhe.MHEOracle(model, He=3, direct=False, **kw)
    Xm, Xk = [], []
    for i in range(6):
        y = np.array([50.0, 31.0]) + rng.standard_normal(2)
        Xm.append(est.preparestate(y, [25]).copy())
        Xk.append(kf.preparestate(y, [25]).copy())
        est.updatestate([11, 50], y, [25])
        kf.updatestate([11, 50], y, [25])
        assert est.status == 0
    assert np.abs(np.array(Xm) - np.array(Xk)).max() <= 1e-8


def test_mhe_known_answers_at_the_operating_point():
    model = _plant().setop(uop=[10, 50], yop=[50, 30], dop=[5])
    for direct in (True, False):
        # (the reference's realisation of `sys` has nx = 4, hence its default σQ = 1/nx = 0.25)
        est = mhe.MHEOracle(model, He=2, direct=direct, sigmaQ=[0.25, 0.25], sigmaP_0=[0.25, 0.25])
        est.preparestate([50, 30], [5])
        x = est.updatestate([10, 50], [50, 30], [5])
        assert np.abs(x).max() <= 1e-9 and np.abs(est.x0).max() <= 1e-9
        for _ in range(40):
            est.preparestate([50, 30], [5]); est.updatestate([11, 52], [50, 30], [5])
        est.preparestate([50, 30], [5])
        assert np.abs(est.evaloutput([5]) - [50, 30]).max() <= (1e-3 if direct else 1e-2)
        for _ in range(40):
            est.preparestate([51, 32], [5]); est.updatestate([10, 50], [51, 32], [5])
        est.preparestate([51, 32], [5])
        assert np.abs(est.evaloutput([5]) - [51, 32]).max() <= (1e-3 if direct else 1e-2)


def test_mhe_bounds_are_active():
    """Hard bound on an estimated state and a soft bound on the sensor noise: both hold at the optimum."""
    model = _plant().setop(uop=[10, 50], yop=[50, 30], dop=[5])
    est = mhe.MHEOracle(model, He=4, direct=True, Cwt=1e4)
    est.setconstraint(xhatmax=[0.1, np.inf, np.inf, np.inf], vhatmin=[-0.2, -0.2], c_vhatmin=[1.0, 1.0])
    hit = False
    for k in range(8):
        est.preparestate([53, 26], [5])
        est.updatestate([12, 48], [53, 26], [5])
        assert est.status == 0
        Nk = est.Nk
        X = est.X0.reshape(Nk, est.nxh)
        eps = est.Zt[0]
        assert X[:, 0].max() <= 0.1 + 1e-8 and est.x0arr[0] <= 0.1 + 1e-8
        hit = hit or abs(X[:, 0].max() - 0.1) <= 1e-6
        assert est.Vhat.min() >= -0.2 - eps - 1e-8 and eps >= -1e-12
    assert hit                                              # the hard state bound did limit the estimate


@pytest.mark.parametrize("soft", [True, False], ids=["soft", "hard"])
def test_reference_constraint_violation_known_answers(soft):
    """test/2_test_state_estim.jl:1491-1539 on the oracle: x̂ ≈ ±[1,1], Ŵ ≈ ±[1,1], V̂ ≈ ±[1,1] (atol 5e-2 there; the
    hard answers are exact, the soft ones give way by ε ~ 1e-4)."""
    from tests import mhe_util
    res = mhe_util.reference_constraint_violation(soft, oracle=True)
    assert set(res) == {"x̂min", "x̂max", "ŵmin", "ŵmax", "v̂min", "v̂max"}
    for k, v in res.items():
        assert v <= (5e-4 if soft else 1e-8), (k, v)


@pytest.mark.parametrize("direct", [True, False])
def test_reference_unfilled_window(direct):
    """test/2_test_state_estim.jl:1313-1337 on the oracle (atol 1e-6 in the reference)."""
    from tests import mhe_util
    assert mhe_util.reference_unfilled_window(direct, oracle=True) <= 1e-9


def test_reference_setmodel_known_answers():
    """"MHE set model", test/2_test_state_estim.jl:1668-1718, on the oracle (atol 1e-3 for the estimates there)."""
    from tests import mhe_util
    for k, (v, want) in mhe_util.reference_setmodel(oracle=True).items():
        assert abs(v - want) <= 1e-3 * max(1.0, abs(want)), (k, v, want)


def test_c_port_matches_the_numpy_oracle():
    """oracle/mhe_ref.c (the cpu_baseline of bench.py --config C5: the estimator period in the block-tridiagonal
    state-sequence form, like the GPU kernel) against oracle/mhe.py (the reference's condensed Z̃ = [x̂0arr; Ŵ], dense) on
    the same data: the estimate x̂0(k) of every period -- growing window, first full window, moving window with the arrival
    covariance corrected and advanced -- with state bounds that become active."""
    import numpy as np
    from mpcqp import synth
    from oracle import estim as es, mhe as om, mhe_cport
    cfg = synth.MheConfig("c-port check", nx=3, nu=2, nym=2, nd=0, He=4, xabs=0.8)
    B, nper = 3, 9
    bt = synth.make_mhe_batch(cfg, B, seed=1)
    Y, U, _ = synth.make_mhe_data(cfg, bt, nper, seed=1)
    xh, it, st = mhe_cport.run(bt, Y, U, cfg.He, cfg.xabs)
    assert np.all(st == 0)
    on_bound = False
    for b in range(B):
        e = om.MHEOracle(es.LinModelOracle(bt["A"][b], bt["Bu"][b], bt["C"][b]), He=cfg.He, direct=True,
                         sigmaQ=np.full(cfg.nx, cfg.sigmaQ), sigmaR=np.full(cfg.nym, cfg.sigmaR),
                         sigmaQint_ym=np.full(cfg.nym, cfg.sigmaQint), sigmaP_0=np.full(cfg.nx, cfg.sigmaP0),
                         sigmaPint_ym_0=np.full(cfg.nym, cfg.sigmaP0), nint_ym=[1] * cfg.nym)
        e.setconstraint(xhatmin=np.full(cfg.nxh, -cfg.xabs), xhatmax=np.full(cfg.nxh, cfg.xabs))
        for k in range(nper):
            x = e.preparestate(Y[k][b])
            e.updatestate(U[k][b], Y[k][b])
            assert e.status == 0
            assert np.abs(x - xh[k, b]).max() <= 1e-7, (b, k)
            on_bound = on_bound or np.abs(x).max() >= cfg.xabs - 1e-9
    assert on_bound
