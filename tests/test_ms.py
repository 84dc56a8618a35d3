"""MultipleShooting LinMPC without a GPU: the dense MultipleShooting oracle (oracle/ms.py) against the
SingleShooting oracle and the reference's known answers, and the stage-structured kernel body (csrc/ms_bodies.h) on the
CPU wave emulator against the oracle.  The GPU runs of the same cases are in tests/test_gpu_ms.py."""
import os
import subprocess

import numpy as np
import pytest

import mpcqp
from mpcqp import api
from oracle import condense as cd, estim as es, ms, qp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libmpcqp_emu.so")
TOL = 1e-5


@pytest.fixture(scope="module")
def emulib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    return api.load_library(EMU)


def _plant(rng, nx, nu, ny):
    A = rng.standard_normal((nx, nx)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    Bu, C = rng.standard_normal((nx, nu)), rng.standard_normal((ny, nx))
    Ah = np.block([[A, np.zeros((nx, ny))], [np.zeros((ny, nx)), np.eye(ny)]])
    return Ah, np.vstack([Bu, np.zeros((ny, nu))]), np.hstack([C, np.eye(ny)])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_multiple_shooting_oracle_equals_single_shooting_oracle(seed):
    """The two transcriptions describe one optimisation problem (the reference asserts the same answers from both,
    test/3_test_predictive_control.jl:120-127): ΔU*, ϵ* and J* of the dense MultipleShooting restatement (matrices of
    transcription.jl:196-240, 303-414 over Z = [ΔU; X̂0]) equal the SingleShooting oracle's; the returned X̂0 satisfies the
    equality constraints and the full-space KKT conditions hold."""
    rng = np.random.default_rng(seed)
    nx, nu, ny = 3, 2, 2
    Ah, Bhu, Ch = _plant(rng, nx, nu, ny)
    Hc = [2, 1, 3] if seed == 2 else 4                       # a move-blocking vector too (construct.jl:597-660)
    kw = dict(Hp=9, Hc=Hc, Mwt=[1, 2.0], Nwt=[0.1, 0.2], Lwt=[0.01, 0.0], Cwt=1e5)
    o1, o2 = cd.LinMPCOracle(Ah, Bhu, Ch, **kw), ms.LinMPCOracleMS(Ah, Bhu, Ch, **kw)
    con = dict(umin=[-1, -1], umax=[1, 1], ymax=[0.5, 0.6], dumin=[-0.3, -0.3], dumax=[0.3, 0.3], c_dumax=[0.5, 0.0],
               xhatmax=np.r_[np.full(nx, np.inf), 0.8, np.inf])
    o1.setconstraint(**con); o2.setconstraint(**con)
    assert o2.nZ == o2.nDU + o2.nxh * o2.Hp and o2.Aeq.shape == (o2.nxh * o2.Hp, o2.nZt)
    x0 = rng.standard_normal(nx + ny)
    for _ in range(3):
        ry = 2 * rng.standard_normal(ny)
        u1, u2 = o1.moveinput(x0, ry), o2.moveinput(x0, ry)
        assert o1.status == 0 and o2.status == 0
        assert np.abs(o1.Zt[:o1.nDU] - o2.Zt[:o2.nDU]).max() <= 1e-9
        if o1.neps:
            assert abs(o1.Zt[-1] - o2.Zt[-1]) <= 1e-9
        assert max(o2.info["kkt_full"].values()) <= 1e-10, o2.info["kkt_full"]
        i1, i2 = o1.getinfo(), o2.getinfo()
        assert abs(i1["J"] - i2["J"]) <= 1e-8 * max(1.0, abs(i1["J"]))
        assert np.abs(i1["Ŷ"] - i2["Ŷ"]).max() <= 1e-8 and np.abs(i1["x̂end"] - i2["x̂end"]).max() <= 1e-8
        x0 = Ah @ x0 + Bhu @ u1


def test_multiple_shooting_oracle_known_answers():
    """test/3_test_predictive_control.jl:570-579 on the dense MultipleShooting oracle (Hp = 1000, Hc = 1, Nwt = 0 like the
    reference: a 2000 x 2002 A_eq): tf(5,[2,1]) gives u ≈ 3 for r = 15 and, after the model change to tf(10,[2,1]), u ≈ 4
    for r = 40; Ŷ[end] reaches the set point (atol 1e-2 as there)."""
    for gain, r, uexp in ((5.0, 15.0, 3.0), (10.0, 40.0, 4.0)):
        model = es.LinModelOracle(*es.tf1_zoh(gain, 2.0, 3.0), Ts=3.0)
        kf = es.SteadyKalmanFilterOracle(model)
        o = ms.LinMPCOracleMS(kf.Ah, kf.Bhu, kf.Ch, Hp=1000, Hc=1, Nwt=[0.0])
        u = o.moveinput(np.zeros(kf.nxh), [r])
        assert o.status == 0 and abs(u[0] - uexp) <= 1e-2
        assert abs(o.getinfo()["Ŷ"][-1] - r) <= 1e-2


@pytest.mark.slow
@pytest.mark.parametrize("seed", [0, 1, 3, 4, 5, 7, 8, 9, 11, 13, 14, 15])
def test_multiple_shooting_kernel_on_the_emulator(seed, emulib):
    """The randomised controller families of the SingleShooting tests (dimensions, move blocking, ±Inf holes, hard / soft
    mixes, terminal bounds, measured disturbances, Cwt finite or Inf; seed 4 has no bound at all) with
    transcription=MultipleShooting: the stage-structured kernel body on the CPU wave emulator against the certified
    optimum, two periods each."""
    from tests.parity_util import run_random_case
    kinds = []
    e = run_random_case(seed, lib=emulib, B=2, small=True, kinds=kinds, transcription="MultipleShooting")
    assert e is not None and e <= TOL, e
    assert [k for k, _ in kinds] == [api.KERNEL_MS]


@pytest.mark.slow
@pytest.mark.parametrize("seed", [60, 278])
def test_polish_of_the_stage_structured_kernel_on_the_emulator(seed, emulib):
    """Two families of the full-size generator whose interior-point systems lose their accuracy below mu = 1e-8 (278: nine
    soft output rows at the barrier's cap -- without the polish the third member was returned OPTIMAL 1.2e-5 from the
    certified optimum; 60: nu = 4 > ny = 2, where the polish needs rho = 1e8): with the active-set polish every member and
    period is at the floor of the oracle comparison."""
    from tests.parity_util import run_random_case
    e = run_random_case(seed, lib=emulib, B=3, transcription="MultipleShooting")
    assert e is not None and e <= 1e-10, e


@pytest.mark.slow
def test_multiple_shooting_kernel_on_an_unstable_plant_on_the_emulator(emulib):
    """Hp = Hc = 50 on a plant with eigenvalues 1.12 and 1.05: cond(H̃) > 1e6 (here 1e8).  The Riccati recursion of the
    MultipleShooting kernel agrees with the dense MultipleShooting oracle far below the tolerance and returns an X̂0 that
    satisfies the model equations to rounding."""
    from tests.parity_util import run_unstable_plant
    r = run_unstable_plant(lib=emulib, B=2)
    assert r["kind"] == api.KERNEL_MS and np.all(r["status"] == 0)
    assert r["cond"].min() > 1e6
    assert r["err"].max() <= 1e-8, r
    assert r["defect"].max() <= 1e-12


def test_terminal_cost_is_lqr_on_the_multiple_shooting_kernel(emulib):
    """T6 (test/3_test_predictive_control.jl:498-527) with transcription = MultipleShooting: the block-diagonal
    M_Hp = blkdiag(I, I, P_DARE) runs on the stage-structured kernel (block weights in the stage cost, round 5) and the closed
    loop is the LQR's."""
    from tests.parity_util import run_lqr_terminal_cost
    kinds = []
    X_mpc, X_lqr = run_lqr_terminal_cost(lib=emulib, B=2, transcription="MultipleShooting", kinds=kinds)
    assert kinds == [api.KERNEL_MS]
    assert np.abs(X_mpc - X_lqr).max() <= 1e-8


def test_block_weights_on_the_multiple_shooting_kernel_match_the_oracle(emulib):
    """A block-diagonal M_Hp with off-diagonal entries inside every block, input bounds active: the stage-structured kernel
    against the dense oracle."""
    rng = np.random.default_rng(5)
    Ah, Bhu, Ch = _plant(rng, 3, 2, 2)
    rep = lambda M: np.repeat(np.asarray(M, float)[None], 2, 0)
    Hp = 6
    M = np.kron(np.eye(Hp), np.array([[2.0, 0.3], [0.3, 1.0]]))
    M[-2:, -2:] = [[5.0, -1.0], [-1.0, 3.0]]
    kw = dict(Hp=Hp, Hc=2, M_Hp=M, Nwt=[0.1, 0.1])
    mpc = mpcqp.BatchLinMPC(rep(Ah), rep(Bhu), rep(Ch), transcription="MultipleShooting", lib=emulib, **kw)
    mpc.setconstraint(umin=[-0.3, -0.3], umax=[0.3, 0.3])
    x0, ry = rng.standard_normal(5), 2.0 * rng.standard_normal(2)
    mpc.moveinput(rep(x0), ry)
    assert mpc.kernel == api.KERNEL_MS and np.all(mpc.status == 0)
    o = cd.LinMPCOracle(Ah, Bhu, Ch, **kw)
    o.setconstraint(umin=[-0.3, -0.3], umax=[0.3, 0.3])
    o.moveinput(x0, ry)
    assert np.abs(o.Zt[:o.nDU]).max() > 0.05
    assert np.abs(mpc.Z[0, :o.nDU] - o.Zt[:o.nDU]).max() <= 1e-7


def test_multiple_shooting_fallback_is_announced(emulib):
    """A MultipleShooting controller the stage-structured kernel does not take (here: an M_Hp that couples different prediction
    steps) keeps the SingleShooting kernels -- same optimal ΔU -- and says so."""
    rng = np.random.default_rng(5)
    Ah, Bhu, Ch = _plant(rng, 3, 2, 2)
    rep = lambda M: np.repeat(np.asarray(M, float)[None], 2, 0)
    Hp = 6
    M = np.kron(np.eye(Hp), np.array([[2.0, 0.3], [0.3, 1.0]]))
    M[0, 3] = M[3, 0] = 0.2                         # couples steps 1 and 2: a dense M_Hp
    kw = dict(Hp=Hp, Hc=2, M_Hp=M, Nwt=[0.1, 0.1])
    mpc = mpcqp.BatchLinMPC(rep(Ah), rep(Bhu), rep(Ch), transcription="MultipleShooting", lib=emulib, **kw)
    x0, ry = rng.standard_normal(5), rng.standard_normal(2)
    with pytest.warns(RuntimeWarning, match="MultipleShooting kernel not available"):
        mpc.moveinput(rep(x0), ry)
    assert mpc.kernel != api.KERNEL_MS
    o = cd.LinMPCOracle(Ah, Bhu, Ch, **kw)
    o.moveinput(x0, ry)
    assert np.abs(mpc.Z[0, :o.nDU] - o.Zt[:o.nDU]).max() <= 1e-8
    assert mpc.getinfo()["Z̃"].shape[1] == o.nDU + 5 * Hp + 1          # [ΔU; X̂0; ϵ]: the MultipleShooting layout all the same


def test_softness_that_varies_inside_a_blocking_interval_runs_on_the_stage_structured_kernel(emulib):
    """`C_umax` / `C_umin` of `setconstraint!` take any horizon-long vector (construct.jl:454-463).  The condensed kernels merge
    the input rows of a move-blocking interval, which needs one softness per interval: a vector that varies inside one is
    served by the stage-structured kernel -- same optimum as the dense oracle, slack active."""
    from tests.parity_util import varying_softness_case
    worst, kind, st, eps0 = varying_softness_case(lib=emulib)
    assert kind == api.KERNEL_MS and np.all(st == 0)
    assert eps0 > 1e-4                      # (the soft bounds are in play)
    assert worst <= TOL, worst


@pytest.mark.slow
def test_problems_beyond_256_variables_run_on_the_stage_structured_kernel(emulib):
    """nZ̃ = 257: the condensed kernels end at 256 variables (the Newton matrix must fit the LDS), the reference has no size
    limit (transcription.jl:2-4).  A SingleShooting handle of that size is served by the stage-structured kernel -- same QP,
    same optimal ΔU as the dense oracle -- and nothing is condensed for it."""
    from tests.parity_util import large_problem_case
    worst, kind, st = large_problem_case(lib=emulib, B=1)
    assert kind == api.KERNEL_MS and np.all(st == 0)
    assert worst <= TOL, worst
