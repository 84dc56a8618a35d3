"""MultipleShooting LinMPC on the GPU: the stage-structured kernel (csrc/ms_bodies.h, k_ms_step) through
the C-ABI against the certified oracle optimum / the dense MultipleShooting oracle (oracle/ms.py)."""
import json
import os

import numpy as np
import pytest

import mpcqp
from mpcqp import api, synth
from tests.parity_util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", list(range(12)) + [355, 2000, 2014])
def test_multiple_shooting_families_on_gpu(seed, hiplib):
    """The randomised controller families of test_gpu_parity.py (dimensions, move blocking, ±Inf holes, hard / soft mixes,
    terminal bounds, measured disturbances with preview, Cwt finite or Inf) with transcription=MultipleShooting: every
    member against the certified optimum over two periods, on the stage-structured kernel."""
    from tests.parity_util import run_random_case
    kinds = []
    e = run_random_case(seed, B=5, kinds=kinds, large=seed >= 2000, transcription="MultipleShooting")
    assert e is not None and e <= TOL, e
    assert [k for k, _ in kinds] == [api.KERNEL_MS]


def test_multiple_shooting_unstable_plant_on_gpu(hiplib):
    """unstable plants (eigenvalues 1.12, 1.05), Hp = Hc = 50, cond(H̃) ~ 1e8: 64 different plants on the
    MultipleShooting kernel, every one against the dense MultipleShooting oracle; the returned X̂0 satisfies the model
    equations to rounding."""
    from tests.parity_util import run_unstable_plant
    r = run_unstable_plant(B=64, check=range(0, 64, 4))
    assert r["kind"] == api.KERNEL_MS and np.all(r["status"] == 0), r["status"]
    assert r["cond"].min() > 1e6
    assert r["err"].max() <= 1e-7, r["err"]
    assert r["defect"].max() <= 1e-11


def test_multiple_shooting_beats_condensation_on_a_harder_unstable_plant(hiplib):
    """Eigenvalues 1.3 and 1.2 over 50 steps: entries of the condensed E reach 1.3^50 = 5e5 and cond(H̃) 1e13+, beyond what
    the float64 Cholesky of the condensed kernels can certify; the Riccati recursion is unaffected.  The oracle is the dense
    MultipleShooting QP solved in the ORTHONORMAL null space of the model equations (condition number 1e7-1e8 there),
    with the exact active-set certificate and the full-space KKT check."""
    from tests.parity_util import run_unstable_plant
    r = run_unstable_plant(B=8, rho=(1.3, 1.2))
    assert all(c == "active-set" for c in r["cert"])
    assert r["kind"] == api.KERNEL_MS and np.all(r["status"] == 0), r["status"]
    assert r["cond"].min() > 1e10
    assert r["err"].max() <= TOL, r["err"]


def test_terminal_cost_is_lqr_on_the_multiple_shooting_kernel_on_gpu(hiplib):
    """T6 (test/3_test_predictive_control.jl:498-527: M_Hp = blkdiag(I, I, P_DARE), Hp = Hc = 3 => the closed loop is the LQR's,
    atol 1e-5 there) with transcription = MultipleShooting on the stage-structured kernel (block weights, round 5)."""
    from tests.parity_util import run_lqr_terminal_cost
    kinds = []
    X_mpc, X_lqr = run_lqr_terminal_cost(B=64, transcription="MultipleShooting", kinds=kinds)
    assert kinds == [api.KERNEL_MS]
    assert np.abs(X_mpc - X_lqr).max() <= 1e-8


def test_transcriptions_agree_on_C3(hiplib):
    """BASELINE configs[2] shapes, 512 controllers: the MultipleShooting kernel and the condensed (SingleShooting)
    specialisation return the same ΔU, ϵ and Ŷ."""
    from tests.parity_util import make_controller
    cfg = synth.C3
    bt = synth.make_batch(cfg, 2048, seed=5)
    out = {}
    import warnings
    for tr in ("SingleShooting", "MultipleShooting"):
        mpc = make_controller(cfg, bt, transcription=tr)
        mpc.lastu0 = bt["lastu0"].copy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)      # (the @warn branch of a controller at the iteration limit)
            mpc.moveinput(bt["xhat0"], bt["ry"], want_info=True)
        out[tr] = (mpc.Z.copy(), mpc.getinfo()["Ŷ"], mpc.kernel, mpc.status.copy())
    assert out["SingleShooting"][2] == api.KERNEL_AOT and out["MultipleShooting"][2] == api.KERNEL_MS
    st_ss, st_ms = out["SingleShooting"][3], out["MultipleShooting"][3]
    assert np.all(st_ss == 0) and np.all(st_ms != api.STATUS_ERROR)
    # C3 is a heavily constrained workload (59 % of the input rows on a bound, multipliers of 1e5 on the soft rows).  The
    # stage-structured kernel (interior point + its own active-set polish since round 4) may leave a controller at its
    # iteration limit where the condensed kernel converges (3 of 8192 in the bench batch): such a solve is FLAGGED
    # (status 1, the reference's @warn branch keeps the iterate, execute.jl:491-496) and is adjudicated here against the
    # condensed optimum -- the kept iterate must still be that optimum to 1e-3; every solve called OPTIMAL agrees to TOL.
    # (round 6: held to the condensed kernels' bar -- every controller OPTIMAL and within TOL; rounds 4-5 accepted 0.5 % at the
    #  iteration limit with 1e-3.  Measured: 2048 of 2048 on this seed, 8192 of 8192 on the bench batch, worst difference
    #  3.8e-7: profiles/r6e/ms_c3_seed0.txt, profiles/r6c/stage_diag.txt)
    ok = st_ms == 0
    assert ok.all(), (ok.mean(), np.flatnonzero(~ok)[:8])
    nDU = cfg.nu * cfg.Hc
    assert rel_err(out["MultipleShooting"][0], out["SingleShooting"][0], nDU).max() <= TOL
    assert np.abs(out["MultipleShooting"][1] - out["SingleShooting"][1]).max() <= 1e-4


def test_multiple_shooting_known_answers_run_on_the_ms_kernel(hiplib):
    """test/3_test_predictive_control.jl:570-579 as it stands there (Hp = 1000, Hc = 1, Nwt = 0): u ≈ 3 for r = 15 on
    tf(5,[2,1]), then u ≈ 4 for r = 40 after setmodel!(tf(10,[2,1])); Ŷ[end] ≈ r.  A thousand stages: the horizon-long data
    live in the per-wavefront HBM scratch (k_ms_step_g)."""
    from oracle import estim as es
    B = 4
    rep = lambda M: np.repeat(np.asarray(M, float)[None], B, 0)
    kf = es.SteadyKalmanFilterOracle(es.LinModelOracle(*es.tf1_zoh(5.0, 2.0, 3.0), Ts=3.0))
    mpc = mpcqp.BatchLinMPC(rep(kf.Ah), rep(kf.Bhu), rep(kf.Ch), Hp=1000, Hc=1, Nwt=[0], transcription="MultipleShooting")
    u = mpc.moveinput(np.zeros((B, kf.nxh)), [15.0], want_info=True)
    assert mpc.kernel == api.KERNEL_MS and np.allclose(u, 3.0, atol=1e-2)
    assert np.allclose(mpc.getinfo()["Ŷ"][:, -1], 15.0, atol=1e-2)
    kf2 = es.SteadyKalmanFilterOracle(es.LinModelOracle(*es.tf1_zoh(10.0, 2.0, 3.0), Ts=3.0))
    mpc.setmodel(rep(kf2.Ah), rep(kf2.Bhu), rep(kf2.Ch))
    assert np.allclose(mpc.moveinput(np.zeros((B, kf.nxh)), [40.0]), 4.0, atol=1e-2)


def test_ill_conditioned_instances_against_extended_precision_optimum(hiplib):
    """VERDICT r3 item 1d.  tests/golden/hp_optima.json holds the optimum of seven long-horizon instances in 60-digit
    arithmetic (oracle/qp_hp.py; scripts/adjudicate_instances.py), among them instance 99 of shapes 8,2,2,60,40 and
    8,2,2,64,60 where the float64 answers of kernel and C port differed by 2.8e-4 with status OPTIMAL.  Adjudication: the
    oracle was right (6e-7 / 1e-5 from the optimum), the interior-point iteration on the float64 normal equations creeps
    on that instance (40 iterations of 6e-6 steps) and had passed its last-step test on a BLOCKED step.  Now: a solve the
    kernel calls OPTIMAL is within the tolerance of the extended-precision optimum; the creeping instance ends at the
    iteration limit (status 1, the reference's @warn branch) instead of claiming optimality."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "hp_optima.json")))["cases"]
    from tests.parity_util import make_controller
    seen = 0
    for shape in sorted({c["shape"] for c in cases}):
        cs = [c for c in cases if c["shape"] == shape]
        cfg = synth.get_config(shape)
        bt = synth.make_batch(cfg, 256, seed=cs[0]["seed"])
        mpc = make_controller(cfg, bt)
        mpc.lastu0 = bt["lastu0"].copy()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)          # (the @warn branch of the creeping instance)
            mpc.moveinput(bt["xhat0"], bt["ry"])
        assert mpc.kernel == api.KERNEL_ONDEMAND
        nDU = cfg.nu * cfg.Hc
        for c in cs:
            i, z = c["instance"], np.array(c["z"])
            e = rel_err(mpc.Z[i:i + 1], z[None, :], nDU).max()
            assert mpc.status[i] != api.STATUS_ERROR
            if mpc.status[i] == api.STATUS_OPTIMAL:
                assert e <= TOL, (shape, i, e)
            else:
                assert e <= 1e-3, (shape, i, e)
            seen += 1
        assert np.sum(mpc.status != 0) <= 2, np.flatnonzero(mpc.status)
    assert seen == len(cases)


def test_problems_beyond_256_variables_on_gpu(hiplib):
    """nZ̃ = 257 > 256 (SingleShooting): served by the stage-structured kernel; two periods against the dense oracle."""
    from tests.parity_util import large_problem_case
    worst, kind, st = large_problem_case(B=64)
    assert kind == api.KERNEL_MS and np.all(st == 0)
    assert worst <= TOL, worst


def test_softness_that_varies_inside_a_blocking_interval_on_gpu(hiplib):
    """Horizon-long `C_umax` / `C_umin` / `C_ymax` / `C_Δumax` (construct.jl:446-483), `C_umax` varying inside the move-blocking
    intervals: the handle runs on the stage-structured kernel (one input row per step); two periods against the dense oracle."""
    from tests.parity_util import varying_softness_case
    worst, kind, st, eps0 = varying_softness_case(B=64)
    assert kind == api.KERNEL_MS and np.all(st == 0)
    assert eps0 > 1e-4
    assert worst <= TOL, worst


@pytest.mark.parametrize("seed", [51, 60, 68, 278, 290, 2031])
def test_families_that_needed_the_polish(seed, hiplib):
    """Families of the round-4 sweep the interior-point iteration alone left 1e-6 ... 1.6e-5 from the certified optimum (or at
    its iteration limit): with the active-set polish of the stage-structured kernel every member is at 1e-7 or better."""
    from tests.parity_util import run_random_case
    e = run_random_case(seed, B=3, large=seed >= 2000, transcription="MultipleShooting")
    assert e is not None and e <= 1e-7, e
