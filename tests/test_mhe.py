"""CPU-side checks of the batched linear MovingHorizonEstimator: the HIP library
exports every symbol include/mpcqp_mhe.h declares, the header is plain C, the host mirror validates
arguments like the reference, and the kernel bodies (csrc/mhe_bodies.h), run on the CPU wave emulator of
tests/emu, reproduce oracle/mhe.py -- window bookkeeping, arrival covariance, both forms, every bound
class.  The GPU parity tests are in test_gpu_mhe.py."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import mpcqp
from mpcqp import mhe as pm
from mpcqp import synth
from tests import mhe_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emulib():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", d])
    return mpcqp.api.load_library(os.path.join(d, "libmpcqp_emu.so"))


def test_library_exports_every_declared_mhe_symbol():
    hdr = open(os.path.join(ROOT, "include", "mpcqp_mhe.h")).read()
    declared = set(re.findall(r"\b(mpcqp_mhe_[a-z_]+)\s*\(", hdr))
    assert declared == set(pm.EXPORTS)
    lib = ctypes.CDLL(mpcqp.DEFAULT_LIB)          # fails loudly if the HIP build is missing
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.run(["strings", "-a", mpcqp.DEFAULT_LIB], capture_output=True, text=True).stdout
    assert "k_mhe_step" in out


def test_mhe_header_is_plain_c(tmp_path):
    exe = str(tmp_path / "abi_mhe_client")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_mhe_client.c"), "-o", exe,
                           "-L", os.path.dirname(mpcqp.DEFAULT_LIB), "-lmpcqp",
                           "-Wl,-rpath," + os.path.dirname(mpcqp.DEFAULT_LIB)])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "15 mhe entry points" in out.stdout, out.stdout + out.stderr


def _tiny(**kw):
    base = dict(nx=2, nu=2, nym=2, nd=0, He=3)
    base.update(kw)
    return synth.MheConfig("tiny", **base)


def test_batchmhe_argument_checks(emulib):
    cfg = _tiny()
    bt = synth.make_mhe_batch(cfg, 2, seed=0)
    args = (bt["Ahat"], bt["Bhu"], bt["Chm"])
    with pytest.raises(ValueError, match="He"):
        pm.BatchMHE(*args, He=0, lib=emulib)
    with pytest.raises(ValueError, match="Cwt"):
        pm.BatchMHE(*args, He=3, Cwt=-1.0, lib=emulib)
    with pytest.raises(ValueError, match="inconsistent"):
        pm.BatchMHE(bt["Ahat"], bt["Bhu"][:, :3], bt["Chm"], He=3, lib=emulib)
    bm = pm.BatchMHE(*args, He=3, lib=emulib)
    with pytest.raises(ValueError, match="size"):
        bm.setconstraint(x̂min=[0.0, 0.0])                       # nx̂ = 4
    with pytest.raises(ValueError, match="infeasible"):
        bm.setconstraint(x̂min=np.ones(4), x̂max=np.zeros(4))
    with pytest.raises(ValueError, match="Cwt must be finite"):        # construct.jl: softness needs a slack variable
        bm.setconstraint(c_x̂min=np.ones(4))
    with pytest.raises(ValueError, match="non-negative"):
        pm.BatchMHE(*args, He=3, Cwt=1e3, lib=emulib).setconstraint(c_v̂max=[-1.0, 0.0])
    with pytest.raises(ValueError, match="Cwt must be finite"):        # window-long softness needs the slack variable too
        bm.setconstraint(C_x̂min=np.zeros(16))
    bs = pm.BatchMHE(*args, He=3, Cwt=1e3, lib=emulib)
    with pytest.raises(ValueError, match="size must be"):               # nx̂ (He + 1) = 16
        bs.setconstraint(C_x̂min=np.zeros(12))
    with pytest.raises(ValueError, match="non-negative"):
        bs.setconstraint(C_v̂max=-np.ones(6))
    with pytest.raises(ValueError, match="size"):
        bm.setconstraint(X̂min=np.zeros(12))                     # nx̂ (He + 1) = 16
    with pytest.raises(ValueError, match="ym size"):
        bm.preparestate(np.zeros(3))
    # more than 16 augmented states: one estimator no longer fits a DPP row
    big = synth.make_mhe_batch(_tiny(nx=15), 1, seed=0)
    with pytest.raises(mpcqp.MpcqpError, match="not supported"):
        pm.BatchMHE(big["Ahat"], big["Bhu"], big["Chm"], He=3, lib=emulib)


@pytest.mark.slow
@pytest.mark.parametrize("direct", [True, False])
def test_unconstrained_mhe_on_emulator_matches_oracle(emulib, direct):
    """No finite bound: one Newton step; growing then moving window, arrival covariance updates, nd > 0.
    (The oracle itself is pinned on MHE == KalmanFilter, tests/test_oracle_mhe.py.)"""
    cfg = _tiny(nd=1, direct=direct)
    bt = synth.make_mhe_batch(cfg, 6, seed=3)              # 6 estimators: a partly filled second wavefront
    rows, bm = mhe_util.run_periods(cfg, bt, 6, [0, 3, 5], lib=emulib, bounds={})
    for r in rows:
        assert np.all(r["status"] == 0) and r["ex"] <= 1e-12 and r["ew"] <= 1e-12 and r["ep"] <= 1e-13, r
    assert rows[-1]["Nk"] == cfg.He


@pytest.mark.slow
@pytest.mark.parametrize("kw", [dict(xabs=0.8), dict(wabs=0.25), dict(vabs=0.3), dict(xabs=1.0, vabs=0.5, direct=False)],
                         ids=["xhat", "what", "vhat", "xhat+vhat predictor form"])
def test_constrained_mhe_on_emulator_matches_oracle(emulib, kw):
    cfg = _tiny(**kw)
    bt = synth.make_mhe_batch(cfg, 4, seed=4)              # one wavefront
    rows, bm = mhe_util.run_periods(cfg, bt, 4, [0, 2, 3], lib=emulib)
    active = 0
    for r in rows:
        assert r["ostatus"] == [0, 0, 0]
        assert np.all(r["status"] == 0), r
        assert r["ex"] <= 2e-6 and r["ew"] <= 2e-6 and r["ep"] <= 1e-13, r      # interior point (gap 1e-12, no polish) vs exact active set: a weakly active bound leaves ~3e-7
        active += int(r["iters"].max() > 0)
    assert active == len(rows)                              # every period needed interior-point iterations
    info = bm.getinfo()
    tol = 1e-7
    if np.isfinite(cfg.xabs):
        assert np.abs(info["X̂"]).max() <= cfg.xabs + tol and np.abs(info["x̂arr"]).max() <= cfg.xabs + tol
    if np.isfinite(cfg.wabs):
        assert np.abs(info["Ŵ"]).max() <= cfg.wabs + tol
    if np.isfinite(cfg.vabs):
        assert np.abs(info["V̂"]).max() <= cfg.vabs + tol


@pytest.mark.slow
def test_soft_constraints_on_emulator_match_oracle(emulib):
    """Finite Cwt: one slack ε relaxes the rows with softness c > 0 (relaxX̂ / relaxV̂, construct.jl:1151-1288; the
    slack is an arrow border of the block-tridiagonal Newton matrix).  Hard and soft rows mixed, ε ends > 0."""
    cfg = _tiny(xabs=0.5, vabs=0.2, Cwt=1e3)
    bt = synth.make_mhe_batch(cfg, 4, seed=4)
    bounds = mhe_util.bounds_of(cfg)
    bounds.update(c_xhatmax=np.array([1.0, 0.5, 0.0, 0.0]), c_xhatmin=np.array([0.0, 1.0, 0.0, 0.0]),
                  c_vhatmin=np.array([1.0, 1.0]), c_vhatmax=np.array([0.5, 1.0]))
    rows, bm = mhe_util.run_periods(cfg, bt, 4, [0, 1, 3], lib=emulib, bounds=bounds)
    for r in rows:
        assert r["ostatus"] == [0, 0, 0] and np.all(r["status"] == 0)
        assert r["ex"] <= 2e-6 and r["ew"] <= 2e-6 and r["ee"] <= 2e-6, r
    assert rows[-1]["eps"].max() > 0.05                     # the constraints are being relaxed


@pytest.mark.slow
def test_reference_known_answers_through_the_product_on_emulator(emulib):
    for direct, r in mhe_util.reference_known_answers(lib=emulib, B=1, forms=(True,)).items():      # (both forms: GPU suite)
        assert r["x_at_op"] <= 1e-9 and r["y_hold"] <= r["tol"] and r["y_step"] <= r["tol"], (direct, r)


@pytest.mark.slow
@pytest.mark.parametrize("kw", [dict(nx=1, nu=1, nym=1, nd=0, He=1, xabs=0.7), dict(nx=2, nu=1, nym=1, nd=1, He=1, vabs=0.3, direct=False)],
                         ids=["He1", "He1 predictor"])
def test_horizon_of_one_on_emulator(emulib, kw):
    cfg = synth.MheConfig("he1", **kw)
    rows, _ = mhe_util.run_periods(cfg, synth.make_mhe_batch(cfg, 4, seed=3), 4, [0, 1, 3], lib=emulib)
    assert all(np.all(r["status"] == 0) and max(r["ex"], r["ew"]) <= 2e-6 and r["ep"] <= 1e-13 for r in rows)


def test_reference_constraint_violation_through_the_product_on_emulator(emulib):
    """test/2_test_state_estim.jl:1491-1539 through BatchMHE (kernel bodies on the CPU wave emulator), hard bounds
    (soft: GPU suite)."""
    got = mhe_util.reference_constraint_violation(False, lib=emulib, B=1)
    for k, v in got.items():
        assert v <= 1e-5, (k, v)


def test_reference_unfilled_window_through_the_product_on_emulator(emulib):
    assert mhe_util.reference_unfilled_window(True, lib=emulib, B=1) <= 1e-6


def test_setstate_keeps_windows_and_arrival_covariance(emulib):
    """setstate!(::MovingHorizonEstimator, x̂) overwrites x̂0 only (src/estimator/execute.jl:424-429) and raises when a
    covariance is offered (mhe/execute.jl:938-941): the product follows the oracle through a setstate in mid-window."""
    cfg = synth.MheConfig("setst", nx=2, nu=1, nym=2, nd=0, He=4, xabs=1.5)
    bt = synth.make_mhe_batch(cfg, 3, seed=5)
    Y, U, D = synth.make_mhe_data(cfg, bt, 7, seed=2)
    bm = mhe_util.make_product(cfg, bt, lib=emulib)
    ors = mhe_util.make_oracles(cfg, bt, [0, 1, 2])
    for k in range(7):
        xg = bm.preparestate(Y[k])
        xo = np.array([e.preparestate(Y[k][b]) for b, e in enumerate(ors)])
        assert np.abs(xg - xo).max() <= 2e-6 * max(1.0, np.abs(xo).max()), k
        bm.updatestate(U[k], Y[k])
        for b, e in enumerate(ors):
            e.updatestate(U[k][b], Y[k][b])
        if k == 2:
            Nk, Pb = bm.handle.Nk, bm.handle.get(pm.GET_PBAR).copy()
            xnew = 0.3 * np.ones((3, cfg.nxh))
            bm.setstate(xnew)
            for b, e in enumerate(ors):
                e.setstate(xnew[b])
            assert bm.handle.Nk == Nk and np.array_equal(bm.handle.get(pm.GET_PBAR), Pb)
            with pytest.raises(mpcqp.MpcqpError):
                bm.setstate(xnew, P̂=np.eye(cfg.nxh))


def test_window_long_bounds_on_emulator(emulib):
    """setconstraint!(estim; X̂min, ..., V̂max): stage-dependent bounds, growing then moving window, against the oracle."""
    ex, ew, active = mhe_util.window_long_bounds(lib=emulib, B=2, nper=9)
    assert active > 0
    assert ex <= 2e-6 and ew <= 2e-6, (ex, ew)


@pytest.mark.parametrize("seed", list(range(24)))
def test_randomised_families_on_emulator(seed, emulib):
    """Randomised MovingHorizonEstimator families (dimensions, form, horizon, bound classes, hard / soft: tests/mhe_util.random_family,
    the families of the GPU sweeps) through the product on the CPU wave emulator against oracle/mhe.py -- affordable since the
    emulator's lanes are fibers (tests/emu/emu_fiber.h)."""
    worst, compared, _ = mhe_util.random_family(seed, lib=emulib, B=2)
    assert compared > 0 and worst <= 1e-5, (worst, compared)


def test_window_long_softness_on_emulator(emulib):
    """setconstraint!(estim; C_x̂min, ..., C_v̂max) (construct.jl:937-1020): a softness per channel and stage (zero = hard on
    some rows), growing then moving window (the softness column is not truncated, transcription.jl:737-752), vs the oracle."""
    eps = []
    ex, ew, active = mhe_util.window_long_bounds(lib=emulib, B=2, nper=9, csoft=True, eps_seen=eps)
    assert active > 0 and max(eps) > 1e-6, (active, eps)
    assert ex <= 2e-6 and ew <= 2e-6, (ex, ew)


def test_window_long_softness_survives_a_bound_reupload_on_emulator(emulib):
    """BatchMHE.setmodel() re-uploads the bounds after C_x̂min ... were set: the window-long softness must stay in force
    (mpcqp_mhe_set_bounds / _window keep CLS_C), same answers as without the call."""
    eps = []
    ex, ew, active = mhe_util.window_long_bounds(lib=emulib, B=2, nper=9, csoft=True, eps_seen=eps, noop_setmodel=True)
    assert active > 0 and max(eps) > 1e-6, (active, eps)
    assert ex <= 2e-6 and ew <= 2e-6, (ex, ew)


def test_a_failed_estimator_next_to_solved_ones_in_one_wavefront_on_emulator(emulib):
    """One estimator of a wavefront fails (bounds its data cannot meet: status 2, finite open-loop estimate) while its
    neighbours are solved, and getinfo asks for V̂: the V̂ roll-out of write_outputs once ran a cross-lane product under the
    per-estimator `failed` condition -- the groups of the wavefront then disagreed on the sequence of cross-lane operations
    (the emulator's "stack smashing" of ADVICE r4).  Same scenario as tests/test_gpu_mhe.py::test_infeasible_estimator_fails_alone."""
    cfg = synth.MheConfig("inf", nx=2, nu=2, nym=2, nd=1, He=3, xabs=0.8, wabs=0.3)
    B = 6
    bt = synth.make_mhe_batch(cfg, B, seed=2)
    Y, U, D = synth.make_mhe_data(cfg, bt, 4)
    bm = mhe_util.make_product(cfg, bt, lib=emulib)
    ors = mhe_util.make_oracles(cfg, bt, range(B))
    failed = np.zeros(B, bool)
    for k in range(4):
        xg = bm.preparestate(Y[k], D[k])
        info = bm.getinfo()
        assert np.all(np.isfinite(info["V̂"])) and np.all(np.isfinite(xg))
        for b, e in enumerate(ors):
            xo = e.preparestate(Y[k][b], D[k][b])
            if e.status != 0:
                assert bm.status[b] == 2
                failed[b] = True
            elif not failed[b]:
                assert bm.status[b] == 0 and np.abs(xg[b] - xo).max() <= 2e-6 * max(1.0, np.abs(xo).max())
        bm.updatestate(U[k], Y[k], D[k])
        for b, e in enumerate(ors):
            e.updatestate(U[k][b], Y[k][b], D[k][b])
    assert failed.any() and not failed.all()


def test_reference_setmodel_through_the_product_on_emulator(emulib):
    """setmodel!(::MovingHorizonEstimator, model), test/2_test_state_estim.jl:1668-1718, through BatchMHE.setmodel
    (mpcqp_mhe_set_model + mpcqp_mhe_shift_windows)."""
    for k, (v, want) in mhe_util.reference_setmodel(lib=emulib, B=1).items():
        assert abs(v - want) <= 1e-3 * max(1.0, abs(want)), (k, v, want)
