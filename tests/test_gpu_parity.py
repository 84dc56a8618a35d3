"""GPU parity tests proper: the HIP path, called through the C-ABI (ctypes -> libmpcqp.so), against
the CPU oracle on the same seeded inputs.  Tolerance (north star):

    max_b ‖ΔU_gpu − ΔU_oracle‖∞ / max(1, ‖ΔU_oracle‖∞)  ≤  1e-5      (float64 path)

measured against the oracle's *certified optimum* -- the reference's default OSQP run is itself
only ~1e-3 accurate (SURVEY.md section 7 "hard parts").
"""
import os

import numpy as np
import pytest

import mpcqp
from mpcqp import synth
from oracle import condense as cd, estim as es, qp
from tests.parity_util import (make_controller, make_oracle, oracle_batch, rel_err, run_batch)

pytestmark = pytest.mark.gpu
TOL = 1e-5


def assert_specialised(kinds):
    """Every handle of a family ran on the kernel its shape is entitled to: a specialisation (ahead-of-time, on-demand
    or the small-problem kernel) whatever its size (round 5: up to nZ~ = 128 before, the runtime-dimension kernel beyond).  An on-demand kernel that
    mpcqp_prepare rejected would show up here as KERNEL_GENERIC (round 3: a rejected object passed its family test on
    the fallback); conftest.py also turns the library's fallback warning into an error for every GPU test."""
    assert kinds, "the family did not report its kernel"
    for kind, nZ in kinds:
        assert kind in (mpcqp.api.KERNEL_AOT, mpcqp.api.KERNEL_ONDEMAND, mpcqp.api.KERNEL_SMALL), (kind, nZ)


@pytest.mark.parametrize("name,B", [("C2", 256), ("C3", 192)])
def test_condensation_tables_match_oracle(name, B, hiplib):
    """K1/K2: Σ_m, K, H̃ and per-step F, q̃ against the dense restatement (a4, a8, a11)."""
    cfg = synth.CONFIGS[name]
    bt = synth.make_batch(cfg, B, seed=3)
    got = run_batch(cfg, bt, keep_qp=True)
    hd = got["mpc"].hd
    H = hd.get(mpcqp.GET_HESSIAN)
    S = hd.get(mpcqp.GET_STEPRESP)            # (B, Hp, nu, ny)
    K = hd.get(mpcqp.GET_KMAT)                # (B, nx̂, nY)
    F, q = hd.get(mpcqp.GET_FVEC), hd.get(mpcqp.GET_QTILDE)
    for i in range(0, B, 17):
        m = make_oracle(cfg, bt, i)
        m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i])
        assert np.abs(H[i] - m.Ht).max() <= 1e-12 * np.abs(m.Ht).max()
        assert np.abs(K[i].T - m.K).max() <= 1e-12 * max(1.0, np.abs(m.K).max())
        V = m.V.reshape(cfg.Hp, cfg.ny, cfg.nu)                     # V block t = Σ_t
        assert np.abs(S[i].transpose(0, 2, 1) - V).max() <= 1e-12 * max(1.0, np.abs(V).max())
        assert np.abs(F[i] - m.F).max() <= 1e-11 * max(1.0, np.abs(m.F).max())
        assert np.abs(q[i] - m.qt).max() <= 1e-11 * max(1.0, np.abs(m.qt).max())


@pytest.mark.parametrize("shape", [(3, 2, 2, 7, 3), (12, 4, 4, 30, 10), (5, 3, 2, 12, 12)])
def test_prediction_offset_table_with_fop_different_from_xop(shape, hiplib):
    """f̂op ≠ x̂op, one offset per member (successive linearisation, docs/src/manual/nonlinmpc.md:501): B = [Ĉ S(t)](f̂op − x̂op)
    and bx̂ of init_predmat (transcription.jl:184-192) -- MPCQP_GET_BVEC and the free response F directly, bx̂ through the
    optimum of a controller with a terminal bound.  (3, ..) and (5, ..) take the LDS form of K1, (12, ..) = C3's sizes
    (nx̂ = 16) the matrix-core form."""
    from tests.parity_util import offset_tables_case
    nx, nu, ny, Hp, Hc = shape
    eB, eF, eZ, bmax = offset_tables_case(nx=nx, nu=nu, ny=ny, Hp=Hp, Hc=Hc, B=5)
    assert bmax > 0.05
    assert eB <= 1e-12 * max(1.0, bmax) and eF <= 1e-11 * max(1.0, bmax) and eZ <= TOL, (eB, eF, eZ)


@pytest.mark.parametrize("name,B,seed", [("C2", 1024, 0), ("C3", 256, 0), ("C3", 256, 7)])
def test_step_matches_oracle(name, B, seed, hiplib):
    """Full moveinput! (a11-a15) on BASELINE configs[1] (full size) and configs[2] (sampled)."""
    cfg = synth.CONFIGS[name]
    bt = synth.make_batch(cfg, B, seed=seed)
    got = run_batch(cfg, bt)
    ref = oracle_batch(cfg, bt)
    assert np.all(got["status"] == mpcqp.STATUS_OPTIMAL)
    err = rel_err(got["Z"], ref["Z"], cfg.nu * cfg.Hc)
    cert = ref["certified"]
    assert cert.mean() > 0.95
    assert err[cert].max() <= TOL, f"max rel ΔU err {err[cert].max():.3e}"
    # the few oracle points without the exact-KKT certificate carry the oracle's rigorous error bound
    # (oracle/qp.py: error_bound) instead: the kernel's polished optimum must lie inside it
    assert err.max() <= TOL, f"max rel ΔU err over all instances {err.max():.3e}"
    assert np.abs(got["u"] - ref["u"]).max() <= TOL * max(1.0, np.abs(ref["u"]).max())


@pytest.mark.parametrize("name", ["4,2,2,12,8", "6,4,4,12,8", "6,4,3,20,12", "8,4,4,24,20"])
def test_slack_row_of_shapes_with_nDU_a_multiple_of_16(name, hiplib):
    """nu Hc = 16, 32, 48, 80: the ϵ row of the Newton matrix starts a 16-row tile of its own that the matrix-core
    passes of E'DE do not cover.  Round 3's overwrite mode left it uninitialised (twice the iterations, 5 % failed
    solves; the eight controllers of mpcqp_prepare's comparison did not show it): every instance against the oracle, on
    the specialised kernel, with the iteration count of the oracle's C port."""
    cfg = synth.get_config(name)
    assert (cfg.nu * cfg.Hc) % 16 == 0
    bt = synth.make_batch(cfg, 192, seed=3)
    got = run_batch(cfg, bt)
    assert got["mpc"].hd.kernel_kind() == mpcqp.api.KERNEL_ONDEMAND
    assert np.all(got["status"] == mpcqp.STATUS_OPTIMAL), np.unique(got["status"], return_counts=True)
    ref = oracle_batch(cfg, bt)
    err = rel_err(got["Z"], ref["Z"], cfg.nu * cfg.Hc)
    assert err.max() <= TOL, f"max rel ΔU err {err.max():.3e}"
    # the C port runs the same interior-point iteration on a dense Newton matrix: with a correct matrix the counts agree
    from oracle import cport
    _, _, st_c, it_c = cport.from_synth(cfg, bt).step(bt["xhat0"], bt["lastu0"], bt["ry"])
    assert np.all(st_c == 0)
    assert abs(got["iters"].mean() - it_c.mean()) <= 1.0, (got["iters"].mean(), it_c.mean())


@pytest.mark.parametrize("name,pattern", [("4,1,1,16,16", "c3"), ("6,2,3,32,31", "c3"), ("6,3,3,21,21", "all"), ("6,2,2,40,32", "yband"),
                                          ("8,4,4,24,20", "c3"), ("8,3,2,45,42", "all"), ("8,5,4,20,16", "box"),
                                          ("12,3,3,50,50", "c3"), ("12,2,2,70,70", "all"),
                                          # (round 6: the register-operand form of E'DE, MPCQP_ETDE_VREG, on every nu that divides 16 and
                                          #  on ny = 8 -- the BASELINE shapes and the random families only have nu = ny = 4 of these)
                                          ("6,2,4,20,12", "all"), ("8,8,4,10,8", "c3"), ("6,2,8,12,10", "all"), ("16,16,4,6,3", "c3"),
                                          ("6,1,4,20,20", "c3")])
def test_shapes_and_constraint_patterns_against_the_c_port(name, pattern, hiplib):
    """A few hundred instances per shape on its on-demand specialisation (nZ~ = 17 .. 127, around the one-row-per-lane
    limit, 16-multiples of nu*Hc, four constraint patterns; nZ~ = 151 and 141: beyond the 128 the specialisations stopped at
    before round 5, three rows per lane) against the oracle's C port: every instance optimal, the
    same optimum (99 % quantile; a single ill-conditioned instance may sit further at equal objective), the same number
    of iterations -- what a handful of instances per family cannot show (scripts/shape_sweep.py, profiles/r3/)."""
    import dataclasses
    from tests.parity_util import shape_vs_cport
    pats = {"c3": {}, "box": dict(ymax=np.inf, dumin=-0.2, dumax=0.2, Cwt=np.inf),
            "yband": dict(ymin=-1.0, ymax=1.0, umin=-np.inf, umax=np.inf, dumin=-0.3, dumax=0.3),
            "all": dict(ymin=-1.2, ymax=1.0, dumin=-0.4, dumax=0.4)}
    r = shape_vs_cport(dataclasses.replace(synth.get_config(name), **pats[pattern]), B=256)
    assert r["kind"] == mpcqp.api.KERNEL_ONDEMAND, r
    assert r["optimal"] == 1.0 and r["optimal_cport"] == 1.0, r
    assert r["errmax"] <= TOL, r
    assert abs(r["iters"] - r["iters_cport"]) <= 1.0, r


def test_condensed_problem_beyond_the_lds_runs_in_stage_form(hiplib):
    """nZ~ = 185 (nu = 4, Hc = Hp = 46): the condensed problem does not fit the 160 KB of LDS of a CU -- the handle keeps
    its SingleShooting transcription and its steps run on the stage-structured kernel (round 5: MPCQP_ERR_UNSUPPORTED before),
    same optimum as the oracle's C port."""
    from tests.parity_util import shape_vs_cport
    r = shape_vs_cport(synth.get_config("12,4,4,46,46"), B=256)
    assert r["kind"] == mpcqp.api.KERNEL_MS, r
    # (round 6: the bar of the condensed kernels -- every instance OPTIMAL, the worst one within TOL; round 5 accepted 98 % and
    #  the 99 % quantile.  Measured: 256 of 256 OPTIMAL, worst difference 1.1e-9, profiles/r6a/stage_diag.txt)
    assert r["optimal"] == 1.0 and r["optimal_cport"] == 1.0, r
    assert r["errmax"] <= TOL, r


def test_full_size_properties_C3(hiplib):
    """BASELINE configs[2] at full size (B = 65536): size-independent properties."""
    cfg = synth.C3
    B = 65536
    bt = synth.make_batch(cfg, B, seed=0)
    got = run_batch(cfg, bt)
    Z, st = got["Z"], got["status"]
    assert np.all(st == mpcqp.STATUS_OPTIMAL)
    nu, Hc, nDU = cfg.nu, cfg.Hc, cfg.nu * cfg.Hc
    # hard input bounds hold over the whole horizon (Pu = held cumulative sum)
    U0 = np.cumsum(Z[:, :nDU].reshape(B, Hc, nu), axis=1) + bt["lastu0"][:, None, :]
    assert U0.max() <= cfg.umax + 1e-9 and U0.min() >= cfg.umin - 1e-9
    # slack is non-negative and the soft output bound holds up to it: Ŷ ≤ ymax + ϵ
    eps = Z[:, -1]
    assert eps.min() >= -1e-12
    assert np.all(got["Yhat"] <= cfg.ymax + eps[:, None] + 1e-8)
    # batch-position independence + determinism: the same instances as a shard give the same bits
    lo, n = 40000, 512
    sub = {k: (v[lo:lo + n] if isinstance(v, np.ndarray) else v) for k, v in bt.items()}
    got2 = run_batch(cfg, sub)
    assert np.array_equal(got2["Z"], Z[lo:lo + n])
    # shard of the seeded generator is the same as slicing the big batch
    bt3 = synth.make_batch(cfg, n, seed=0, lo=lo)
    assert np.array_equal(bt3["Ahat"], sub["Ahat"])
    # sampled parity at full size: every sampled instance, certified by the oracle or not (test_step_matches_oracle)
    idx = np.arange(0, B, 1024)
    ref = oracle_batch(cfg, {k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in bt.items()})
    err = rel_err(Z[idx], ref["Z"], nDU)
    assert ref["certified"].mean() > 0.95
    assert err.max() <= TOL
    # every one of the 65536 instances against the oracle's C port (same iteration, dense algebra,
    # host threads): two independent float64 evaluations of the same optimum
    from oracle import cport
    Zc, u0c, stc, itc = cport.from_synth(cfg, bt).step(bt["xhat0"], bt["lastu0"], bt["ry"])
    assert np.all(stc == 0)
    dif = rel_err(Z, Zc, nDU)
    assert np.percentile(dif, 99.99) <= 0.2 * TOL
    assert dif.max() <= TOL                    # no instance is allowed to part from the C port by more than TOL
    # The C port is the kernel's twin, not an independent optimum: the instances where the two differ most (the
    # ill-conditioned tail -- slack > 1, multipliers ~1e6, up to 77 iterations) go to the independent oracle, ALL of
    # them: each must carry the oracle's exact active-set certificate and the kernel must be within TOL of it.
    hard = np.argsort(-dif)[:12]
    refh = oracle_batch(cfg, {k: (v[hard] if isinstance(v, np.ndarray) else v) for k, v in bt.items()})
    assert np.all(refh["certified"]), hard[~refh["certified"]]
    errh = rel_err(Z[hard], refh["Z"], nDU)
    assert errh.max() <= TOL, (hard[np.argmax(errh)], errh.max())


def test_c_client_runs_steps_on_gpu(tmp_path, hiplib):
    """tests/abi_c_client.c, plain C99, drives the full call sequence of the C-ABI on the GPU (raw
    float64 fixtures of both BASELINE shapes) and checks the optimum against the golden one."""
    import subprocess
    from tests.parity_util import build_c_client, write_c_fixture
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_c_client")
    build_c_client(exe, mpcqp.DEFAULT_LIB, root)
    for name in ("C2", "C3"):
        fx = str(tmp_path / f"{name}.bin")
        write_c_fixture(fx, name, 16)
        out = subprocess.run([exe, "run", fx], capture_output=True, text=True)
        # (C2 qualifies for the small-problem kernel, kind 3; C3 runs its ahead-of-time specialisation, kind 1)
        want = "kernel kind 3" if name == "C2" else "kernel kind 1"
        assert out.returncode == 0 and "run ok" in out.stdout and want in out.stdout, out.stdout + out.stderr


def test_multi_device_entry_points_on_gpu(hiplib):
    """mpcqp_multi_*: one batch over two shards (this box's single GPU twice) equals the single-handle
    run bit for bit; the device-side gather collects the shards' results on the root."""
    import torch
    cfg = synth.C3
    B = 1000
    bt = synth.make_batch(cfg, B, seed=6)
    ref = run_batch(cfg, bt, cold_start=True)
    mh = mpcqp.MultiHandle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, [0, 0, 0], neps=1,
                           flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START)
    mh.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    mh.set_weights(np.full((B, mh.nY), cfg.Mwt), np.full((B, mh.nDU), cfg.Nwt), np.full((B, mh.nU), cfg.Lwt), np.full(B, cfg.Cwt))
    mh.set_bounds(U0min=np.full((B, mh.nU), cfg.umin), U0max=np.full((B, mh.nU), cfg.umax), Y0max=np.full((B, mh.nY), cfg.ymax))
    assert mh.prepare() == mpcqp.KERNEL_AOT
    Z = np.zeros((B, mh.nZ))
    u0, st, it = mh.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
    assert np.array_equal(Z, ref["Z"]) and np.all(st == 0)
    assert [mh.shard(g) for g in range(3)] == [(0, 334), (334, 333), (667, 333)]


def _device_resident_multi(devices):
    """Inputs born on the root device -> mpcqp_multi_scatter_device -> mpcqp_step_device per shard ->
    mpcqp_multi_gather_device: equals the single-handle run bit for bit, nothing crosses PCIe in between."""
    import torch
    cfg = synth.C3
    B = 1000
    bt = synth.make_batch(cfg, B, seed=6)
    ref = run_batch(cfg, bt, cold_start=True)
    mh = mpcqp.MultiHandle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, devices, neps=1,
                           flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START)
    mh.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    mh.set_weights(np.full((B, mh.nY), cfg.Mwt), np.full((B, mh.nDU), cfg.Nwt), np.full((B, mh.nU), cfg.Lwt), np.full(B, cfg.Cwt))
    mh.set_bounds(U0min=np.full((B, mh.nU), cfg.umin), U0max=np.full((B, mh.nU), cfg.umax), Y0max=np.full((B, mh.nY), cfg.ymax))
    assert mh.prepare() == mpcqp.KERNEL_AOT
    root = 0
    dr = torch.device("cuda", devices[root])
    x_r, lu_r, ry_r = (torch.from_numpy(bt[k]).to(dr) for k in ("xhat0", "lastu0", "ry"))
    sh = []
    for g, dv in enumerate(devices):
        o, n = mh.shard(g)
        d = torch.device("cuda", dv)
        f = lambda *shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=d)
        sh.append(dict(x=f(n, cfg.nxh), lu=f(n, cfg.nu), ry=f(n, cfg.ny), Z=f(n, mh.nZ), u=f(n, cfg.nu),
                       st=f(n, dt=torch.int32), it=f(n, dt=torch.int32)))
    torch.cuda.synchronize()
    ptrs = lambda k: [s_[k].data_ptr() for s_ in sh]
    mh.scatter_device(root, x_r.data_ptr(), lu_r.data_ptr(), ry_r.data_ptr(), cfg.ny, ptrs("x"), ptrs("lu"), ptrs("ry"))
    for g, s_ in enumerate(sh):
        o, n = mh.shard(g)
        assert np.array_equal(s_["x"].cpu().numpy(), bt["xhat0"][o:o + n])
        mh.step_device_shard(g, s_["x"].data_ptr(), s_["lu"].data_ptr(), s_["ry"].data_ptr(), s_["Z"].data_ptr(),
                             s_["u"].data_ptr(), s_["st"].data_ptr(), iters=s_["it"].data_ptr())
    for dv in set(devices):
        torch.cuda.synchronize(dv)
    Z_r = torch.zeros((B, mh.nZ), dtype=torch.float64, device=dr)
    u_r = torch.zeros((B, cfg.nu), dtype=torch.float64, device=dr)
    st_r = torch.full((B,), -1, dtype=torch.int32, device=dr)
    mh.gather_device(root, ptrs("Z"), ptrs("u"), ptrs("st"), Z_r.data_ptr(), u_r.data_ptr(), st_r.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(Z_r.cpu().numpy(), ref["Z"]) and np.all(st_r.cpu().numpy() == 0)
    assert np.array_equal(u_r.cpu().numpy(), ref["u"])


def test_device_resident_scatter_step_gather_one_gpu(hiplib):
    _device_resident_multi([0, 0, 0])


def test_device_resident_scatter_step_gather_distinct_gpus(hiplib):
    """The same on distinct ordinals (peer copies over xGMI): needs a box with >= 2 GPUs."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU on this box")
    _device_resident_multi(list(range(min(n, 4))))


def test_fused_loop_equals_separate_steps_on_gpu(hiplib):
    """mpcqp_loop_device (one launch per control period) against kf_correct + step + kf_predict."""
    import torch
    from tests.parity_util import fused_loop_vs_separate_steps
    assert fused_loop_vs_separate_steps(B=64, periods=5, torch_device=torch.device("cuda", 0)) == 0.0
    # round 6: also on the stage-structured (MultipleShooting) kernel
    assert fused_loop_vs_separate_steps(B=64, periods=5, torch_device=torch.device("cuda", 0), multiple_shooting=True) == 0.0


def test_config4_batch_on_one_gpu(hiplib):
    """BASELINE configs[3]'s batch (B = 262144, C3 shapes) on ONE GPU: every instance OPTIMAL, the
    size-independent properties, sampled parity against the certified oracle, and shard = slice: the
    first and the last 32768 instances solved on their own (the 8-GPU sharding of config 4) give the
    same bits as inside the big batch."""
    cfg = synth.C3
    B = 262144
    bt = synth.make_batch(cfg, B, seed=3)
    got = run_batch(cfg, bt)
    Z, st = got["Z"], got["status"]
    # (round 4: a Newton step cut short by the boundary no longer passes the last-step test -- instance 120520 of this
    #  batch crept to the iteration limit instead of stopping on a blocked step; it is 3.2e-6 from the 50-digit optimum,
    #  scripts/nonoptimal_instances.py -- so: nobody fails, at most two of 262144 are flagged ITERATION_LIMIT)
    assert np.all(st != mpcqp.STATUS_ERROR) and (st != mpcqp.STATUS_OPTIMAL).sum() <= 2, np.flatnonzero(st)
    nu, Hc, nDU = cfg.nu, cfg.Hc, cfg.nu * cfg.Hc
    U0 = np.cumsum(Z[:, :nDU].reshape(B, Hc, nu), axis=1) + bt["lastu0"][:, None, :]
    assert U0.max() <= cfg.umax + 1e-9 and U0.min() >= cfg.umin - 1e-9
    eps = Z[:, -1]
    assert eps.min() >= -1e-12
    # soft output bound up to the slack.  Polished instances hold it to 1e-11; an instance the polish
    # could not take over (degenerate vertex) ends on the interior-point rule, primal residual
    # <= 1e-9 nh or stalled below 1e-7 nh: a handful in 262144
    viol = (got["Yhat"] - cfg.ymax - eps[:, None]).max(axis=1)
    assert viol.max() <= 1e-6 and (viol > 1e-8).sum() <= 4, (viol.max(), (viol > 1e-8).sum())
    idx = np.arange(0, B, 4096)
    ref = oracle_batch(cfg, {k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in bt.items()})
    err = rel_err(Z[idx], ref["Z"], nDU)
    assert ref["certified"].mean() > 0.9 and err[ref["certified"]].max() <= TOL
    for lo in (0, B - 32768):
        sub = synth.make_batch(cfg, 32768, seed=3, lo=lo)
        assert np.array_equal(sub["xhat0"], bt["xhat0"][lo:lo + 32768])
        assert np.array_equal(run_batch(cfg, sub)["Z"], Z[lo:lo + 32768])


def _pair(model_kw, mpc_kw, B=3, con=None):
    """One oracle LinMPC and a batch of B identical GPU controllers built from the same model."""
    kf = es.SteadyKalmanFilterOracle(model_kw["model"], **model_kw.get("skf", {}))
    m = model_kw["model"]
    orc = cd.LinMPCOracle(kf.Ah, kf.Bhu, kf.Ch, kf.Bhd, kf.Dhd, uop=m.uop, yop=m.yop, dop=m.dop,
                          xhop=kf.xhop, fhop=kf.fhop, **mpc_kw)
    rep = lambda a: np.broadcast_to(a, (B,) + a.shape).copy()
    gkw = {k: v for k, v in mpc_kw.items()}
    gpu = mpcqp.BatchLinMPC(rep(kf.Ah), rep(kf.Bhu), rep(kf.Ch),
                            rep(kf.Bhd) if m.nd else None, rep(kf.Dhd) if m.nd else None,
                            uop=m.uop, yop=m.yop, dop=m.dop, xhop=kf.xhop, fhop=kf.fhop, **gkw)
    return kf, orc, gpu


def _tf(gain, tau, Ts, **op):
    A, B, C = es.tf1_zoh(gain, tau, Ts)
    return es.LinModelOracle(A, B, C, Ts=Ts).setop(**op)


def test_T1_T2_known_answers_on_gpu(hiplib):
    kf, orc, gpu = _pair({"model": _tf(5.0, 2.0, 3.0, yop=[10])}, dict(Nwt=[0], Hp=1000, Hc=1))
    kf.preparestate([10])
    x = np.tile(kf.x0, (3, 1))
    u = gpu.moveinput(x, [15])
    assert u == pytest.approx(np.ones((3, 1)), abs=1e-2)
    u = gpu.moveinput(x, [15], lastu=[-1], want_info=True)
    assert u == pytest.approx(np.ones((3, 1)), abs=1e-2)
    info = gpu.getinfo()
    assert info["ΔU"] == pytest.approx(np.full((3, 1), 2.0), abs=1e-2)
    assert info["Ŷ"][:, -1] == pytest.approx(np.full(3, 15.0), abs=1e-2)
    orc.moveinput(kf.x0, [15], lastu=[-1])
    assert np.abs(gpu.Z[0] - orc.Zt).max() <= TOL
    kf, orc, gpu = _pair({"model": _tf(5.0, 2.0, 3.0, yop=[10])},
                         dict(Mwt=[0], Nwt=[0], Lwt=[1], Hp=10, Hc=2))
    u = gpu.moveinput(np.zeros((3, 2)), [0], Rhatu=np.full(10, 12.0))
    assert u == pytest.approx(np.full((3, 1), 12.0), abs=1e-2)


def test_T3_move_blocking_on_gpu(hiplib):
    kf, orc, gpu = _pair({"model": _tf(5.0, 2.0, 3.0, yop=[10])}, dict(Hp=10, Hc=[1, 2, 3, 4], Nwt=[10]))
    assert gpu.nb == [1, 2, 3, 4]
    x = np.zeros((3, 2))
    gpu.moveinput(x, [15], want_info=True)
    orc.moveinput(np.zeros(2), [15])
    dU = np.diff(gpu.getinfo()["U"], axis=1)
    assert np.abs(dU[:, [1, 3, 4, 6, 7, 8]]).max() <= 1e-9
    assert np.abs(gpu.Z[1] - orc.Zt).max() <= TOL * max(1, np.abs(orc.Zt).max())


def test_T4_infeasible_on_gpu(hiplib):
    kf, orc, gpu = _pair({"model": _tf(5.0, 2000.0, 3000.0)}, dict(Hp=1, Hc=1, Cwt=np.inf))
    gpu.setconstraint(umin=[1.0], umax=[-1.0])
    gpu.Z[:] = 0.25
    with pytest.warns(RuntimeWarning, match="terminated without solution"):
        u = gpu.moveinput(np.zeros((3, 2)), [0])
    assert np.all(gpu.status == mpcqp.STATUS_ERROR)
    assert np.all(gpu.Z == 0.0)            # shifted warm start, transcription.jl:1001-1004
    assert u == pytest.approx(np.zeros((3, 1)))


@pytest.mark.parametrize("soft", [True, False])
def test_T5_constraints_on_gpu(soft, hiplib):
    """Hard and soft u, Δu, y, time-varying Y and terminal x̂ bounds -- every row group of A."""
    kf, orc, gpu = _pair({"model": _tf(2.0, 10.0, 3.0)}, dict(Hp=50, Hc=5, Cwt=1e5 if soft else np.inf))
    base = dict(xhatmin=[-1e6, -np.inf], xhatmax=[1e6, np.inf], umin=[-10], umax=[10],
                dumin=[-15], dumax=[15], ymin=[-100], ymax=[100])
    gbase = dict(x̂min=[-1e6, -np.inf], x̂max=[1e6, np.inf], umin=[-10], umax=[10],
                 Δumin=[-15], Δumax=[15], ymin=[-100], ymax=[100])
    if soft:
        sc = dict(c_umin=[0.1], c_umax=[0.1], c_ymin=[1], c_ymax=[1])
        orc.setconstraint(**base, c_xhatmin=[1, 1], c_xhatmax=[1, 1], c_dumin=[0.1], c_dumax=[0.1], **sc)
        gpu.setconstraint(**gbase, c_x̂min=[1, 1], c_x̂max=[1, 1], c_Δumin=[0.1], c_Δumax=[0.1], **sc)
    else:
        orc.setconstraint(**base)
        gpu.setconstraint(**gbase)
    x = np.zeros((3, 2))

    def both(ry, okw, gkw):
        orc.setconstraint(**okw)
        gpu.setconstraint(**gkw)
        gpu.moveinput(x, [ry], want_info=True)
        orc.moveinput(np.zeros(2), [ry])
        assert np.all(gpu.status == 0)
        assert np.abs(gpu.Z[2, :5] - orc.Zt[:5]).max() <= TOL * max(1.0, np.abs(orc.Zt[:5]).max())
        return gpu.getinfo()

    assert both(-100, dict(umin=[-3], umax=[4]), dict(umin=[-3], umax=[4]))["U"] == pytest.approx(np.full((3, 50), -3), abs=1e-1)
    assert both(100, {}, {})["U"] == pytest.approx(np.full((3, 50), 4), abs=1e-1)
    both(0, dict(umin=[-10], umax=[10]), dict(umin=[-10], umax=[10]))
    assert both(-100, dict(dumin=[-1.5], dumax=[1.25]), dict(Δumin=[-1.5], Δumax=[1.25]))["ΔU"] == pytest.approx(np.full((3, 5), -1.5), abs=1e-1)
    assert both(100, {}, {})["ΔU"] == pytest.approx(np.full((3, 5), 1.25), abs=1e-1)
    both(0, dict(dumin=[-15], dumax=[15]), dict(Δumin=[-15], Δumax=[15]))
    assert both(-100, dict(ymin=[-0.5], ymax=[0.9]), dict(ymin=[-0.5], ymax=[0.9]))["Ŷ"] == pytest.approx(np.full((3, 50), -0.5), abs=1e-1)
    assert both(100, {}, {})["Ŷ"] == pytest.approx(np.full((3, 50), 0.9), abs=1e-1)
    tv = dict(Ymin=np.r_[-0.5, np.full(49, -100.0)], Ymax=np.r_[0.9, np.full(49, 100.0)])
    Y = both(-10, tv, tv)["Ŷ"]
    assert Y[:, 0] == pytest.approx(np.full(3, -0.5), abs=1e-1) and Y[:, -1] == pytest.approx(np.full(3, -10), abs=1e-1)
    both(0, dict(ymin=[-100], ymax=[100]), dict(ymin=[-100], ymax=[100]))
    both(-100, dict(xhatmin=[-1e-6, -np.inf], xhatmax=[1e-6, np.inf]), dict(x̂min=[-1e-6, -np.inf], x̂max=[1e-6, np.inf]))
    both(100, {}, {})
    with pytest.raises(RuntimeError):
        gpu.setconstraint(umin=[-np.inf])


def test_T7_unconstrained_and_T8_golden_on_gpu(hiplib):
    rng = np.random.default_rng(0)
    A = np.diag([0.9, 0.5, 0.2]); Bu = rng.standard_normal((3, 2)); C = rng.standard_normal((2, 3))
    kf, orc, gpu = _pair({"model": es.LinModelOracle(A, Bu, C)}, dict(Hp=30, Hc=[2, 3, 4, 21], Cwt=np.inf))
    x0 = rng.standard_normal(kf.nxh)
    gpu.moveinput(np.tile(x0, (3, 1)), [1.0, -2.0])
    orc.moveinput(x0, [1.0, -2.0])
    zexp = -np.linalg.solve(orc.Ht, orc.qt)                # ExplicitMPC closed form
    assert np.abs(gpu.Z[0] - zexp).max() <= 1e-9 * max(1.0, np.abs(zexp).max())
    assert np.all(gpu.iters == 0) and np.all(gpu.status == 0)
    # T8: doctest golden, SKF correction on the host (oracle), condense + solve on the GPU
    kf, orc, gpu = _pair({"model": _tf(2.0, 10.0, 1.0), "skf": dict(sigmaQ=[1], sigmaR=[1], sigmaQint_ym=[1])},
                         dict(Hp=10, Hc=2))
    kf.preparestate([1.0])
    u = gpu.moveinput(np.tile(kf.x0, (3, 1)), [10.0])
    assert [round(float(v), 6) for v in u[:, 0]] == [17.577311] * 3


def test_measured_disturbance_feedforward(hiplib):
    """nd > 0: G d0 + J D̂0 in F (execute.jl:252-255) with operating points on every signal."""
    rng = np.random.default_rng(5)
    A = np.diag([0.8, 0.6, 0.3]); Bu = rng.standard_normal((3, 2)); C = rng.standard_normal((2, 3))
    Bd = rng.standard_normal((3, 1)); Dd = rng.standard_normal((2, 1))
    model = es.LinModelOracle(A, Bu, C, Bd, Dd).setop(uop=[1.0, -2.0], yop=[5.0, 3.0], dop=[0.7])
    kf, orc, gpu = _pair({"model": model}, dict(Hp=12, Hc=3, Lwt=[0.1, 0.2]))
    for o in (orc, gpu):
        o.setconstraint(umin=[-0.5, -2.5], umax=[1.5, -1.0], ymax=[5.5, 3.5])
    x0 = 0.3 * rng.standard_normal(kf.nxh)
    Dhat = 0.7 + 0.2 * rng.standard_normal(12)
    gpu.initstate([1.0, -2.0]); orc.lastu0 = np.zeros(2)
    ug = gpu.moveinput(np.tile(x0, (3, 1)), [5.3, 2.8], [0.9], Dhat=Dhat, Rhatu=np.tile([1.1, -1.9], 12))
    uo = orc.moveinput(x0, [5.3, 2.8], [0.9], Dhat=Dhat, Rhatu=np.tile([1.1, -1.9], 12))
    assert np.abs(ug[1] - uo).max() <= TOL
    assert np.abs(gpu.Z[1] - orc.Zt)[:6].max() <= TOL * max(1.0, np.abs(orc.Zt[:6]).max())


def test_closed_loop_warm_start(hiplib):
    """20 control periods with the shifted warm start (a13) and lastu0 carried (a15)."""
    cfg = synth.C2
    B = 8
    bt = synth.make_batch(cfg, B, seed=11)
    gpu = make_controller(cfg, bt)
    orcs = [make_oracle(cfg, bt, i) for i in range(B)]
    gpu.lastu0 = bt["lastu0"].copy()
    x = bt["xhat0"].copy()
    for i in range(B):
        orcs[i].lastu0 = bt["lastu0"][i].copy()
    for k in range(20):
        ug = gpu.moveinput(x, bt["ry"])
        for i in range(B):
            uo = orcs[i].moveinput(x[i], bt["ry"][i])
            assert np.abs(ug[i] - uo).max() <= TOL
        x = np.einsum("bij,bj->bi", bt["Ahat"], x) + np.einsum("bij,bj->bi", bt["Bhu"], ug)


def test_abi_error_codes(hiplib):
    H = mpcqp.Handle
    with pytest.raises(mpcqp.MpcqpError, match="illegal"):
        H(4, 3, 1, 1, 0, Hp=5, Hc=6)                    # Hc > Hp
    with pytest.raises(mpcqp.MpcqpError, match="dimension"):
        H(4, 3, 1, 1, 0, Hp=5, Hc=2, nb=[1, 2])         # sum(nb) != Hp
    big = H(4, 3, 4, 1, 0, Hp=80, Hc=70)                # nZ > 256: no condensed kernel, the stage-structured kernel takes it
    assert big.nZ > 256 and big.kernel_kind() == mpcqp.api.KERNEL_MS
    h = H(4, 3, 1, 1, 0, Hp=5, Hc=2)
    with pytest.raises(mpcqp.MpcqpError, match="must be set before"):
        h.step(np.zeros((4, 3)), np.zeros((4, 1)), np.zeros((4, 5)), np.zeros((4, 3)))


def test_kalman_closed_loop_on_gpu(hiplib):
    """SteadyKalmanFilter preparestate!/updatestate! on the GPU around moveinput!.
    (a) the doctest golden u = 17.577311 with the correction step done on the device;
    (b) a 15-period closed loop of a C2 batch with per-instance Kalman gains and measurement
        noise, against the oracle estimator + controller."""
    kf, orc, gpu = _pair({"model": _tf(2.0, 10.0, 1.0), "skf": dict(sigmaQ=[1], sigmaR=[1], sigmaQint_ym=[1])},
                         dict(Hp=10, Hc=2))
    K = mpcqp.steady_kalman_gain(np.tile(kf.Ah, (3, 1, 1)), np.tile(kf.Ch, (3, 1, 1)), np.eye(2), np.eye(1))
    gpu.setestimator(K)
    gpu.preparestate([1.0])
    u = gpu.moveinput(None, [10.0])
    assert [round(float(v), 6) for v in u[:, 0]] == [17.577311] * 3
    # (b)
    cfg = synth.C2
    B = 6
    bt = synth.make_batch(cfg, B, seed=21)
    gpu = make_controller(cfg, bt)
    nxh = cfg.nxh
    Q = np.diag(np.r_[np.full(cfg.nx, 1.0 / cfg.nx), np.ones(cfg.ny)] ** 2)
    K = mpcqp.steady_kalman_gain(bt["Ahat"], bt["Chat"], Q, np.eye(cfg.ny))
    gpu.setestimator(K, xhat0=bt["xhat0"])
    gpu.lastu0 = bt["lastu0"].copy()
    orcs = [make_oracle(cfg, bt, i) for i in range(B)]
    xo = bt["xhat0"].copy()
    xp = bt["xhat0"].copy()                       # "plant" = the augmented model itself
    rng = np.random.default_rng(0)
    for i in range(B):
        orcs[i].lastu0 = bt["lastu0"][i].copy()
    for k in range(15):
        y = np.einsum("bij,bj->bi", bt["Chat"], xp) + 0.02 * rng.standard_normal((B, cfg.ny))
        gpu.preparestate(y)
        ug = gpu.moveinput(None, bt["ry"])
        for i in range(B):
            xo[i] = xo[i] + K[i] @ (y[i] - bt["Chat"][i] @ xo[i])
            uo = orcs[i].moveinput(xo[i], bt["ry"][i])
            assert np.abs(ug[i] - uo).max() <= TOL
            xo[i] = bt["Ahat"][i] @ xo[i] + bt["Bhu"][i] @ uo
        gpu.updatestate(ug, y)
        assert np.abs(gpu.xhat0 - xo).max() <= 1e-5 * max(1.0, np.abs(xo).max())
        xp = np.einsum("bij,bj->bi", bt["Ahat"], xp) + np.einsum("bij,bj->bi", bt["Bhu"], ug)


def test_readme_example_closed_loop_on_gpu(hiplib):
    """BASELINE config 0: the reference's README example (README.md:47-74), `sim!(mpc, 40, [5, 0])`
    with the estimator steps and moveinput! on the GPU, against the oracle loop; y2 rides its
    bound of 35 during the transient, y1 reaches the set point after the 20-sample delay."""
    from tests.parity_util import readme_example
    from tests.test_abi_and_host import _check_readme_example
    _check_readme_example(*readme_example(B=3))


def test_on_demand_specialisation(hiplib):
    """Dimensions outside the ahead-of-time list get a compile-time-dims kernel built at first use
    (csrc/mpcqp_spec.hip through the installation's hipcc, cached under lib/spec_cache): odd sizes
    (nu=3, ny=2, nx̂=7, Hp=12, Hc=4), every row group that can be specialised."""
    import glob
    import os
    cfg = synth.Config("odd", nx=5, nu=3, ny=2, Hp=12, Hc=4, umin=-0.8, umax=0.9, dumin=-0.5,
                       dumax=0.4, ymin=-1.5, ymax=1.2)
    B = 40
    bt = synth.make_batch(cfg, B, seed=9)
    got = run_batch(cfg, bt)
    ref = oracle_batch(cfg, bt)
    assert np.all(got["status"] == 0)
    err = rel_err(got["Z"], ref["Z"], cfg.nu * cfg.Hc)
    assert err[ref["certified"]].max() <= TOL
    cache = os.path.join(os.path.dirname(mpcqp.DEFAULT_LIB), "spec_cache")
    if os.environ.get("MPCQP_JIT", "1") != "0" and os.environ.get("MPCQP_FORCE_GENERIC", "0") != "1":
        assert glob.glob(os.path.join(cache, "spec_r*_3_2_7_12_4_1_*.so")), "specialisation was not built"
    # second handle of the same dimensions: served from the in-process / on-disk cache
    got2 = run_batch(cfg, bt)
    assert np.array_equal(got2["Z"], got["Z"])


def test_T6_terminal_cost_is_lqr_on_gpu(hiplib):
    """T6, the reference's tight analytic pin of condense + solve (block-diagonal M_Hp with the
    DARE solution as terminal weight => the MPC law is the LQR), test/3_test_predictive_control.jl:498-527
    (atol 1e-5 there), through mpcqp_set_output_weight_blocks on the device."""
    from tests.parity_util import run_lqr_terminal_cost
    X_mpc, X_lqr = run_lqr_terminal_cost(B=64)
    assert np.abs(X_mpc - X_lqr).max() < 1e-10


def test_maximum_size_nZ_64_on_gpu(hiplib):
    """Largest problem one wavefront holds: nZ~ = nu Hc + 1 = 64 (every lane owns a row of the
    factor, four 16-wide tiles / three panel updates in the Cholesky), hard and soft rows of every
    kind.  One more variable is rejected (test_abi_errors...)."""
    cfg = synth.Config("max", nx=6, nu=7, ny=3, Hp=12, Hc=9, umin=-0.7, umax=0.8, dumin=-0.45,
                       dumax=0.4, ymin=-1.6, ymax=1.3)
    B = 24
    bt = synth.make_batch(cfg, B, seed=5)
    got = run_batch(cfg, bt)
    assert got["Z"].shape[1] == 64
    ref = oracle_batch(cfg, bt)
    assert np.all(got["status"] == 0)
    err = rel_err(got["Z"], ref["Z"], cfg.nu * cfg.Hc)
    assert ref["certified"].sum() >= B // 2
    assert err[ref["certified"]].max() <= TOL


def test_custom_linear_constraints_on_gpu(hiplib):
    """Wy / Wu / Wd / Wr custom linear constraints (mpcqp_set_custom_constraints,
    mpcqp_set_custom_bounds): the reference's four known answers
    (test/3_test_predictive_control.jl:466-495) and a soft, mixed case against the oracle."""
    from tests.parity_util import run_custom_constraint_cases, run_soft_custom_constraints
    kinds = []
    assert run_soft_custom_constraints(B=33, kinds=kinds) <= 1e-6
    # (round 3: handles with custom rows get an on-demand specialisation with the rows compiled in, -DMPCQP_SPEC_NW,
    #  accepted by the comparison with the runtime-dimension kernel like every on-demand kernel)
    assert kinds == [mpcqp.api.KERNEL_ONDEMAND]
    assert run_custom_constraint_cases(B=5) <= TOL


def test_team_kernel_with_custom_rows_and_terminal_bound(hiplib):
    """Round 6: beyond one row per lane the on-demand kernels run a TEAM of wavefronts per controller (k_step_team); the
    helpers also take the custom-row, terminal-row and input-row parts of the Newton matrix.  nZ̃ = 121 (nu = 2, Hp = Hc = 60)
    with two soft custom rows per step, a soft terminal bound, u / y bounds and a measured disturbance, against the oracle."""
    from tests.parity_util import run_soft_custom_constraints
    kinds = []
    e = run_soft_custom_constraints(B=5, kinds=kinds, Hp=60, Hc=60, terminal=True, periods=2)
    assert kinds == [mpcqp.api.KERNEL_ONDEMAND], kinds
    assert e <= 1e-6, e
    # ... and with a move-blocking vector (60 intervals over Hp = 70: no zero blocks in front of the Sigma table, so the Toeplitz
    # products take their general forms on wavefront 0 alone while the matrix-core passes are still split over the team)
    kinds = []
    e = run_soft_custom_constraints(B=3, kinds=kinds, Hp=70, Hc=[1] * 55 + [3] * 5, terminal=True, periods=2)
    assert kinds == [mpcqp.api.KERNEL_ONDEMAND], kinds
    assert e <= 1e-6, e


def test_dual_warm_start_closed_loop_on_gpu(hiplib):
    """MPCQP_FLAG_WARM_DUAL in a noisy closed loop (C3, 512 controllers, 5 periods): the same optimum
    as the plain start at every period, in fewer iterations from the second period on."""
    from tests.parity_util import closed_loop_pair
    cfg = synth.C3
    bt = synth.make_batch(cfg, 512, seed=6)
    res = closed_loop_pair(cfg, bt, 5, warm_dual=True)
    nDU = cfg.nu * cfg.Hc
    for Za, Zb, ita, itb in res:
        assert rel_err(Zb, Za, nDU).max() <= TOL
    plain = np.mean([r[2].mean() for r in res[1:]])
    warm = np.mean([r[3].mean() for r in res[1:]])
    assert warm <= plain - 1.0, (plain, warm)


@pytest.mark.parametrize("seed", list(range(10)) + [355])   # 355: dual residual stalls at its float64 floor
def test_random_controller_families_on_gpu(seed, hiplib):
    """Randomly drawn dimensions (nu ≤ 4, ny ≤ 3, Hp ≤ 23), move-blocking vectors, bound patterns
    with ±Inf holes, hard/soft mixes, terminal bounds, measured disturbances with preview, finite or
    infinite Cwt: each family gets its own on-demand specialisation and is stepped twice (the second
    step from the shifted warm start) against the certified oracle optimum."""
    from tests.parity_util import run_random_case
    kinds = []
    e = run_random_case(seed, B=5, kinds=kinds)
    assert e is not None and e <= TOL, e
    assert_specialised(kinds)


def test_prediction_tables_on_the_matrix_cores_for_every_eligible_shape(hiplib):
    """K1 has two forms: LDS loops, and chains of v_mfma_f64_16x16x4 whose state stays in the accumulators (predmat_mfma,
    nx̂ >= 10 by default: C3 -- test_condensation_tables_match_oracle pins its tables at 1e-12).  The randomised families
    have nx̂ <= 9; here they run with MPCQP_K1_MFMA_MIN_NX=1 (own process: the threshold is read once), so the terminal
    tables (ex̂, kx̂, bx̂), the measured-disturbance tables (Gd, Xd) and ragged dimensions go through the matrix-core form."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, '.')\n"
            "from tests.parity_util import run_random_case\n"
            "errs = [run_random_case(s, B=4) for s in range(12)]\n"
            "assert all(e is not None for e in errs), errs\n"
            "print('WORST', max(errs))\n")
    env = dict(os.environ, MPCQP_K1_MFMA_MIN_NX="1")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    worst = float(out.stdout.split("WORST")[1].split()[0])
    assert worst <= TOL, worst


@pytest.mark.parametrize("seed", [2000, 2004, 2014, 2021, 2028, 2083])
def test_families_near_wave_limit(seed, hiplib):
    """Families drawn at 49 <= nZ̃ <= 64 (four 16-wide tiles on the matrix-core paths, the largest
    specialisations the library builds).  2028 (nu=2, ny=3, Hp=Hc=30) and 2083 (nu=3, ny=3, Hp=26,
    Hc=21, nZ̃ = 64) are the two that exposed the out-of-line cholesky()/EtDE_add() miscompilation
    (csrc/mpcqp_types.h, MPCQP_HD)."""
    from tests.parity_util import run_random_case
    kinds = []
    e = run_random_case(seed, B=3, large=True, kinds=kinds)
    assert e is not None and e <= TOL, e
    assert_specialised(kinds)


@pytest.mark.parametrize("seed", [3000, 3004, 3008, 3010])
def test_families_beyond_one_row_per_lane(seed, hiplib):
    """64 < nZ̃ <= ~130 (e.g. nu = 3, Hc = 35): the runtime-dimension kernel gives every lane several
    rows of the factorisation (Step::cholesky_big / solve_big); same families, same oracle, same
    tolerance as the one-row-per-lane kernels."""
    from tests.parity_util import run_random_case
    kinds = []
    e = run_random_case(seed, B=3, huge=True, kinds=kinds)
    assert e is not None and e <= TOL, e
    assert_specialised(kinds)


@pytest.mark.parametrize("seed", [4000, 4001, 4009, 4010])
def test_families_beyond_two_rows_per_lane(seed, hiplib):
    """130 < nZ̃ <= 165 (nZ̃ = 133, 148, 153, 157): the three-rows-per-lane specialisations of round 5 against the INDEPENDENT
    oracle (oracle/qp.py, exact-KKT certificate) -- VERDICT r5 weak 1: until round 6 these sizes were only compared with
    oracle/linmpc_ref.c, the kernel's twin (itself pinned on the oracle at nZ̃ = 106 / 141 / 151 in tests/test_oracle_c_port.py)."""
    from tests.parity_util import run_random_case
    kinds = []
    e = run_random_case(seed, B=3, huge2=True, kinds=kinds)
    assert kinds and kinds[0][1] > 130, kinds
    assert e is not None and e <= TOL, e
    assert_specialised(kinds)


@pytest.mark.parametrize("seed", [5001, 5005, 5006, 5007, 5010, 5011])
def test_random_families_with_four_outputs(seed, hiplib):
    """Round 6: E'DE takes its matrix-core operands from registers when ny is a multiple of 4 and nu divides 16
    (MPCQP_ETDE_VREG) -- shapes the other random families never draw (ny <= 3).  Families with ny = 4, nu in {1, 2, 4}, default move
    blocking or a blocking vector (the latter keeps the operands-from-LDS form), every bound pattern, against the independent oracle."""
    from tests.parity_util import run_random_case
    kinds = []
    e = run_random_case(seed, B=3, ny4=True, kinds=kinds)
    assert e is not None and e <= TOL, e
    assert_specialised(kinds)


def test_hessian_is_recomputed_when_relaxed_bounds_make_the_problem_fit(hiplib):
    """ADVICE r5 (medium), see the emulator twin in tests/test_abi_and_host.py: stage-structured kernel first (does not fit),
    a condensed kernel after the bounds were reduced, whose packed H̃ (nΔU > 64: it is read) must exist by then."""
    from tests.parity_util import hessian_after_refit_case
    kinds, lds, ez, eh = hessian_after_refit_case(B=8)
    assert kinds[0] == mpcqp.api.KERNEL_MS and kinds[1] != mpcqp.api.KERNEL_MS and lds[0] > 160 * 1024 >= lds[1], (kinds, lds)
    assert ez <= 1e-9 and eh == 0.0, (ez, eh)


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_horizon_wide_forms_on_gpu(seed, hiplib):
    """Time-varying Umin/Umax/Ymin/Ymax vectors with ±Inf holes, R̂y / R̂u / D̂ trajectories, a
    block-diagonal M_Hp with a dense terminal block and (odd seeds) custom linear constraints."""
    from tests.parity_util import run_random_case2
    kinds = []
    e = run_random_case2(seed, B=4, kinds=kinds)
    assert e is not None and e <= TOL, e
    assert_specialised(kinds)


def test_setmodel_after_first_step_on_gpu(hiplib):
    """setmodel! + new weights on a controller that has already stepped: K1 + K2 re-run, next step equals a
    freshly constructed controller's bit for bit and the oracle's optimum (C2 and C3 shapes)."""
    from tests.parity_util import setmodel_after_first_step
    assert setmodel_after_first_step(B=6, cfg=synth.C2) <= TOL
    assert setmodel_after_first_step(B=3, cfg=synth.C3) <= TOL


def test_multiple_shooting_known_answers_on_gpu(hiplib):
    """test/3_test_predictive_control.jl:570-579 (MultipleShooting, Hp = 1000, Hc = 1, setmodel!) through the C-ABI."""
    from tests.parity_util import multiple_shooting_known_answers
    r = multiple_shooting_known_answers(B=5)
    assert np.allclose(r["u3"], 3.0, atol=1e-2) and np.allclose(r["u4"], 4.0, atol=1e-2)
    assert np.allclose(r["yend"], 15.0, atol=1e-2) and r["defect"] <= 1e-9 and r["yerr"] <= 1e-8


@pytest.mark.expects_generic_fallback
def test_rejected_specialisation_falls_back_to_generic_kernel(hiplib, tmp_path, monkeypatch):
    """mpcqp_prepare checks a fresh on-demand kernel against the runtime-dimension kernel; one that fails the check
    (forced here with a negative tolerance, in a private cache directory) is renamed *.bad and the handle runs --
    correctly -- on the generic kernel."""
    import os
    monkeypatch.setenv("MPCQP_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("MPCQP_JIT_SELFTEST_TOL", "-1")
    # (Hc = 8: nZ̃ = 17, beyond the small-problem kernel, which needs no specialisation)
    cfg = synth.Config("reject", nx=3, nu=2, ny=2, Hp=9, Hc=8, umin=-0.7, umax=0.7, ymax=0.8)
    bt = synth.make_batch(cfg, 4, seed=5)
    got = run_batch(cfg, bt)
    assert got["mpc"].hd.kernel_kind() == mpcqp.api.KERNEL_GENERIC
    files = os.listdir(tmp_path)
    assert any(f.endswith(".so.bad") for f in files) and not any(f.endswith(".ok") for f in files)
    ref = oracle_batch(cfg, bt)
    assert np.all(got["status"] == 0) and rel_err(got["Z"], ref["Z"], cfg.nu * cfg.Hc).max() <= TOL
    # the same shape with the check enabled normally: accepted
    monkeypatch.setenv("MPCQP_JIT_SELFTEST_TOL", "1e-6")
    for f in files:
        os.remove(os.path.join(tmp_path, f))
    cfg2 = synth.Config("accept", nx=3, nu=2, ny=2, Hp=10, Hc=8, umin=-0.7, umax=0.7, ymax=0.8)
    bt2 = synth.make_batch(cfg2, 4, seed=5)
    got2 = run_batch(cfg2, bt2)
    assert got2["mpc"].hd.kernel_kind() == mpcqp.api.KERNEL_ONDEMAND and any(f.endswith(".ok") for f in os.listdir(tmp_path))


@pytest.mark.parametrize("which", [("N",), ("M",), ("L",), ("M", "N", "L")], ids=["N_Hc", "M_Hp", "L_Hp", "all"])
def test_dense_weight_matrices_on_gpu(hiplib, which, monkeypatch):
    """Full Hermitian M_Hp (coupling prediction steps), N_Hc, L_Hp through the C-ABI vs the oracle (ΔU and J)."""
    from tests.parity_util import dense_weight_case
    monkeypatch.setenv("MPCQP_SMALL_Y", "1")      # (the dense-N case on the small-problem kernel whatever the batch size)
    worst, kind = dense_weight_case(B=5, which=which)
    assert worst <= TOL, worst
    # (a dense M_Hp / L_Hp handle gets an on-demand variant of its own that carries the dense gradient products,
    #  accepted by mpcqp_prepare's comparison with the runtime-dimension kernel; a dense N_Hc only enters H̃, so this
    #  nZ̃ = 7 controller with its output bound runs on the small-problem kernel since round 4)
    assert kind == (mpcqp.api.KERNEL_SMALL if which == ("N",) else mpcqp.api.KERNEL_ONDEMAND)


def test_audit_of_the_convergence_test(hiplib):
    """mpcqp_get(MPCQP_GET_AUDIT): the residuals behind every OPTIMAL are visible to the caller."""
    cfg = synth.C3
    bt = synth.make_batch(cfg, 256, seed=2)
    got = run_batch(cfg, bt)
    au = got["mpc"].hd.audit()
    assert np.all(got["status"] == 0)
    pol = au["polished"]
    assert pol.mean() > 0.8                                        # most C3 instances end in an accepted polish
    ipm = ~pol                                                     # the others ended on the interior-point criterion
    assert np.all(au["mu"][ipm] <= 1e-12) and np.all(au["rp"][ipm] <= 1e-9)
    assert np.all(np.isfinite(au["rd"])) and np.all(au["rd"] <= 1e-6)


def test_small_problem_kernel_on_gpu(hiplib):
    """Four controllers per wavefront (nZ̃ <= 16, box + input-bound rows): C2, soft input bounds, move blocking with R̂u,
    no bounds at all -- every member vs the oracle over a cold and a warm-started period."""
    from tests.parity_util import small_kernel_cases
    worst, kinds = small_kernel_cases(B=9)
    assert worst <= TOL, worst
    assert kinds == [mpcqp.api.KERNEL_SMALL] * 4


def test_small_problem_kernel_with_output_bounds_on_gpu(hiplib, monkeypatch):
    """Output-bound rows on the small-problem kernel (k_step_small_y: soft band with an active ϵ, hard horizon-long bound
    with +-Inf holes, soft y + soft u, ymin with move blocking, soft and hard terminal rows): every member vs the oracle, rows on
    their bounds.  (MPCQP_SMALL_Y=1: at this batch size the handle's own specialisation would take the steps, see below.)"""
    from tests.parity_util import small_kernel_cases
    monkeypatch.setenv("MPCQP_SMALL_Y", "1")
    worst, kinds, yact = small_kernel_cases(B=9, with_y=True)
    assert worst <= TOL, worst
    assert kinds == [mpcqp.api.KERNEL_SMALL] * 6
    assert all(n > 0 for n, _ in yact) and max(e for _, e in yact) > 1e-3, yact


def test_small_batches_with_output_bounds_take_their_specialisation(hiplib, monkeypatch):
    """Round 5: with dense rows and at most 1024 controllers (one per SIMD as one-controller-per-wavefront grid) the handle's
    own specialisation is 1.5 times faster than the four-per-wavefront kernel's dense-row variant (scripts/small_vs_wave.py):
    the same cases as above run on it, with the same answers."""
    from tests.parity_util import small_kernel_cases
    monkeypatch.delenv("MPCQP_SMALL_Y", raising=False)
    worst, kinds, yact = small_kernel_cases(B=9, with_y=True)
    assert worst <= TOL, worst
    assert kinds == [mpcqp.api.KERNEL_ONDEMAND] * 6, kinds


def test_small_problem_kernel_with_soft_ymax_full_batch(hiplib):
    """65536 C2-size controllers with a soft output band (the setconstraint!(ymax = ...) case that used to leave the
    small-problem kernel): all solved, and the same optimum as the one-controller-per-wavefront kernels
    (MPCQP_SMALL_Y=0 in a process of its own) on a strided sample."""
    import subprocess, sys, json
    code = ("import sys, json; sys.path.insert(0, '.')\n"
            "import numpy as np, mpcqp\nfrom mpcqp import synth\n"
            "cfg = synth.Config('C2y', nx=4, nu=2, ny=2, Hp=20, Hc=5, Cwt=1e5)\n"
            "B = 65536; bt = synth.make_batch(cfg, B, seed=0)\n"
            "mpc = mpcqp.BatchLinMPC(bt['Ahat'], bt['Bhu'], bt['Chat'], Hp=20, Hc=5, Cwt=1e5, Mwt=np.ones(2), Nwt=np.full(2, 0.1), Lwt=np.zeros(2))\n"
            "mpc.setconstraint(umin=[-1, -1], umax=[1, 1], Δumin=[-0.5, -0.5], Δumax=[0.5, 0.5], ymin=[-0.15, -0.2], ymax=[0.15, 0.2])\n"
            "mpc.lastu0 = bt['lastu0'].copy(); mpc.moveinput(bt['xhat0'], bt['ry'])\n"
            "print('RESULT', json.dumps(dict(kind=int(mpc.hd.kernel_kind()), ok=float(np.mean(mpc.status == 0)), ms=mpc.hd.last_step_ms(),"
            " Z=mpc.Z[::257].tolist(), eps=float(mpc.Z[:, -1].max()))))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("small", {}), ("wave", {"MPCQP_SMALL_Y": "0"})):
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        res[tag] = json.loads(out.stdout.split("RESULT")[1])
    assert res["small"]["kind"] == mpcqp.api.KERNEL_SMALL and res["wave"]["kind"] != mpcqp.api.KERNEL_SMALL
    assert res["small"]["ok"] == 1.0 and res["small"]["eps"] > 1e-3
    Zs, Zw = np.array(res["small"]["Z"]), np.array(res["wave"]["Z"])
    err = np.abs(Zs - Zw)[:, :-1].max(axis=1) / np.maximum(1.0, np.abs(Zw[:, :-1]).max(axis=1))
    assert err.max() <= TOL, err.max()
    print(f"soft-ymax C2 shapes, B = 65536: small-problem kernel {res['small']['ms']:.2f} ms, one controller per wavefront {res['wave']['ms']:.2f} ms")


def test_small_problem_kernel_full_batch_C2(hiplib):
    """65536 C2 controllers on the small-problem kernel: all optimal (the wide-neighbourhood safeguard of the step
    length matters: instance 10539 cycles without it), bounds respected, a strided sample of 1024 instances against the oracle."""
    cfg = synth.C2
    B = 65536
    bt = synth.make_batch(cfg, B, seed=0)
    mpc = make_controller(cfg, bt)
    assert mpc.hd.kernel_kind() == mpcqp.api.KERNEL_SMALL
    mpc.lastu0 = bt["lastu0"].copy()
    u = mpc.moveinput(bt["xhat0"], bt["ry"])
    assert np.all(mpc.status == 0) and mpc.iters.max() <= 40
    nDU = cfg.nu * cfg.Hc
    DU = mpc.Z[:, :nDU]
    assert DU.max() <= cfg.dumax + 1e-9 and DU.min() >= cfg.dumin - 1e-9
    U = np.cumsum(DU.reshape(B, cfg.Hc, cfg.nu), axis=1) + bt["lastu0"][:, None, :]
    assert U.max() <= cfg.umax + 1e-9 and U.min() >= cfg.umin - 1e-9
    idx = np.arange(0, B, 64)                      # a strided sample of 1024 instances over the whole batch
    sub = {k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in bt.items()}
    ref = oracle_batch(cfg, sub)
    assert rel_err(mpc.Z[idx], ref["Z"], nDU).max() <= TOL
