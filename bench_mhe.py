"""bench.py --config C5: estimator periods (= QP solves) per second of the batched linear
MovingHorizonEstimator (BASELINE.json configs[4]: He = 20, nx̂ = 12, batch 65536 on one MI355X).

A "step" = one estimator period of every estimator of the resident batch in steady state (full, moving
window): preparestate! (add the new measurement to the windows, correct the arrival covariance, solve the
QP) and updatestate! (advance the arrival covariance, keep u) -- kernels k_mhe_cov, k_mhe_step, k_mhe_cov.
Every timed step consumes a NEW measurement / input of a pre-generated plant record that already lies in HBM;
the windows are filled (He periods) before the timed region.  N > 1: one shard of 65536 estimators per
rank, no collective (weak scaling; BASELINE quotes config 5 on one GPU)."""
from __future__ import annotations

import json
import os
import time

import numpy as np

FP64_PEAK_TFLOPS = 78.6


def mhe_flops(cfg, mean_iters):
    """Structure-exploiting FP64 count of one solve (FMA = 2 flop), see DESIGN.md 'f2': per interior-point
    iteration and stage, the block Thomas recursion on nx̂ x nx̂ blocks: two products for S = Φ - O Si O'
    (4 n^3), its inverse (2 n^3), and two solves of two sweeps with two block mat-vecs each (16 n^2)."""
    n = cfg.nxh
    stages = cfg.He + 1
    w_iter = stages * (6 * n ** 3 + 16 * n * n)
    w_setup = stages * 8 * n * n
    return w_setup + mean_iters * w_iter, w_iter


class MheShard:
    def __init__(self, cfg, lo, B, seed, local, periods):
        import torch
        from mpcqp import mhe as pm
        from mpcqp import synth
        self.cfg, self.B = cfg, B
        bt = synth.make_mhe_batch(cfg, B, seed=seed, lo=lo)
        self.bt = bt
        Y, U, D = synth.make_mhe_data(cfg, bt, periods, seed=seed, lo=lo)
        h = pm.MheHandle(B, cfg.nxh, cfg.nu, cfg.nym, cfg.nd, cfg.He, direct=cfg.direct, device=local)
        nd = cfg.nd
        h.set_model(bt["Ahat"], bt["Bhu"], bt["Chm"], bt["Bhd"] if nd else None, bt["Dhdm"] if nd else None, None,
                    bt["Qhat"], bt["Rhat"])
        full = lambda v, n: None if not np.isfinite(v) else np.full((B, n), float(v))
        neg = lambda a: None if a is None else -a
        h.set_bounds(neg(full(cfg.xabs, cfg.nxh)), full(cfg.xabs, cfg.nxh), neg(full(cfg.wabs, cfg.nxh)),
                     full(cfg.wabs, cfg.nxh), neg(full(cfg.vabs, cfg.nym)), full(cfg.vabs, cfg.nym))
        if np.isfinite(cfg.Cwt):                 # bounds relaxed by the slack (soft kernel variant)
            one = np.ones((B, cfg.nxh))
            h.set_softness(np.full(B, cfg.Cwt), c_xmin=one if np.isfinite(cfg.xabs) else None, c_xmax=one if np.isfinite(cfg.xabs) else None)
        h.init(None, bt["P0"])
        self.h = h
        dev = torch.device("cuda", local)
        self.dev = dev
        self.Y, self.U = torch.from_numpy(Y).to(dev), torch.from_numpy(U).to(dev)
        self.D = torch.from_numpy(D).to(dev) if nd else None
        self.k = 0
        self.periods = periods
        self.stream = None
        self.kern_ms = []

    def step(self, record=False):
        k = self.k % self.periods
        dk = self.D[k].data_ptr() if self.D is not None else 0
        self.h.prepare_device(self.Y[k].data_ptr(), dk)
        self.h.update_device(self.U[k].data_ptr(), self.Y[k].data_ptr(), dk)
        self.k += 1
        if record:
            self.h.sync()
            self.kern_ms.append(self.h.last_ms())


def measure(args, rank, world, local, dist, cpu=True):
    """One benchmark of the estimator period; returns the JSON-able record on rank 0 (None elsewhere)."""
    import torch
    from mpcqp import mhe as pm
    from mpcqp import synth
    cfg = synth.get_mhe_config(args.config)
    B = args.batch or 65536
    Bglobal = B * world
    fill = cfg.He
    periods = fill + args.warmup + args.steps
    sh = MheShard(cfg, rank * B, B, args.seed, local, periods)

    def barrier():
        if dist is not None:
            dist.barrier()
        sh.h.sync()
        torch.cuda.synchronize()

    for _ in range(fill + args.warmup):          # fill the windows, then warm up in steady state
        sh.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sh.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status, iters = sh.h.get(pm.GET_STATUS), sh.h.get(pm.GET_ITERS)
    # kernel time of the solve from the handle's HIP events (recorded on the handle's stream around
    # k_mhe_step), measured in extra periods after the timed region so that the timed loop stays asynchronous
    sh.periods += 0
    sh.k = fill
    for _ in range(min(args.steps, periods - fill)):
        sh.step(record=True)
    n_opt, it_sum = int((status == 0).sum()), float(iters.sum())
    if dist is not None:
        agg = torch.tensor([n_opt, it_sum], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(agg)
        n_opt, it_sum = int(agg[0].item()), float(agg[1].item())
    if rank != 0:
        return None
    mean_it = it_sum / Bglobal
    # (the reported iteration count of an estimator is its last iteration index: factorisations = it + 1)
    flops, w_iter = mhe_flops(cfg, mean_it + 1.0)
    kms = float(np.mean(sh.kern_ms))
    achieved = flops * B / (kms * 1e-3) / 1e12
    Zt = sh.h.get(pm.GET_ZTILDE)
    # HBM bytes per launch from the PMC passes committed under profiles/ (traffic.json: scripts/profile_all.sh +
    # pmc_summary_all.py, the steady-state launch of each variant) and the algorithmic bytes of one period next to them
    traffic, algo = None, None
    try:
        tr = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")))
        k = tr["k_mhe_step_12_soft" if args.config == "C5S" else "k_mhe_step_12_hard"]
        if k["units_per_launch"] == B and cfg.nxh == 12:
            traffic, algo = k["hbm_bytes_per_launch"], k.get("algorithmic_bytes_per_unit")
    except Exception:
        pass
    out = {
        "metric": "QP solves/sec (MovingHorizonEstimator period)",
        "value": Bglobal * args.steps / elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg.name, "batch_per_gpu": B, "global_batch": Bglobal, "nxhat": cfg.nxh, "nu": cfg.nu,
                   "nym": cfg.nym, "nd": cfg.nd, "He": cfg.He, "form": "current" if cfg.direct else "predictor",
                   "window": "full and moving (filled before the timed region)",
                   "decision_variables_reference": cfg.nxh * (cfg.He + 1),
                   "rows": int(2 * cfg.nxh * (cfg.He + 1) * np.isfinite(cfg.xabs) + 2 * cfg.nxh * cfg.He * np.isfinite(cfg.wabs)
                               + 2 * cfg.nym * cfg.He * np.isfinite(cfg.vabs)),
                   "ipm_mean_iters": mean_it + 1.0, "optimal_fraction": n_opt / Bglobal,
                   "estimators_on_a_state_bound": float(np.mean(np.abs(Zt[:, :cfg.nxh]).max(axis=1) >= cfg.xabs - 1e-6))
                   if np.isfinite(cfg.xabs) else None,
                   "register_columns": sh.h.register_columns(),
                   "kernels_per_period": "k_mhe_cov (correct) + k_mhe_step + k_mhe_cov (predict)"},
        "roofline": {"bound": "mfma", "kernel": "k_mhe_step", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "kernel_ms": kms,
                     "hbm_measured_GBps": (traffic / (kms * 1e-3) / 1e9) if traffic else None,
                     "algorithmic_bytes_per_period": algo,
                     "traffic_over_algorithmic": (traffic / B / algo) if (traffic and algo) else None,
                     "flops_per_solve": flops,
                     "note": "FP64 vector peak (v_fma_f64; the kernel's blocks are 12 x 12, below the 16 x 16 x 4 MFMA "
                             "tile, and run on v_fma_f64 + DPP row broadcasts); flops = setup + I W_iter, block "
                             "tridiagonal count, I = mean factorisations per solve; the window state of a wavefront (~190 KB) "
                             "streams through HBM: traffic / kernel_ms is the measured HBM rate (peak 8000 GB/s), the "
                             "limiter of this kernel next to its dependent sweeps"},
    }
    if world == 1 and cpu and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args)
    return out


def run(args, rank, world, local, dist):
    out = measure(args, rank, world, local, dist)
    if out is not None:
        print(json.dumps(out), flush=True)


def _cpu_worker(job):
    """Steady-state periods of ONE estimator of the workload on one core (oracle/mhe.py); returns (periods, seconds)."""
    cfg, seed, b, budget = job
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    from mpcqp import synth
    from oracle import estim as es
    from oracle import mhe as om
    bt = synth.make_mhe_batch(cfg, 1, seed=seed, lo=b)
    nper = cfg.He + 64
    Y, U, D = synth.make_mhe_data(cfg, bt, nper, seed=seed, lo=b)
    model = es.LinModelOracle(bt["A"][0], bt["Bu"][0], bt["C"][0], bt["Bd"][0] if cfg.nd else None,
                              np.zeros((cfg.nym, cfg.nd)) if cfg.nd else None)
    e = om.MHEOracle(model, He=cfg.He, direct=cfg.direct, sigmaQ=np.full(cfg.nx, cfg.sigmaQ),
                     sigmaR=np.full(cfg.nym, cfg.sigmaR), sigmaQint_ym=np.full(cfg.nym, cfg.sigmaQint),
                     sigmaP_0=np.full(cfg.nx, cfg.sigmaP0), sigmaPint_ym_0=np.full(cfg.nym, cfg.sigmaP0),
                     nint_ym=[1] * cfg.nym)
    if np.isfinite(cfg.xabs):
        e.setconstraint(xhatmin=np.full(cfg.nxh, -cfg.xabs), xhatmax=np.full(cfg.nxh, cfg.xabs))
    done, t_solve, k = 0, 0.0, 0
    while k < nper and (k < cfg.He + 2 or t_solve < budget):
        d = D[k][0] if cfg.nd else ()
        t0 = time.perf_counter()
        e.preparestate(Y[k][0], d)
        e.updatestate(U[k][0], Y[k][0], d)
        if k >= cfg.He:                       # steady state only
            t_solve += time.perf_counter() - t0
            done += 1
        k += 1
    return done, t_solve


def cpu_baseline(cfg, args):
    """The oracle's C port of the estimator period (oracle/mhe_ref.c: the same QP in the state-sequence, block-tridiagonal
    form the GPU kernel solves, Mehrotra interior point, OpenMP over estimators) on this box's host cores: a bounded
    sample of the same workload.  Steady-state periods only: the time of a run over the window fill alone is subtracted
    from a run over fill + K periods.  (Round 3 timed the dense NumPy oracle here, 166 solves/s on 256 cores: the
    reference's condensed 252-variable QP -- kept as `numpy_dense_oracle` for the record when MPCQP_BENCH_NUMPY_MHE=1.)"""
    from mpcqp import synth
    from oracle import mhe_cport
    if cfg.nd or not cfg.direct or np.isfinite(cfg.wabs) or np.isfinite(cfg.vabs) or np.isfinite(cfg.Cwt):
        return {"value": None, "unit": "solves/s", "cores": 0, "kind": "port", "sample": "the C port covers x̂ bounds, current form, nd = 0"}
    threads = mhe_cport.threads()
    n = 64 * max(1, threads // 4)
    K = 8
    bt = synth.make_mhe_batch(cfg, n, seed=args.seed)
    Y, U, _ = synth.make_mhe_data(cfg, bt, cfg.He + K, seed=args.seed)
    t0 = time.perf_counter()
    mhe_cport.run(bt, Y[:cfg.He], U[:cfg.He], cfg.He, cfg.xabs)
    t_fill = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, it, st = mhe_cport.run(bt, Y, U, cfg.He, cfg.xabs)
    t_all = time.perf_counter() - t0
    rate = n * K / max(t_all - t_fill, 1e-9)
    # scale the sample to ~cpu_seconds of work
    K2 = int(max(K, min(400, rate * args.cpu_seconds / n)))
    if K2 > K:
        Y, U, _ = synth.make_mhe_data(cfg, bt, cfg.He + K2, seed=args.seed)
        t0 = time.perf_counter()
        _, it, st = mhe_cport.run(bt, Y, U, cfg.He, cfg.xabs)
        t_all = time.perf_counter() - t0
        K = K2
        rate = n * K / max(t_all - t_fill, 1e-9)
    out = {"value": rate, "unit": "solves/s", "cores": int(threads), "kind": "port",
           "sample": f"{n} estimators of the same workload x {K} steady-state periods (full window; the window fill timed separately and "
                     f"subtracted), {t_all - t_fill:.1f} s; oracle/mhe_ref.c: block-tridiagonal state-sequence QP, Mehrotra interior point, "
                     f"OpenMP over estimators, mean {float(it[cfg.He:].mean()):.1f} iterations, all solved: {bool((st == 0).all())}"}
    if os.environ.get("MPCQP_BENCH_NUMPY_MHE") == "1":
        out["numpy_dense_oracle"] = cpu_baseline_numpy(cfg, args)
    return out


def cpu_baseline_numpy(cfg, args):
    """oracle/mhe.py (dense NumPy restatement of the reference's condensed QP + exact active-set solve) on every host
    core: one estimator of the same workload per core (single-threaded BLAS), a bounded number of steady-state periods
    each; value = sum over the cores of periods / seconds.  Runs in a fresh interpreter (no GPU runtime in the process
    that forks the workers)."""
    import subprocess
    import sys
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline", str(args.config), str(float(args.cpu_seconds)), str(int(args.seed))]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.abspath(__file__)))
    if r.returncode != 0:
        return {"value": None, "unit": "solves/s", "cores": 0, "kind": "port", "sample": "failed: " + r.stderr[-300:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def _cpu_baseline_main(name, budget, seed):
    import multiprocessing as mp
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from mpcqp import synth
    cfg = synth.get_mhe_config(name)
    cores = max(1, len(os.sched_getaffinity(0)))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(cfg, seed, b, budget) for b in range(cores)])
    wall = time.perf_counter() - t0
    rate = sum(d / t for d, t in res if t > 0)
    done = sum(d for d, _ in res)
    print(json.dumps({"value": rate, "unit": "solves/s", "cores": cores, "kind": "port",
                      "sample": f"{done} steady-state periods (full window) in all, one estimator of the workload per core, "
                                f"{budget:.0f} s of solves per core ({wall:.0f} s wall with the window fill); oracle/mhe.py: the "
                                "reference's condensed QP (nZ̃ = 252) built densely in NumPy and solved by oracle/qp.py, "
                                "single-threaded BLAS per process"}))


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 5 and sys.argv[1] == "--cpu-baseline":
        _cpu_baseline_main(sys.argv[2], float(sys.argv[3]), int(sys.argv[4]))
