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


def run(args, rank, world, local, dist):
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
        return
    mean_it = it_sum / Bglobal
    # (the reported iteration count of an estimator is its last iteration index: factorisations = it + 1)
    flops, w_iter = mhe_flops(cfg, mean_it + 1.0)
    kms = float(np.mean(sh.kern_ms))
    achieved = flops * B / (kms * 1e-3) / 1e12
    Zt = sh.h.get(pm.GET_ZTILDE)
    traffic = None      # HBM bytes per launch from the PMC passes committed under profiles/
    try:
        tr = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic_k_mhe_step.json")))
        if tr["config"] == args.config and tr["batch"] == B:
            traffic = tr["hbm_bytes_per_launch"]
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
                     "flops_per_solve": flops,
                     "note": "FP64 vector peak (v_fma_f64; the kernel's blocks are 12 x 12, below the 16 x 16 x 4 MFMA "
                             "tile, and run on v_fma_f64 + DPP row broadcasts); flops = setup + I W_iter, block "
                             "tridiagonal count, I = mean factorisations per solve; the window state of a wavefront (~190 KB) "
                             "streams through HBM: traffic / kernel_ms is the measured HBM rate (peak 8000 GB/s), the "
                             "limiter of this kernel next to its dependent sweeps"},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args)
    print(json.dumps(out), flush=True)


def cpu_baseline(cfg, args):
    """oracle/mhe.py (dense NumPy restatement of the reference's condensed QP + exact active-set solve) on one
    host core: a bounded number of steady-state periods of the first estimators of the same workload."""
    from mpcqp import synth
    from oracle import estim as es
    from oracle import mhe as om
    n = 2
    bt = synth.make_mhe_batch(cfg, n, seed=args.seed)
    nper = cfg.He + 64
    Y, U, D = synth.make_mhe_data(cfg, bt, nper, seed=args.seed)
    ests = []
    for b in range(n):
        model = es.LinModelOracle(bt["A"][b], bt["Bu"][b], bt["C"][b], bt["Bd"][b] if cfg.nd else None,
                                  np.zeros((cfg.nym, cfg.nd)) if cfg.nd else None)
        e = om.MHEOracle(model, He=cfg.He, direct=cfg.direct, sigmaQ=np.full(cfg.nx, cfg.sigmaQ),
                         sigmaR=np.full(cfg.nym, cfg.sigmaR), sigmaQint_ym=np.full(cfg.nym, cfg.sigmaQint),
                         sigmaP_0=np.full(cfg.nx, cfg.sigmaP0), sigmaPint_ym_0=np.full(cfg.nym, cfg.sigmaP0),
                         nint_ym=[1] * cfg.nym)
        if np.isfinite(cfg.xabs):
            e.setconstraint(xhatmin=np.full(cfg.nxh, -cfg.xabs), xhatmax=np.full(cfg.nxh, cfg.xabs))
        ests.append(e)
    done, t_solve, k = 0, 0.0, 0
    while k < nper and (k < cfg.He + 2 or t_solve < args.cpu_seconds):
        for b, e in enumerate(ests):
            d = D[k][b] if cfg.nd else ()
            t0 = time.perf_counter()
            e.preparestate(Y[k][b], d)
            e.updatestate(U[k][b], Y[k][b], d)
            if k >= cfg.He:                       # steady state only
                t_solve += time.perf_counter() - t0
                done += 1
        k += 1
    return {"value": done / t_solve if t_solve > 0 else None, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"{done} steady-state periods (full window) of the first {n} estimators, {t_solve:.1f} s; "
                      "oracle/mhe.py: the reference's condensed QP (nZ̃ = 252) built densely in NumPy and solved by "
                      "oracle/qp.py"}
