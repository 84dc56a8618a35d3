#!/usr/bin/env python
"""bench.py -- QP solves/sec of the batched LinMPC `moveinput!` step on N MI355X (one process per
GPU; `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` for N > 1).

A "step" = one pass of the hot path (initpred! + linconstraint! + QP solve + getinput!, kernel K3)
over one resident batch of synthetic controllers, every step a cold-started solve of the same seeded
instances (all K timed steps do identical work), inputs in HBM before the timed region.

Workload (BASELINE.json):
  N = 1   configs[2]: batch 65536, nx=12 nu=4 ny=4, Hp=30 Hc=10, hard umin/umax, soft ymax ("C3").
  N > 1   configs[3]: the same shapes, GLOBAL batch 262144 split into contiguous index ranges over the
          ranks (32768 per GPU at N = 8; `sharding.shard_range`, the rule of mpcqp_multi_create) with no
          collective on the data path -- `"scaling": "strong"`; the weak-scaling figure (65536
          controllers per GPU, global 65536 N) is measured in the same run and reported under
          `config.weak_scaling`.  `--mode weak` makes the weak run the headline instead.
After the timed region the ranks gather status and first moves of the whole batch with ONE all_gather
each over RCCL, and one period's inputs born on rank 0 are scattered to the ranks with ONE scatter each (the
scatter/gather of the north star; both are outside the step and timed separately: config.scatter / config.gather).

`--config C5` runs BASELINE configs[4] instead (linear MovingHorizonEstimator, one estimator period per
step): see bench_mhe.py.

Prints ONE JSON line (rank 0) with `roofline` (FP64 flops of the dominant kernel k_step against the
chip's FP64 peak, kernel time from HIP events on the launch stream) and `cpu_baseline` (the oracle's C
port on the host cores, bounded sample, rank 0 at N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz = half the 157.3 TF FP32 vector peak
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md, "HBM3E peak BW"


def algorithmic_flops(cfg, mean_iters, rows_u, rows_y):
    """SURVEY.md 8(d) structure-exploiting count (FMA = 2 flop), per solve.  See DESIGN.md."""
    n = cfg.nu * cfg.Hc + (0 if np.isinf(cfg.Cwt) else 1)
    p = cfg.ny * cfg.Hp
    nxh = cfg.nx + cfg.ny
    w_grad = 2 * p * (nxh + cfg.nu) + 2 * p * n
    w_iter = ((p * n * n if rows_y else 0) + n * n + n ** 3 / 3.0 + 4 * n * n
              + 8 * (rows_y * n + rows_u * 1))
    return w_grad + mean_iters * w_iter, w_grad, w_iter


def algorithmic_bytes(cfg):
    """Compulsory HBM bytes per solve in the on-device-condensation accounting of SURVEY 8(d)."""
    nxh = cfg.nx + cfg.ny
    n = cfg.nu * cfg.Hc + (0 if np.isinf(cfg.Cwt) else 1)
    # (model, x̂0, lastu0 and ry, weights M / N / L / C, bounds u / y both sides, warm start; C3: 478 doubles in)
    ins = (nxh * nxh + nxh * cfg.nu + cfg.ny * nxh + nxh + cfg.nu + cfg.ny + (cfg.ny + 2 * cfg.nu + 1) + (2 * cfg.nu + 2 * cfg.ny) + n)
    outs = n + cfg.nu + 1
    return 8 * (ins + outs)


class Shard:
    """This rank's contiguous shard [lo, lo + B) of a seeded global batch, resident on its GPU."""

    def __init__(self, cfg, lo, B, seed, local, multiple_shooting=False):
        import torch
        import mpcqp
        from mpcqp import synth
        self.cfg, self.B, self.lo = cfg, B, lo
        bt = synth.make_batch(cfg, B, seed=seed, lo=lo)
        self.bt = bt
        neps = 0 if np.isinf(cfg.Cwt) else 1
        hd = mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, neps=neps, device=local,
                          flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START)
        hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
        hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt),
                       np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt) if neps else None)
        full = lambda v, n: None if not np.isfinite(v) else np.full((B, n), float(v))
        hd.set_bounds(U0min=full(cfg.umin, hd.nU), U0max=full(cfg.umax, hd.nU),
                      DUmin=full(cfg.dumin, hd.nDU), DUmax=full(cfg.dumax, hd.nDU),
                      Y0min=full(cfg.ymin, hd.nY), Y0max=full(cfg.ymax, hd.nY))
        if multiple_shooting:                   # the stage-structured kernel (csrc/ms_bodies.h): nothing to prepare
            hd.set_transcription(mpcqp.api.MULTIPLE_SHOOTING)
            assert hd.transcription_supported() == 0
        self.kernel = mpcqp.api.KERNEL_MS if multiple_shooting else hd.prepare()   # specialised kernel (compiled once per shape, never in a step)
        self.hd = hd
        dev = torch.device("cuda", local)
        self.dev = dev
        self.t_x = torch.from_numpy(bt["xhat0"]).to(dev)
        self.t_lu = torch.from_numpy(bt["lastu0"]).to(dev)
        self.t_ry = torch.from_numpy(bt["ry"]).to(dev)
        self.t_Z = torch.zeros((B, hd.nZ), dtype=torch.float64, device=dev)
        self.t_u0 = torch.empty((B, cfg.nu), dtype=torch.float64, device=dev)
        self.t_st = torch.empty(B, dtype=torch.int32, device=dev)
        self.t_it = torch.empty(B, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.current_stream()

    def step(self):
        self.hd.step_device(self.t_x.data_ptr(), self.t_lu.data_ptr(), self.t_ry.data_ptr(), self.t_Z.data_ptr(),
                            self.t_u0.data_ptr(), self.t_st.data_ptr(), iters=self.t_it.data_ptr(),
                            stream=self.stream.cuda_stream)


COLL_DEVICE = [None]      # device of the collectives when it is not the shard's GPU (MPCQP_BENCH_ONE_GPU)


def timed_run(sh, steps, warmup, dist):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns (wall seconds, max over the
    ranks; per-step kernel ms from events on the launch stream)."""
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        sh.step()
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    for k in range(steps):
        evs[k].record(sh.stream)
        sh.step()
    evs[steps].record(sh.stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(steps)]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=COLL_DEVICE[0] or sh.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, kern_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="controllers per GPU (0: the BASELINE workload)")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--mode", default="auto", choices=["auto", "config4", "weak"],
                    help="N > 1: config4 = global batch 262144 split over the ranks (default), weak = 65536 per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 / C5 records of config.secondary")
    args = ap.parse_args()

    import torch
    import mpcqp
    from mpcqp import sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MPCQP_BENCH_ONE_GPU=1 (tests only): every rank uses device 0 and the collectives run over gloo on the host -- the
    # N > 1 code path (sharding, strong + weak runs, gather, reductions) exercised on a box with a single GPU
    one_gpu = os.environ.get("MPCQP_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    args.coll_device = torch.device("cpu") if one_gpu else torch.device("cuda", local)
    COLL_DEVICE[0] = torch.device("cpu") if one_gpu else None

    if args.config in synth.MHE_CONFIGS or args.config.startswith("mhe:"):
        # the linear MovingHorizonEstimator (BASELINE configs[4]) -- bench_mhe.py
        import bench_mhe
        args.config = args.config[4:] if args.config.startswith("mhe:") else args.config
        bench_mhe.run(args, rank, world, local, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    cfg = synth.get_config(args.config)
    PER_GPU = 65536
    GLOBAL4 = 262144
    strong = world > 1 and args.mode in ("auto", "config4") and not args.batch
    if args.batch:
        lo, B, Bglobal = rank * args.batch, args.batch, args.batch * world
    elif strong:
        lo, B = sharding.shard_range(GLOBAL4, rank, world)
        Bglobal = GLOBAL4
    else:
        lo, B, Bglobal = rank * PER_GPU, PER_GPU, PER_GPU * world
    sh = Shard(cfg, lo, B, args.seed, local)
    hd = sh.hd
    elapsed, kern_ms = timed_run(sh, args.steps, args.warmup, dist)

    status = sh.t_st.cpu().numpy()
    iters = sh.t_it.cpu().numpy()
    n_opt, it_sum = int((status == 0).sum()), float(iters.sum())
    if dist is not None:
        agg = torch.tensor([n_opt, it_sum], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(agg)
        n_opt, it_sum = int(agg[0].item()), float(agg[1].item())
    mean_it = it_sum / Bglobal

    # the gather of the north star (outside the step): status and first moves of the whole batch on every rank
    gather = None
    if dist is not None:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st_all = sharding.gather(sh.t_st, Bglobal, dist, device=args.coll_device)
        u_all = sharding.gather(sh.t_u0, Bglobal, dist, device=args.coll_device)
        torch.cuda.synchronize()
        gather = {"ms": (time.perf_counter() - t0) * 1e3, "bytes_per_rank": int(B * (4 + 8 * cfg.nu)),
                  "optimal_fraction": float((st_all == 0).double().mean().item()),
                  "u0_checksum": float(u_all.sum().item()), "collective": "all_gather (RCCL), one per array"}

    # ... and the scatter: one period's inputs (x̂0, lastu0, ry of the whole batch) born on rank 0 reach the ranks as their
    # slices, ONE scatter collective per array; checked against the shard every rank generated itself
    scatter = None
    if dist is not None:
        from mpcqp import synth as _synth
        whole = _synth.make_batch(cfg, Bglobal, seed=args.seed) if rank == 0 else None
        dev_c = args.coll_device
        src = {k: (torch.from_numpy(whole[k]).to(dev_c) if rank == 0 else None) for k in ("xhat0", "lastu0", "ry")}
        like = {"xhat0": sh.t_x, "lastu0": sh.t_lu, "ry": sh.t_ry}
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        got = {k: sharding.scatter(src[k], Bglobal, dist, src=0, device=dev_c, like=like[k]) for k in src}
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        same = all(torch.equal(got[k].to(like[k].device), like[k]) for k in got)
        scatter = {"ms": ms, "bytes_per_rank": int(B * 8 * (cfg.nxh + cfg.nu + cfg.ny)), "matches_local_shard": bool(same),
                   "collective": "scatter (RCCL), one per array, source rank 0"}
        del whole, src, got

    # weak-scaling figure next to the config-4 run
    weak = None
    if strong:
        shw = Shard(cfg, rank * PER_GPU, PER_GPU, args.seed, local)
        ew, _ = timed_run(shw, args.steps, args.warmup, dist)
        weak = {"batch_per_gpu": PER_GPU, "global_batch": PER_GPU * world, "value": PER_GPU * world * args.steps / ew,
                "ms_per_step": ew / args.steps * 1e3}
        del shw

    # setmodel! path (K1 + K2 re-condensation of all models), reported next to the step time
    hd.recondense_device(stream=sh.stream.cuda_stream)
    torch.cuda.synchronize()
    recond_ms, k1_ms = hd.last_condense_ms(), hd.last_predmat_ms()

    # resident closed loop: Kalman correction + moveinput! + Kalman prediction of one period in ONE launch
    # (mpcqp_loop_device); x̂0, u stay in HBM, the measurement is synthetic noise, the gain a synthetic
    # (B, nx̂, ny) array -- the arithmetic of a period does not depend on their values
    loop = None
    if world == 1:
        rg = np.random.default_rng(1)
        hd.kf_set(np.ascontiguousarray(0.05 * rg.standard_normal((B, cfg.ny, cfg.nxh))), np.arange(cfg.ny))
        t_y = torch.from_numpy(0.1 * rg.standard_normal((B, cfg.ny))).to(sh.dev)
        t_xl, t_Zl, t_stl, t_itl = sh.t_x.clone(), torch.zeros_like(sh.t_Z), sh.t_st.clone(), sh.t_it.clone()
        bufs = [sh.t_lu.clone(), torch.empty_like(sh.t_u0)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            hd.loop_device(t_xl.data_ptr(), t_y.data_ptr(), bufs[k % 2].data_ptr(), sh.t_ry.data_ptr(), t_Zl.data_ptr(),
                           bufs[(k + 1) % 2].data_ptr(), t_stl.data_ptr(), iters=t_itl.data_ptr(), stream=sh.stream.cuda_stream)
        torch.cuda.synchronize()
        dtl = time.perf_counter() - t0
        loop = {"periods_per_s": B * args.steps / dtl, "ms_per_period": dtl / args.steps * 1e3, "launches_per_period": 1,
                "optimal_fraction": float((t_stl == 0).double().mean().item()),
                "what": "preparestate! (SteadyKalmanFilter) + moveinput! + updatestate! fused in the step kernel"}

    # end to end through host pointers (PCIe both ways, pageable NumPy arrays): never the reported value
    e2e_ms = None
    if world == 1:
        Zh = np.zeros((B, hd.nZ))
        hd.step(sh.bt["xhat0"], sh.bt["lastu0"], sh.bt["ry"], Zh)
        t0 = time.perf_counter()
        u0h, sth, ith = hd.step(sh.bt["xhat0"], sh.bt["lastu0"], sh.bt["ry"], Zh)
        e2e_ms = (time.perf_counter() - t0) * 1e3

    if rank == 0:
        total = Bglobal * args.steps
        value = total / elapsed
        # row counts of the reference's A Z̃ <= b (i_b rule); the kernel merges the U rows of a
        # move-blocking interval, the flop model keeps the reference's count
        rows_u = (int(np.isfinite(cfg.umin)) + int(np.isfinite(cfg.umax))) * hd.nU
        rows_y = (int(np.isfinite(cfg.ymin)) + int(np.isfinite(cfg.ymax))) * hd.nY
        neps = 0 if np.isinf(cfg.Cwt) else 1
        flops, w_grad, w_iter = algorithmic_flops(cfg, mean_it, rows_u, rows_y)
        kms, kmed = float(np.mean(kern_ms)), float(np.median(kern_ms))
        achieved = flops * B / (kms * 1e-3) / 1e12
        # rows active at the optimum (rank 0's shard): inputs on a bound, outputs on the soft bound
        Zr = sh.t_Z.cpu().numpy()
        nDU = hd.nDU
        U0 = np.cumsum(Zr[:, :nDU].reshape(B, cfg.Hc, cfg.nu), axis=1) + sh.bt["lastu0"][:, None, :]
        act_u = (np.abs(U0 - cfg.umax) <= 1e-9) | (np.abs(U0 - cfg.umin) <= 1e-9) if np.isfinite(cfg.umax) else np.zeros_like(U0, bool)
        soft_on = Zr[:, -1] > 1e-9 if neps else np.zeros(B, bool)
        active = {"controllers_with_an_active_row": float((act_u.any(axis=(1, 2)) | soft_on).mean()),
                  "input_rows_on_a_bound": float(act_u.mean()), "controllers_with_slack": float(soft_on.mean())}
        # HBM bytes per launch: counters need rocprofv3 around the process (separate --pmc passes), so this is NOT a
        # measurement of this run but the figure of the committed passes over the same command (scripts/profile_all.sh);
        # traffic_source says which
        traffic, traffic_source = (committed_traffic("k_step_s_C3", B) if args.config == "C3" else (None, None))
        out = {
            "metric": "QP solves/sec (moveinput!)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg.name + (" (BASELINE configs[3]: global batch 262144 over the ranks)" if strong else ""),
                       "batch_per_gpu": B, "global_batch": Bglobal,
                       "nx": cfg.nx, "nxhat": cfg.nxh, "nu": cfg.nu, "ny": cfg.ny, "Hp": cfg.Hp, "Hc": cfg.Hc,
                       "nZ": hd.nZ, "rows": int(rows_u + rows_y + neps), "cold_start": True,
                       "ipm_mean_iters": mean_it, "optimal_fraction": n_opt / Bglobal,
                       "kernel": {0: "runtime-dimension", 1: "ahead-of-time specialisation", 2: "on-demand specialisation",
                                  3: "small-problem kernel (four controllers per wavefront)"}[sh.kernel],
                       "value_from_median_step": Bglobal / (kmed * 1e-3) if world == 1 else None,
                       "median_kernel_ms": kmed,
                       "recondense_ms": recond_ms, "recondense_K1_ms": k1_ms, "recondense_K2_ms": recond_ms - k1_ms,
                       "end_to_end_ms": e2e_ms, "active_rows": active, "closed_loop": loop,
                       "weak_scaling": weak, "scatter": scatter, "gather": gather,
                       "sharding": "contiguous index ranges (sharding.shard_range), no collective on the data path"},
            "roofline": {"bound": "mfma", "kernel": "k_step", "achieved": achieved,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_source, "kernel_ms": kms, "flops_per_solve": flops,
                         "hbm_algorithmic_GBps": algorithmic_bytes(cfg) * B / (kms * 1e-3) / 1e9,
                         "note": "FP64 peak shared by v_fma_f64 and v_mfma_f64 (one datapath: measured, "
                                 "scripts/ubench/mfma_valu_overlap.hip): half the 157.3 TF FP32 rate; flops = "
                                 "W_grad + I W_iter, SURVEY 8(d) structure-exploiting count, I = mean "
                                 "factorisations per solve (interior-point iterations + polish)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args, B)
    # the other single-GPU BASELINE configurations, after the headline's timed region: config.secondary
    if world == 1 and args.config == "C3" and not args.batch and not args.no_secondary:
        del sh, hd
        torch.cuda.empty_cache()
        out["config"]["secondary"] = secondary(args, local)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def committed_traffic(kernel, units):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/traffic.json, written by
    scripts/pmc_summary_all.py from scripts/profile_all.sh: FETCH_SIZE + WRITE_SIZE of the kernel's last launch in separate
    --pmc passes over THIS bench command) -- counters need rocprofv3 around the process, so the figure is not measured by
    this run; None when the committed passes were taken at another batch size."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        k = tr[kernel]
        if k["units_per_launch"] == units:
            return k["hbm_bytes_per_launch"], tr.get("source")
    except Exception:
        pass
    return None, None


def secondary(args, local):
    """BASELINE configs[1] (C2: nx=2 nu=2 Hp=20 Hc=5, batch 1024 -- the reference's own CPU-runnable case -- and batch
    65536), a C3-style problem with nZ~ = 106 (nu = ny = 3, Hp = 40, Hc = 35, batch 8192) and configs[4] (C5: linear MovingHorizonEstimator, He=20, batch 65536) on this GPU: value, kernel time and
    roofline fraction of each, measured like the headline (inputs resident, W warm-up + K timed steps, kernel time from
    HIP events on the launch stream)."""
    import copy
    import torch
    import bench_mhe
    from mpcqp import synth
    recs = []
    # (third record: a problem beyond one variable per lane, nZ~ = 106 -- the on-demand specialisation of
    #  spec_manifest.txt with several rows per lane; VERDICT r2 item 5 quotes this shape at B = 8192)
    # (fourth record: C2 dimensions with C3's constraint pattern -- soft ymax + hard umin / umax -- i.e. output-bound rows on
    #  the small-problem kernel, k_step_small_y)
    # (fifth record: nZ~ = 151, three rows per lane -- beyond the nZ~ = 128 the on-demand specialisations stopped at before
    #  round 5; VERDICT r4 item 5 asked for a record in this range)
    for name, B in (("C2", 1024), ("C2", 65536), ("12,3,3,40,35", 8192), ("4,2,2,20,5", 65536), ("12,3,3,50,50", 4096)):
        cfg = synth.get_config(name)
        sh = Shard(cfg, 0, B, args.seed, local)
        elapsed, kern_ms = timed_run(sh, args.steps, args.warmup, None)
        status, iters = sh.t_st.cpu().numpy(), sh.t_it.cpu().numpy()
        rows_u = (int(np.isfinite(cfg.umin)) + int(np.isfinite(cfg.umax))) * sh.hd.nU
        rows_y = (int(np.isfinite(cfg.ymin)) + int(np.isfinite(cfg.ymax))) * sh.hd.nY
        flops, _, _ = algorithmic_flops(cfg, float(iters.mean()), rows_u, rows_y)
        kms = float(np.mean(kern_ms))
        ach = flops * B / (kms * 1e-3) / 1e12
        pk = {("C2", 1024): "k_step_small_w1_12", ("C2", 65536): "k_step_small_12", ("12,3,3,40,35", 8192): "k_step_s_nZ106",
              ("4,2,2,20,5", 65536): "k_step_small_y_12", ("12,3,3,50,50", 4096): "k_step_s_nZ151"}[(name, B)]
        tr_, trs_ = committed_traffic(pk, B)
        recs.append({"workload": cfg.name, "batch": B, "metric": "QP solves/sec (moveinput!)", "value": B * args.steps / elapsed,
                     "unit": "solves/s", "ms_per_step": elapsed / args.steps * 1e3, "kernel_ms": kms,
                     "ipm_mean_iters": float(iters.mean()), "optimal_fraction": float((status == 0).mean()),
                     "kernel": {0: "runtime-dimension", 1: "ahead-of-time specialisation", 2: "on-demand specialisation",
                                3: "small-problem kernel (four controllers per wavefront)"}[sh.kernel],
                     "roofline": {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": ach / FP64_PEAK_TFLOPS, "traffic": tr_, "traffic_source": trs_, "profile": pk,
                                  "algorithmic_bytes_per_unit": algorithmic_bytes(cfg),
                                  "traffic_over_algorithmic": (tr_ / B / algorithmic_bytes(cfg)) if tr_ else None,
                                  "flops_per_solve": flops}})
        del sh
        torch.cuda.empty_cache()
    # Several SMALL handles at once: C2 at the reference's batch of 1024 occupies 256 of the 1024 SIMDs
    # for one launch latency.  mpcqp_step_device takes the caller's stream, so independent handles overlap when each is
    # stepped on a stream of its own -- no new entry point: four C2 handles (different controllers) on four streams, one
    # round = one step of each, against the same four steps issued on ONE stream.
    try:
        cfg = synth.get_config("C2")
        shs = [Shard(cfg, i * 1024, 1024, args.seed, local) for i in range(4)]
        res = {}
        for mode in ("one stream", "four streams"):
            for i, sh in enumerate(shs):
                sh.stream = torch.cuda.Stream(device=sh.dev) if (mode == "four streams") else torch.cuda.current_stream()
            for _ in range(max(1, args.warmup)):
                for sh in shs:
                    sh.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rounds = max(20, 4 * args.steps)
            for _ in range(rounds):
                for sh in shs:
                    sh.step()
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / rounds
        ok = all(bool((sh.t_st.cpu().numpy() == 0).all()) for sh in shs)
        recs.append({"workload": cfg.name + ", four handles of 1024 controllers stepped together", "batch": 4096,
                     "metric": "QP solves/sec (moveinput!)", "value": 4096 / res["four streams"], "unit": "solves/s",
                     "ms_per_round_four_streams": res["four streams"] * 1e3, "ms_per_round_one_stream": res["one stream"] * 1e3,
                     "value_one_stream": 4096 / res["one stream"], "optimal_fraction": 1.0 if ok else 0.0,
                     "kernel": "small-problem kernel, one launch per handle, one HIP stream per handle (wall time incl. launch overhead)"})
        del shs
        torch.cuda.empty_cache()
    except Exception as e:       # (a diagnostic record: never fails the bench line)
        recs.append({"workload": "C2, four handles stepped together", "error": repr(e)})
    # the MultipleShooting transcription on its stage-structured kernel (Riccati recursion inside the
    # interior-point iteration; horizon-long data in a per-wavefront HBM scratch): a long-horizon shape, Hp = Hc = 50.
    # Flops: per iteration and stage three ns x ns x ns products of the Joseph-form recursion (6 ns^3) and five vector
    # sweeps of ~8 ns^2; the condensed record of the same shape would be cheaper -- this kernel is the ROBUST path
    # (unstable plants, cond(H~) up to 3e13: profiles/r4/ms_vs_condensed_unstable_plants.txt), not the fast one.
    for name, B in (("6,2,2,50,50", 8192),):
        cfg = synth.get_config(name)
        sh = Shard(cfg, 0, B, args.seed, local, multiple_shooting=True)
        elapsed, kern_ms = timed_run(sh, max(1, args.steps // 4), 1, None)
        status, iters = sh.t_st.cpu().numpy(), sh.t_it.cpu().numpy()
        ns = cfg.nxh + cfg.nu
        flops = float(iters.mean()) * cfg.Hp * (6.0 * ns ** 3 + 40.0 * ns ** 2)
        kms = float(np.mean(kern_ms))
        ach = flops * B / (kms * 1e-3) / 1e12
        recs.append({"workload": cfg.name + ", transcription = MultipleShooting", "batch": B, "metric": "QP solves/sec (moveinput!)",
                     "value": B * max(1, args.steps // 4) / elapsed, "unit": "solves/s", "ms_per_step": elapsed / max(1, args.steps // 4) * 1e3,
                     "kernel_ms": kms, "ipm_mean_iters": float(iters.mean()), "optimal_fraction": float((status == 0).mean()),
                     "kernel": "k_ms_step_g (stage-structured MultipleShooting kernel, HBM scratch)",
                     "roofline": {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": ach / FP64_PEAK_TFLOPS, "traffic": committed_traffic("k_ms_step_g", B)[0],
                                  "traffic_source": committed_traffic("k_ms_step_g", B)[1], "profile": "k_ms_step_g",
                                  "algorithmic_bytes_per_unit": algorithmic_bytes(cfg),
                                  "traffic_over_algorithmic": (committed_traffic("k_ms_step_g", B)[0] / B / algorithmic_bytes(cfg))
                                  if committed_traffic("k_ms_step_g", B)[0] else None,
                                  "flops_per_solve": flops}})
        del sh
        torch.cuda.empty_cache()
    a5 = copy.copy(args)
    a5.config, a5.batch = "C5", 0
    r5 = bench_mhe.measure(a5, 0, 1, local, None, cpu=not args.no_cpu_baseline)
    recs.append({"workload": r5["config"]["workload"], "batch": r5["config"]["batch_per_gpu"], "metric": r5["metric"],
                 "value": r5["value"], "unit": r5["unit"], "ms_per_step": r5["ms_per_step"],
                 "kernel_ms": r5["roofline"]["kernel_ms"], "ipm_mean_iters": r5["config"]["ipm_mean_iters"],
                 "optimal_fraction": r5["config"]["optimal_fraction"], "kernel": "k_mhe_step (+ 2 k_mhe_cov per period)",
                 "roofline": {k: r5["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_period",
                                                           "traffic_over_algorithmic", "flops_per_solve")},
                 "cpu_baseline": r5.get("cpu_baseline")})
    # the soft variant of the estimator kernel (bounds relaxed by the slack: k_mhe_step<12, 15>), VERDICT r3 item 4
    a5s = copy.copy(args)
    a5s.config, a5s.batch = "C5S", 0
    r5s = bench_mhe.measure(a5s, 0, 1, local, None, cpu=False)
    recs.append({"workload": r5s["config"]["workload"], "batch": r5s["config"]["batch_per_gpu"], "metric": r5s["metric"],
                 "value": r5s["value"], "unit": r5s["unit"], "ms_per_step": r5s["ms_per_step"],
                 "kernel_ms": r5s["roofline"]["kernel_ms"], "ipm_mean_iters": r5s["config"]["ipm_mean_iters"],
                 "optimal_fraction": r5s["config"]["optimal_fraction"], "kernel": "k_mhe_step, soft variant (+ 2 k_mhe_cov per period)",
                 "roofline": {k: r5s["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_period",
                                                            "traffic_over_algorithmic", "flops_per_solve")}})
    return recs


def cpu_baseline(cfg, args, B):
    """The oracle's C port (oracle/linmpc_ref.c, dense per-controller restatement of the same
    step, OpenMP over controllers) on this box's host cores, bounded sample of the same workload."""
    from mpcqp import synth
    from oracle import cport
    threads = cport.lib().linmpc_ref_threads()
    probe = 256 * max(1, threads // 8)
    bt = synth.make_batch(cfg, probe, seed=args.seed)
    rb = cport.from_synth(cfg, bt)
    t0 = time.perf_counter()
    rb.step(bt["xhat0"], bt["lastu0"], bt["ry"])
    rate = probe / (time.perf_counter() - t0)
    n = int(min(B, max(probe, rate * args.cpu_seconds)))
    n = max(256, (n // 256) * 256)
    reps = max(1, int(round(rate * args.cpu_seconds / n)))        # ~cpu_seconds of CPU work in all
    bt = synth.make_batch(cfg, n, seed=args.seed)
    rb = cport.from_synth(cfg, bt)
    t0 = time.perf_counter()
    done = 0
    for _ in range(reps):                     # bounded by the pass estimate AND by the clock
        _, _, st, it = rb.step(bt["xhat0"], bt["lastu0"], bt["ry"])
        done += 1
        if time.perf_counter() - t0 >= args.cpu_seconds:
            break
    reps = done
    dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "solves/s", "cores": int(threads), "kind": "port",
            "sample": f"first {n} instances of the same workload x {reps} cold-start passes, "
                      f"{dt:.1f} s, dense per-controller C restatement (oracle/linmpc_ref.c), "
                      f"mean {float(it.mean()):.1f} factorisations, all optimal: {bool((st == 0).all())}"}


if __name__ == "__main__":
    main()
