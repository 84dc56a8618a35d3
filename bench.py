#!/usr/bin/env python
"""bench.py -- QP solves/sec of the batched LinMPC `moveinput!` step on N MI355X (one process per
GPU; `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` for N > 1).

A "step" = one pass of the hot path (initpred! + linconstraint! + QP solve + getinput!, kernel K3)
over one resident batch of synthetic controllers: BASELINE.json configs[2] (batch 65536 per GPU,
nx=12 nu=4 ny=4, Hp=30 Hc=10, hard umin/umax, soft ymax).  Every step is a cold-started solve of
the same seeded instances, so all K timed steps do identical work.  Inputs are in HBM before the
timed region; the batch shards by contiguous index range over the ranks with no data-path
collective (weak scaling: 65536 instances per GPU).

Prints ONE JSON line (rank 0) with `roofline` (FP64 flops of the dominant kernel k_step against the
chip's FP64 peak, kernel time from HIP events on the launch stream) and `cpu_baseline` (the
oracle's C port on the host cores, bounded sample, rank 0 at N=1 only).
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
    ins = nxh * nxh + nxh * cfg.nu + cfg.ny * nxh + nxh + 2 * cfg.nu + 2 * cfg.ny + 13 + 16 + n
    outs = n + cfg.nu + 1
    return 8 * (ins + outs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="controllers per GPU")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import mpcqp
    from mpcqp import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    cfg = synth.get_config(args.config)
    B = args.batch
    bt = synth.make_batch(cfg, B, seed=args.seed, lo=rank * B)      # this rank's shard
    neps = 0 if np.isinf(cfg.Cwt) else 1
    nxh, nu, ny, Hp, Hc = cfg.nxh, cfg.nu, cfg.ny, cfg.Hp, cfg.Hc
    hd = mpcqp.Handle(B, nxh, nu, ny, 0, Hp, Hc, neps=neps, device=local,
                      flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START)
    hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt),
                   np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt) if neps else None)
    full = lambda v, n: None if not np.isfinite(v) else np.full((B, n), float(v))
    hd.set_bounds(U0min=full(cfg.umin, hd.nU), U0max=full(cfg.umax, hd.nU),
                  DUmin=full(cfg.dumin, hd.nDU), DUmax=full(cfg.dumax, hd.nDU),
                  Y0min=full(cfg.ymin, hd.nY), Y0max=full(cfg.ymax, hd.nY))
    dev = torch.device("cuda", local)
    t_x = torch.from_numpy(bt["xhat0"]).to(dev)
    t_lu = torch.from_numpy(bt["lastu0"]).to(dev)
    t_ry = torch.from_numpy(bt["ry"]).to(dev)
    t_Z = torch.zeros((B, hd.nZ), dtype=torch.float64, device=dev)
    t_u0 = torch.empty((B, nu), dtype=torch.float64, device=dev)
    t_st = torch.empty(B, dtype=torch.int32, device=dev)
    t_it = torch.empty(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        hd.step_device(t_x.data_ptr(), t_lu.data_ptr(), t_ry.data_ptr(), t_Z.data_ptr(),
                       t_u0.data_ptr(), t_st.data_ptr(), iters=t_it.data_ptr(),
                       stream=stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k].record(stream)
        step()
    evs[args.steps].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(args.steps)]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status = t_st.cpu().numpy()
    iters = t_it.cpu().numpy()
    n_opt, mean_it = int((status == 0).sum()), float(iters.mean())
    if dist is not None:
        agg = torch.tensor([n_opt, float(iters.sum())], dtype=torch.float64, device=dev)
        dist.all_reduce(agg)
        n_opt, mean_it = int(agg[0].item()), float(agg[1].item()) / (B * world)

    # setmodel! path (K1 + K2 re-condensation of all models), reported next to the step time
    hd.recondense_device(stream=stream.cuda_stream)
    torch.cuda.synchronize()
    recond_ms = hd.last_condense_ms()

    if rank == 0:
        total = B * world * args.steps
        value = total / elapsed
        # row counts of the reference's A Z̃ <= b (i_b rule); the kernel merges the U rows of a
        # move-blocking interval, the flop model keeps the reference's count
        rows_u = (int(np.isfinite(cfg.umin)) + int(np.isfinite(cfg.umax))) * hd.nU
        rows_y = (int(np.isfinite(cfg.ymin)) + int(np.isfinite(cfg.ymax))) * hd.nY
        flops, w_grad, w_iter = algorithmic_flops(cfg, mean_it, rows_u, rows_y)
        kms = float(np.mean(kern_ms))
        achieved = flops * B / (kms * 1e-3) / 1e12
        traffic = None      # HBM bytes per launch from the PMC passes committed under profiles/
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_k_step.json")))
            if tr["config"] == args.config and tr["batch"] == B:
                traffic = tr["hbm_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "QP solves/sec (moveinput!)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg.name, "batch_per_gpu": B, "global_batch": B * world,
                       "nx": cfg.nx, "nxhat": nxh, "nu": nu, "ny": ny, "Hp": Hp, "Hc": Hc,
                       "nZ": hd.nZ, "rows": int(rows_u + rows_y + neps), "cold_start": True,
                       "ipm_mean_iters": mean_it, "optimal_fraction": n_opt / (B * world),
                       "recondense_ms": recond_ms,
                       "sharding": "contiguous index ranges, no collective on the data path"},
            "roofline": {"bound": "mfma", "kernel": "k_step", "achieved": achieved,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel_ms": kms, "flops_per_solve": flops,
                         "hbm_algorithmic_GBps": algorithmic_bytes(cfg) * B / (kms * 1e-3) / 1e9,
                         "note": "FP64 vector/matrix peak (no f64 entry in the MFMA table: "
                                 "half the 157.3 TF FP32 rate); flops = W_grad + I W_iter, "
                                 "SURVEY 8(d) structure-exploiting count"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cfg, args):
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
    n = int(min(args.batch, max(probe, rate * args.cpu_seconds)))
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
                      f"mean {float(it.mean()):.1f} IPM iterations, all optimal: {bool((st == 0).all())}"}


if __name__ == "__main__":
    main()
