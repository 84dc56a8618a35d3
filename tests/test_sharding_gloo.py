"""N > 1 paths on CPU, through PRODUCT code (the CPU wave emulator stands in for the device, see
tests/emu): (1) two `gloo` ranks each build their shard with `sharding.shard_range`, step it through
the C-ABI -- the period's inputs arrive from rank 0 through `sharding.scatter` -- and collect the batch with
`sharding.gather` (what bench.py does per GPU, RCCL there);
(2) one process drives two "devices" through the library's multi-device entry points
(`mpcqp_multi_*`: slicing, concurrent steps, gather into the caller's arrays).  Both must reproduce
the unsharded run bit for bit: no collective on the data path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mpcqp
from mpcqp import sharding, synth

EMU = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libmpcqp_emu.so")
CFG = synth.Config("shard", nx=3, nu=2, ny=2, Hp=8, Hc=3, umin=-0.6, umax=0.7, ymax=0.9)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _handle(cfg, bt, lib, multi=None):
    B = bt["xhat0"].shape[0]
    kw = dict(neps=1, flags=mpcqp.FLAG_RY_CONSTANT | mpcqp.FLAG_COLD_START, lib=lib)
    hd = (mpcqp.MultiHandle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, multi, **kw) if multi
          else mpcqp.Handle(B, cfg.nxh, cfg.nu, cfg.ny, 0, cfg.Hp, cfg.Hc, **kw))
    hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
    hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt),
                   np.full(B, cfg.Cwt))
    hd.set_bounds(U0min=np.full((B, hd.nU), cfg.umin), U0max=np.full((B, hd.nU), cfg.umax),
                  Y0max=np.full((B, hd.nY), cfg.ymax))
    hd.prepare()
    return hd


def _solve(cfg, bt, lib, multi=None):
    hd = _handle(cfg, bt, lib, multi)
    Z = np.zeros((bt["xhat0"].shape[0], hd.nZ))
    u0, st, it = hd.step(bt["xhat0"], bt["lastu0"], bt["ry"], Z)
    return Z, u0, st


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = mpcqp.api.load_library(EMU)
    lo, n = sharding.shard_range(B, rank, world)                       # this rank's shard
    bt = synth.make_batch(CFG, n, seed=4, lo=lo)
    # the scatter of the north star: this period's inputs are born on rank 0 (whole batch) and reach the ranks as
    # their contiguous slices, one scatter collective per array
    whole = synth.make_batch(CFG, B, seed=4) if rank == 0 else None
    for key in ("xhat0", "lastu0", "ry"):
        got = sharding.scatter(whole[key] if rank == 0 else None, B, dist, src=0, like=bt[key])
        assert got.shape == bt[key].shape and np.array_equal(got, bt[key]), key
        bt[key] = got
    Z, u0, st = _solve(CFG, bt, lib)
    Zall = sharding.gather(Z, B, dist)
    stall = sharding.gather(st, B, dist)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                           # bench.py's max-over-ranks
    if rank == 0:
        q.put((Zall, stall, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def emulib():
    if not os.path.exists(EMU):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EMU)])
    return mpcqp.api.load_library(EMU)


def test_shard_range_partitions_the_batch():
    for B, world in ((262144, 8), (7, 3), (5, 5), (65536, 3)):
        parts = [sharding.shard_range(B, r, world) for r in range(world)]
        assert parts[0][0] == 0 and sum(n for _, n in parts) == B
        assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        assert max(n for _, n in parts) - min(n for _, n in parts) <= 1
    assert sharding.shard_range(262144, 7, 8) == (7 * 32768, 32768)    # BASELINE config 4


def test_two_rank_sharding_equals_single_run(emulib):
    world, B = 2, 5                  # odd: the shards differ by one controller
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    Zall, stall, tmax = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bt = synth.make_batch(CFG, B, seed=4)
    Z1, _, st1 = _solve(CFG, bt, emulib)
    assert np.array_equal(Zall, Z1) and np.array_equal(stall, st1) and np.all(st1 == 0)
    assert tmax == world


def test_multi_device_handle_equals_single_handle(emulib):
    """mpcqp_multi_*: two shards (the emulator's one device twice) behind one handle."""
    B = 5
    bt = synth.make_batch(CFG, B, seed=9)
    Z1, u1, st1 = _solve(CFG, bt, emulib)
    Z2, u2, st2 = _solve(CFG, bt, emulib, multi=[0, 0])
    assert np.array_equal(Z1, Z2) and np.array_equal(u1, u2) and np.array_equal(st1, st2)
    mh = _handle(CFG, bt, emulib, multi=[0, 0, 0])
    assert [mh.shard(g) for g in range(3)] == [sharding.shard_range(B, g, 3) for g in range(3)]
