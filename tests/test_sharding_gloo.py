"""N > 1 path on CPU: two `gloo` ranks each take a contiguous shard of the seeded batch (exactly
what bench.py does per GPU), run it (oracle C port standing in for the device), and the gathered
result must equal the unsharded run -- no collective on the data path, only the final
gather/MAX-reduce that bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpcqp import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, Bper, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cport
    cfg = synth.C2
    bt = synth.make_batch(cfg, Bper, seed=4, lo=rank * Bper)           # this rank's shard
    rb = cport.from_synth(cfg, bt)
    Z, u0, st, it = rb.step(bt["xhat0"], bt["lastu0"], bt["ry"], nthreads=1)
    out = [torch.zeros((Bper, Z.shape[1]), dtype=torch.float64) for _ in range(world)]
    dist.all_gather(out, torch.from_numpy(Z))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                           # bench.py's max-over-ranks
    n_opt = torch.tensor([float((st == 0).sum())])
    dist.all_reduce(n_opt)
    if rank == 0:
        q.put((torch.cat(out).numpy(), float(t.item()), float(n_opt.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_run():
    world, Bper = 2, 300            # 300 is not a multiple of the generator's chunk (256)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, Bper, q)) for r in range(world)]
    for p in procs:
        p.start()
    Zall, tmax, n_opt = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import cport
    cfg = synth.C2
    bt = synth.make_batch(cfg, world * Bper, seed=4)
    Z1, _, st, _ = cport.from_synth(cfg, bt).step(bt["xhat0"], bt["lastu0"], bt["ry"], nthreads=1)
    assert np.array_equal(Zall, Z1)
    assert tmax == world and n_opt == world * Bper
