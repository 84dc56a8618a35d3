"""Dry run of the driver's multi-GPU bench launch at world size 8 on ONE GPU: the exact command
line of the contract (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8 ...`) with
MPCQP_BENCH_ONE_GPU=1, which maps every rank to device 0 and runs the collectives over gloo.  Everything the 8-GPU run
does is exercised -- BASELINE configs[3] (262 144 controllers split 32 768 per rank, `sharding.shard_range`), the
barrier / MAX-reduce timing, the scatter of one period's inputs from rank 0, the gather of statuses and first moves, the
weak-scaling leg -- except the xGMI links themselves.  The JSON line is kept under gpurun_out/ (copied to profiles/)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_world_size_8_on_one_gpu(hiplib):
    env = dict(os.environ, MPCQP_BENCH_ONE_GPU="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    rec = json.loads(lines[0])
    out = os.path.join(ROOT, "gpurun_out", "bench_world8_one_gpu.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write(lines[0] + "\n")
    assert rec["n_gpus"] == 8 and rec["steps"] == 2 and rec["scaling"] == "strong"
    cfg = rec["config"]
    assert cfg["global_batch"] == 262144 and cfg["batch_per_gpu"] == 32768
    assert cfg["optimal_fraction"] == 1.0
    assert cfg["scatter"]["matches_local_shard"] is True and cfg["gather"]["optimal_fraction"] == 1.0
    assert cfg["weak_scaling"]["global_batch"] == 8 * 65536 and cfg["weak_scaling"]["value"] > 0
    assert rec["value"] > 0 and rec["higher_is_better"] is True
