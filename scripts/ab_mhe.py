"""A/B several builds of the library on the C5 estimator period (bench_mhe workload): python scripts/ab_mhe.py libA.so libB.so ..."""
import sys, os, numpy as np
sys.path.insert(0, '.')
import torch
import mpcqp
from mpcqp import mhe as pm, synth
import bench_mhe
cfg = synth.get_mhe_config(os.environ.get("AB_CFG", "C5")); B = 65536
xref = None
for path in sys.argv[1:]:
    mpcqp.api._lib = None
    mpcqp.api.load_library(os.path.abspath(path))
    periods = cfg.He + 8
    sh = bench_mhe.MheShard(cfg, 0, B, 0, 0, periods)
    for _ in range(cfg.He + 2):
        sh.step()
    sh.h.sync()
    for _ in range(5):
        sh.step(record=True)
    x = sh.h.get(pm.GET_XHAT0); st = sh.h.get(pm.GET_STATUS); it = sh.h.get(pm.GET_ITERS)
    if xref is None: xref = x.copy()
    print(f"{os.path.basename(path)}: k_mhe_step ms {['%.2f' % m for m in sh.kern_ms]} optimal {np.mean(st == 0):.6f} iters {it.mean() + 1:.3f} "
          f"max |x - first| {np.abs(x - xref).max():.2e}", flush=True)
    del sh
    torch.cuda.empty_cache()
