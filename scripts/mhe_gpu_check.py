"""Development check on the GPU: C5-style batches against oracle/mhe.py + timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mpcqp
from mpcqp import synth, mhe as pm
import mhe_util

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nper = int(sys.argv[3]) if len(sys.argv) > 3 else 24
cfg = synth.get_mhe_config(cfgname)
bt = synth.make_mhe_batch(cfg, B, seed=1)
members = list(range(0, B, max(1, B // 6)))[:6]
t0 = time.time()
rows, bm = mhe_util.run_periods(cfg, bt, nper, members)
for r in rows:
    print(f"k={r['k']:2d} Nk={r['Nk']:2d} ex={r['ex']:.2e} ew={r['ew']:.2e} eP={r['ep']:.2e} bad={int((r['status']!=0).sum())} "
          f"it={r['iters'].mean():.1f}/{r['iters'].max()} ms={bm.handle.last_ms():.3f}")
print("worst", max(r["ex"] for r in rows), max(r["ew"] for r in rows), f"{time.time()-t0:.1f}s")
