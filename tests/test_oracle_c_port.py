"""The oracle's C restatement (oracle/linmpc_ref.c: cpu_baseline "port" of bench.py and large-batch
checker) against the certified NumPy oracle on the two BASELINE configurations."""
import numpy as np
import pytest

from mpcqp import synth
from oracle import cport
from tests.parity_util import oracle_batch, rel_err


@pytest.mark.parametrize("name,B", [("C2", 48), ("C3", 24)])
def test_c_port_matches_numpy_oracle(name, B):
    cfg = synth.CONFIGS[name]
    bt = synth.make_batch(cfg, B, seed=5)
    rb = cport.from_synth(cfg, bt)
    Z, u0, st, it = rb.step(bt["xhat0"], bt["lastu0"], bt["ry"])
    ref = oracle_batch(cfg, bt)
    assert np.all(st == 0)
    err = rel_err(Z, ref["Z"], cfg.nu * cfg.Hc)
    assert err[ref["certified"]].max() <= 1e-5
    assert np.abs(u0 - ref["u"]).max() <= 1e-5 * max(1.0, np.abs(ref["u"]).max())
    # warm-started second period: same shift rule as set_warmstart_mpc!
    Z2, _, st2, it2 = rb.step(bt["xhat0"], u0, bt["ry"], Z=Z.copy(), cold=False)
    assert np.all(st2 == 0)
