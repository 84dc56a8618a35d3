"""The oracle's C restatement (oracle/linmpc_ref.c: cpu_baseline "port" of bench.py and large-batch
checker) against the certified NumPy oracle on the two BASELINE configurations -- and (round 6) on one shape beyond one row
per lane (nZ~ = 106) and one beyond two (nZ~ = 151, 141 with bounds on every group): the GPU tests of those specialisations
compare with this port at a few hundred instances per shape, so the port itself is pinned on the independent oracle there."""
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


@pytest.mark.parametrize("name,pattern,B", [("12,3,3,40,35", "c3", 6), ("12,3,3,50,50", "c3", 4), ("12,2,2,70,70", "all", 4)])
def test_c_port_matches_numpy_oracle_beyond_one_row_per_lane(name, pattern, B):
    import dataclasses
    pats = {"c3": {}, "all": dict(ymin=-1.2, ymax=1.0, dumin=-0.4, dumax=0.4)}
    cfg = dataclasses.replace(synth.get_config(name), **pats[pattern])
    assert cfg.nu * cfg.Hc + 1 > 64
    bt = synth.make_batch(cfg, B, seed=9)
    Z, u0, st, it = cport.from_synth(cfg, bt).step(bt["xhat0"], bt["lastu0"], bt["ry"])
    ref = oracle_batch(cfg, bt)
    assert np.all(st == 0)
    err = rel_err(Z, ref["Z"], cfg.nu * cfg.Hc)
    assert ref["certified"].sum() >= B - 1, ref["certified"]
    assert err[ref["certified"]].max() <= 1e-5, err
    assert err.max() <= 1e-5, err            # (uncertified oracle points carry the oracle's rigorous error bound)
