"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle):
inputs and expected H~, q~, F, certified optimum and first move of 32 instances of each BASELINE
configuration.  CPU: the oracle, its C port and the fixture agree (guards the checker against
drift).  GPU: the HIP path through the C-ABI reproduces the fixture."""
import os

import numpy as np
import pytest

import mpcqp
from mpcqp import synth
from tests.parity_util import make_oracle, rel_err, run_batch
from oracle import qp

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-5          # relative dU error, BASELINE.md section 4


def load(name):
    g = np.load(os.path.join(HERE, "golden", f"{name}_seed0-3.npz"))
    cfg = synth.CONFIGS[name]
    for k in ("nx", "nu", "ny", "Hp", "Hc", "Mwt", "Nwt", "Lwt", "Cwt", "umin", "umax", "dumin", "dumax",
              "ymin", "ymax"):
        a, b = float(g["cfg_" + k]), float(getattr(cfg, k))
        assert a == b or (np.isinf(a) and np.isinf(b) and a * b > 0), f"configuration {name}.{k} drifted"
    bt = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    out = {k[4:]: g[k] for k in g.files if k.startswith("out_")}
    return cfg, bt, out


@pytest.mark.parametrize("name", ["C2", "C3"])
def test_oracle_reproduces_golden(name):
    cfg, bt, out = load(name)
    assert out["certified"].all()
    for i in range(0, 32, 5):
        m = make_oracle(cfg, bt, i)
        m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i])
        m.linconstraint()
        assert np.abs(m.Ht - out["H"][i]).max() <= 1e-12 * np.abs(out["H"][i]).max()
        assert np.abs(m.qt - out["q"][i]).max() <= 1e-12 * max(1.0, np.abs(out["q"][i]).max())
        z, st = qp.solve_qp(*m.qp_data(), m.warmstart())
        assert st == 0
        assert np.abs(z - out["Z"][i]).max() <= 1e-9 * max(1.0, np.abs(out["Z"][i]).max())


@pytest.mark.parametrize("name", ["C2", "C3"])
def test_c_port_reproduces_golden(name):
    from oracle import cport
    cfg, bt, out = load(name)
    rb = cport.from_synth(cfg, bt)
    Z, u0, st, it = rb.step(bt["xhat0"], bt["lastu0"], bt["ry"])
    assert (st == 0).all()
    nDU = cfg.nu * cfg.Hc
    assert rel_err(Z, out["Z"], nDU).max() <= TOL
    assert np.abs(u0 - out["u"]).max() <= TOL * max(1.0, np.abs(out["u"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["C2", "C3"])
def test_gpu_reproduces_golden(name, hiplib):
    cfg, bt, out = load(name)
    got = run_batch(cfg, bt, keep_qp=True)
    hd = got["mpc"].hd
    H, F, q = hd.get(mpcqp.GET_HESSIAN), hd.get(mpcqp.GET_FVEC), hd.get(mpcqp.GET_QTILDE)
    assert np.abs(H - out["H"]).max() <= 1e-12 * np.abs(out["H"]).max()
    assert np.abs(F - out["F"]).max() <= 1e-11 * max(1.0, np.abs(out["F"]).max())
    assert np.abs(q - out["q"]).max() <= 1e-11 * max(1.0, np.abs(out["q"]).max())
    assert (got["status"] == 0).all()
    nDU = cfg.nu * cfg.Hc
    assert rel_err(got["Z"], out["Z"], nDU).max() <= TOL
    assert np.abs(got["u"] - out["u"]).max() <= TOL * max(1.0, np.abs(out["u"]).max())
