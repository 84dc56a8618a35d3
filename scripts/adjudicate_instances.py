"""Adjudicate ill-conditioned instances in extended precision.

    python scripts/adjudicate_instances.py            (CPU only; minutes)

For every (shape, seed, instance) of CASES: the float64 oracle (oracle/qp.py), the oracle's C port (the kernel's own
algorithm on a dense matrix) and the 60-digit interior-point solve of oracle/qp_hp.py on the SAME float64 QP data.
Writes tests/golden/hp_optima.json: the extended-precision optimum (float64-rounded), its KKT residuals / error bound,
and how far the two float64 answers are from it.  The GPU suite compares the kernel with these optima
(tests/test_gpu_parity.py::test_ill_conditioned_instances_against_extended_precision_optimum).
"""
import dataclasses
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpcqp import synth                      # noqa: E402
from oracle import cport, qp, qp_hp          # noqa: E402
from tests.parity_util import make_oracle    # noqa: E402

# (shape "nx,nu,ny,Hp,Hc" with C3-style rows, seed, instances): the two the round-3 sweeps flagged + neighbours
CASES = [("8,2,2,60,40", 11, [99]), ("8,2,2,64,60", 11, None)]


def rel(a, b, nDU):
    return float(np.abs(a[:nDU] - b[:nDU]).max() / max(1.0, np.abs(b[:nDU]).max()))


def main():
    out = {"note": "extended-precision optima of ill-conditioned instances (oracle/qp_hp.py, 60 digits); generator: "
                   "scripts/adjudicate_instances.py", "cases": []}
    quick = "--quick" in sys.argv
    for name, seed, idx in CASES:
        cfg = synth.get_config(name)
        B = 256
        bt = synth.make_batch(cfg, B, seed=seed)
        Zc, _, stc, itc = cport.from_synth(cfg, bt).step(bt["xhat0"], bt["lastu0"], bt["ry"])
        nDU = cfg.nu * cfg.Hc
        if idx is None:
            # the instances where the float64 oracle has no active-set certificate or disagrees with the C port
            idx = []
            for i in range(B):
                m = make_oracle(cfg, bt, i)
                m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i]); m.linconstraint()
                z, st, info = qp.solve_qp(*m.qp_data(), m.warmstart(), return_info=True)
                e = rel(Zc[i], z, nDU)
                if info["certificate"] != "active-set" or e > 1e-6:
                    idx.append(i)
                    print(f"{name} instance {i}: oracle certificate {info['certificate']}, C port vs oracle {e:.2e}", flush=True)
            idx = idx[:2 if quick else 6]
        for i in idx:
            m = make_oracle(cfg, bt, i)
            m.initpred(bt["xhat0"][i], bt["lastu0"][i], bt["ry"][i]); m.linconstraint()
            data = m.qp_data()
            zo, sto, info = qp.solve_qp(*data, m.warmstart(), return_info=True)
            t0 = time.time()
            zh, ih = qp_hp.solve_reference_qp(*data, z0=m.warmstart(), digits=60, verbose="-v" in sys.argv)
            J = lambda z: float(0.5 * z @ data[0] @ z + data[1] @ z)
            rec = dict(shape=name, seed=seed, instance=int(i), nZ=len(zh), z=zh.tolist(), hp_status=ih["status"], hp_iters=ih["iters"],
                       hp_stationarity=ih["stationarity"], hp_gap=ih["gap"], hp_violation=ih["violation"], hp_err_bound=ih["err_bound"],
                       oracle_certificate=info["certificate"], oracle_vs_hp=rel(zo, zh, nDU), cport_vs_hp=rel(Zc[i], zh, nDU),
                       cport_status=int(stc[i]), cport_iters=int(itc[i]), cond_H=float(np.linalg.cond(data[0])),
                       J_hp=J(zh), J_oracle=J(zo), J_cport=J(Zc[i]))
            out["cases"].append(rec)
            print({k: v for k, v in rec.items() if k != "z"}, f"({time.time() - t0:.0f} s)", flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "hp_optima.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    main()
