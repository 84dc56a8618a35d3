"""Generate the golden fixtures of tests/golden/ (run once, committed together with its output).

The reference (Julia + JuMP/OSQP) cannot run in this image, so -- as SURVEY.md 8(c) prescribes --
the vectors come from the build's own CPU oracle (oracle/condense.py + oracle/qp.py, itself pinned on
the reference's known answers T1-T8 by tests/test_oracle_known_answers.py): for the BASELINE
configurations C2 and C3, seeds 0-3, 8 instances each.  A fixture holds the inputs of every
instance (augmented model, state, last input, set point, the configuration's weights and bounds)
and the expected outputs (H~, q~, F, certified optimum Z~*, first move u0, whether the optimum
carries an exact active-set certificate).  Only data, no code, is stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcqp import synth                      # noqa: E402
from tests.parity_util import oracle_batch   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PER_SEED = 8

for name in ("C2", "C3"):
    cfg = synth.CONFIGS[name]
    parts, outs = [], []
    for seed in range(4):
        bt = synth.make_batch(cfg, PER_SEED, seed=seed)
        parts.append(bt)
        outs.append(oracle_batch(cfg, bt))
    data = {"in_" + k: np.concatenate([p[k] for p in parts]) for k in parts[0] if k != "cfg"}
    data.update({"out_" + k: np.concatenate([o[k] for o in outs]) for k in outs[0]})
    data["seeds"] = np.repeat(np.arange(4), PER_SEED)
    for k in ("nx", "nu", "ny", "Hp", "Hc", "Mwt", "Nwt", "Lwt", "Cwt", "umin", "umax", "dumin", "dumax", "ymin", "ymax"):
        data["cfg_" + k] = np.float64(getattr(cfg, k))
    path = os.path.join(HERE, f"{name}_seed0-3.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes;", int(data["out_certified"].sum()), "of", len(data["seeds"]),
          "optima certified")
