"""Sweep of randomised controller families on the GPU against the certified oracle optimum
(tests/parity_util.run_random_case): prints the relative ΔU error of every family.
Usage: python scripts/family_sweep.py FIRST LAST [small|large|huge|any] [B] [MultipleShooting]"""
import sys, warnings
sys.path.insert(0, '.')
warnings.filterwarnings("ignore")
import os
import tests.parity_util as pu
from tests.parity_util import run_random_case
if "DR" in os.environ:          # dual regularisation of the kernel (default: the kernel's own)
    pu.EXTRA_KW = dict(dual_reg=float(os.environ["DR"]))

kind = sys.argv[3] if len(sys.argv) > 3 else ""
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
kw = {kind: True} if kind in ("small", "large", "huge") else {}
if len(sys.argv) > 5:
    kw["transcription"] = sys.argv[5]
worst = 0.0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        r = run_random_case(seed, B=B, **kw)
        worst = max(worst, r or 0.0)
        print(seed, "ok", r, flush=True)
    except AssertionError as e:
        print(seed, "FAIL", str(e)[:100], flush=True)
print("worst", worst)
