"""Sweep of randomised controller families on the GPU against the certified oracle optimum
(tests/parity_util.run_random_case): prints the relative ΔU error of every family.
Usage: python scripts/family_sweep.py FIRST LAST [small|large|huge|huge2|ny4|any] [B] [MultipleShooting]
(huge: 64 < nZ~ <= 130, huge2: 130 < nZ~ <= 165 -- the team-of-wavefronts kernels of round 6; the kernel kind and nZ~ are printed)"""
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
kw = {kind: True} if kind in ("small", "large", "huge", "huge2", "ny4") else {}
if len(sys.argv) > 5:
    kw["transcription"] = sys.argv[5]
worst = 0.0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        kinds = []
        r = run_random_case(seed, B=B, kinds=kinds, **kw)
        worst = max(worst, r or 0.0)
        print(seed, "ok", r, "kernel kind / nZ", kinds, flush=True)
    except AssertionError as e:
        print(seed, "FAIL", str(e)[:100], flush=True)
print("worst", worst)
