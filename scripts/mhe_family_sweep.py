"""Sweep of randomised MovingHorizonEstimator families on the GPU against oracle/mhe.py (tests/mhe_util.random_family).
Usage: python scripts/mhe_family_sweep.py FIRST LAST"""
import sys, warnings
sys.path.insert(0, '.')
warnings.filterwarnings("ignore")
from tests import mhe_util

worst, n, nf = 0.0, 0, 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        w, c, f = mhe_util.random_family(seed)
        worst, n, nf = max(worst, w), n + c, nf + f
        print(seed, "ok", f"{w:.2e}", c, f, flush=True)
    except AssertionError as e:
        print(seed, "FAIL", str(e)[:200], flush=True)
print("worst", worst, "solves compared", n, "infeasible windows", nf)
