import sys, os, numpy as np, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
from tests.parity_util import run_random_case, run_random_case2
worst = 0.0
for s in range(5000, 5300):
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            e = run_random_case(s, B=2)
        if e is not None: worst = max(worst, e)
        if e is not None and e > 1e-6: print(s, "ERR", e)
    except AssertionError as ex: print(s, "ASSERT", ex)
print("worst families", worst)
worst = 0.0
for s in range(5000, 5060):
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            e = run_random_case2(s, B=2)
        if e is not None: worst = max(worst, e)
        if e is not None and e > 1e-6: print(s, "ERR2", e)
    except AssertionError as ex: print(s, "ASSERT2", ex)
print("worst horizon-wide", worst)
