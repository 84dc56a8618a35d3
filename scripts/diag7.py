import sys, numpy as np
sys.path.insert(0,'.')
import mpcqp
from tests.parity_util import run_random_case
try:
    print("result", run_random_case(int(sys.argv[1]), B=3, large=True))
except AssertionError as e:
    print("ASSERT", e)
