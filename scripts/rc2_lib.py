"""Developer tool: one randomised horizon-wide family (tests/parity_util.run_random_case2) on a given library build."""
import sys, os
sys.path.insert(0, '.')
import mpcqp
from tests.parity_util import run_random_case2
lib = mpcqp.api.load_library(os.path.abspath(sys.argv[1]))
for seed in [int(v) for v in sys.argv[2:]]:
    kinds = []
    print(sys.argv[1], seed, run_random_case2(seed, lib=lib, B=4, kinds=kinds), kinds, flush=True)
