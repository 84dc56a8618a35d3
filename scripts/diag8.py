import sys, numpy as np, warnings
sys.path.insert(0,'.')
warnings.filterwarnings("ignore")
import mpcqp
from tests.parity_util import run_random_case
import io, contextlib
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    kw = dict(large=True) if len(sys.argv) > 3 and sys.argv[3] == "large" else {}
    try:
        r = run_random_case(seed, B=2, **kw)
        print(seed, "ok", r, flush=True)
    except AssertionError as e:
        print(seed, "FAIL", str(e)[:80], flush=True)
