"""Sweep of randomised SMALL controller families through the product code on the CPU wave emulator (tests/emu, fibers) against the
certified oracle optimum: python scripts/emu_family_sweep.py FIRST LAST   (a quarter of a minute per 400 families).  Counts the
kernel kinds: 3 = the small-problem kernel incl. its dense-row variant (output-bound / terminal rows), 0 = runtime-dimension kernel
(the emulator has no on-demand specialisations).  A family without a certified oracle step is reported as 'no certificate'."""
import sys, warnings, collections, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import mpcqp
from tests.parity_util import run_random_case
lib = mpcqp.api.load_library(os.path.join(ROOT, "tests", "emu", "libmpcqp_emu.so"))
first, last = int(sys.argv[1]), int(sys.argv[2])
worst, kc, fails, nocert = 0.0, collections.Counter(), [], []
for seed in range(first, last):
    kinds = []
    try:
        e = run_random_case(seed, lib=lib, B=2, small=True, kinds=kinds)
        kc.update(k if not isinstance(k, tuple) else k[0] for k in kinds)
        if e is None:
            nocert.append(seed)
        else:
            worst = max(worst, e)
            if e > 1e-5:
                fails.append((seed, e))
    except AssertionError as ex:
        fails.append((seed, str(ex)[:80]))
print(f"families {last - first}: worst {worst:.3e}, kernel kinds {dict(kc)}, no certificate {nocert}, failures {fails}")
