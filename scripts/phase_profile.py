"""Per-phase cycle breakdown of k_step (profiling build: lib/libmpcqp_prof.so, -DMPCQP_PROFILE)."""
import sys, os, numpy as np
sys.path.insert(0, '.')
import mpcqp
from mpcqp import synth
from tests.parity_util import make_controller
lib = mpcqp.api.load_library(os.path.join('modelpredictivecontrol.jl_amd', 'lib', 'libmpcqp_prof.so'))
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cfg = synth.get_config(name); bt = synth.make_batch(cfg, B, seed=0)
mpc = make_controller(cfg, bt, lib=lib, cold_start=True)
mpc.lastu0 = bt["lastu0"].copy()
for rep in range(2):
    mpc.moveinput(bt["xhat0"], bt["ry"])
out = np.empty((B, 16)); 
import ctypes as C
mpcqp.api._chk(lib, lib.mpcqp_get(mpc.hd.h, 99, out.ctypes.data_as(C.c_void_p)))
names = ["apply_G", "apply_Gt", "loadH+Hz", "GtDG rows", "EtDE(mfma)", "GtDG struct", "cholesky", "solve", "", "", "", "", "", "", "", "run total"]
sub = {8: "  Gt: rows+reduce", 9: "  Gt: box/U/eps part", 10: "  Gt: Et_apply", 11: "  G: ucum", 12: "  G: E_apply",
       13: "  step rule + update pass", 14: "  polish (all of it)"}
it = mpc.iters.mean() + 1
tot = out[:, 15].mean()
print(f"{name} B={B} kernel {mpc.hd.last_step_ms():.2f} ms, mean iters {mpc.iters.mean():.2f}; mean cycles per wave: {tot:.0f} ({tot/it:.0f} per iteration)")
acc = 0
for i, n in enumerate(names):
    if n and i < 15:
        v = out[:, i].mean(); acc += v
        print(f"  {n:12s} {v:12.0f} cyc  {100*v/tot:5.1f}%   per-iter {v/it:9.0f}")
for i, n in sub.items():
    v = out[:, i].mean(); print(f"  {n:22s} {v:12.0f} cyc  {100*v/tot:5.1f}%   per-iter {v/it:9.0f}")
print(f"  {'other(rows,..)':12s} {tot-acc:12.0f} cyc  {100*(tot-acc)/tot:5.1f}%   per-iter {(tot-acc)/it:9.0f}")
