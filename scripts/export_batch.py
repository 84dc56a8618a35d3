"""Export a seeded synthetic batch (the bench workload) for bench/ref_julia.jl:
   python scripts/export_batch.py C3 2048 /tmp/c3_batch   ->  /tmp/c3_batch.json + /tmp/c3_batch.bin
The .bin holds little-endian float64: A (B,nx,nx), Bu (B,nx,nu), C (B,ny,nx) of the PLANT models
(row-major), then xhat0 (B,nx+ny), lastu0 (B,nu), ry (B,ny)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpcqp import synth

name, B, prefix = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = synth.get_config(name)
bt = synth.make_batch(cfg, B, seed=0)
nx = cfg.nx
A = bt["Ahat"][:, :nx, :nx]; Bu = bt["Bhu"][:, :nx, :]; C = bt["Chat"][:, :, :nx]
parts = [A, Bu, C, bt["xhat0"], bt["lastu0"], bt["ry"]]
# the .bin is read by Julia as column-major (n, m, B): write every array with the batch index slowest
blob = np.concatenate([np.ascontiguousarray(p, dtype="<f8").reshape(-1) for p in parts])
blob.tofile(prefix + ".bin")
fin = lambda v: float(v) if np.isfinite(v) else None          # null = infinite (bound absent / Cwt = Inf)
hdr = dict(B=B, nx=nx, nu=cfg.nu, ny=cfg.ny, Hp=cfg.Hp, Hc=cfg.Hc, Mwt=cfg.Mwt, Nwt=cfg.Nwt, Lwt=cfg.Lwt,
           Cwt=fin(cfg.Cwt), umin=fin(cfg.umin), umax=fin(cfg.umax), dumin=fin(cfg.dumin), dumax=fin(cfg.dumax),
           ymin=fin(cfg.ymin), ymax=fin(cfg.ymax), doubles=int(blob.size), config=cfg.name)
json.dump(hdr, open(prefix + ".json", "w"))
print(f"wrote {prefix}.bin ({blob.size} doubles) and {prefix}.json")
