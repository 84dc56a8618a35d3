"""Seeded synthetic batches of independent LinMPC instances.

Pure NumPy, no oracle and no device code: shared by tests, bench.py and smoke().  Problems are
generated in chunks of CHUNK instances from `default_rng([seed, chunk])`, so a shard
[lo, hi) of a big batch holds the same instances whatever the number of ranks.

Arrays are returned "logical" (B, rows, cols); the C-ABI wants column-major per problem
(Julia Array{Float64,3} (rows, cols, B)) -- `api.py` does that transposition.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

CHUNK = 256


@dataclass
class Config:
    """Dimensions and constraint pattern of one synthetic workload."""
    name: str
    nx: int
    nu: int
    ny: int
    Hp: int
    Hc: int
    Cwt: float = 1e5
    umin: float = -np.inf
    umax: float = np.inf
    dumin: float = -np.inf
    dumax: float = np.inf
    ymin: float = -np.inf
    ymax: float = np.inf
    Mwt: float = 1.0
    Nwt: float = 0.1
    Lwt: float = 0.0

    @property
    def nxh(self):  # one output integrator per output (estimator/construct.jl:365-376)
        return self.nx + self.ny


# BASELINE.json configs[1] and configs[2]; ny = 4 pinned for C3 (SURVEY 8 preamble)
C2 = Config("C2: nx=4 nu=2 ny=2 Hp=20 Hc=5, hard u/du box", nx=4, nu=2, ny=2, Hp=20, Hc=5,
            umin=-1.0, umax=1.0, dumin=-0.2, dumax=0.2)
C3 = Config("C3: nx=12 nu=4 ny=4 Hp=30 Hc=10, soft ymax + hard umin/umax", nx=12, nu=4, ny=4,
            Hp=30, Hc=10, umin=-1.0, umax=1.0, ymax=1.0)
CONFIGS = {"C2": C2, "C3": C3}


def get_config(name: str) -> Config:
    """A named BASELINE config, or "nx,nu,ny,Hp,Hc" for a C3-style workload (soft ymax, hard
    umin/umax) of other dimensions -- used to time specialisations other than the headline one."""
    if name in CONFIGS:
        return CONFIGS[name]
    nx, nu, ny, Hp, Hc = (int(v) for v in name.split(","))
    return Config(f"custom: nx={nx} nu={nu} ny={ny} Hp={Hp} Hc={Hc}, soft ymax + hard umin/umax",
                  nx=nx, nu=nu, ny=ny, Hp=Hp, Hc=Hc, umin=-1.0, umax=1.0, ymax=1.0)


def _stable_A(rng, n, nb):
    """nb random stable matrices: real block-diagonal (1x1 and 2x2 rotation blocks) with moduli
    U(0.5, 0.98), rotated by a random orthogonal similarity."""
    A = np.zeros((nb, n, n))
    rho = rng.uniform(0.5, 0.98, size=(nb, n))
    theta = rng.uniform(0.0, np.pi, size=(nb, n))
    cplx = rng.random(size=(nb, n)) < 0.5
    sign = np.where(rng.random(size=(nb, n)) < 0.5, -1.0, 1.0)
    for b in range(nb):
        i = 0
        while i < n:
            if i + 1 < n and cplx[b, i]:
                r, t = rho[b, i], theta[b, i]
                A[b, i:i + 2, i:i + 2] = r * np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
                i += 2
            else:
                A[b, i, i] = rho[b, i] * sign[b, i]
                i += 1
    Q, _ = np.linalg.qr(rng.standard_normal((nb, n, n)))
    return Q @ A @ Q.transpose(0, 2, 1)


def make_batch(cfg: Config, B: int, seed: int = 0, lo: int = 0):
    """Instances lo .. lo+B-1 of the seeded workload `cfg`.  Returns a dict of float64 arrays:

    Ahat (B,nxh,nxh)  Bhu (B,nxh,nu)  Chat (B,ny,nxh)  xhat0 (B,nxh)  lastu0 (B,nu)  ry (B,ny)
    plus the (shared) weights and bounds of the config.
    """
    nx, nu, ny, nxh = cfg.nx, cfg.nu, cfg.ny, cfg.nxh
    out = {k: [] for k in ("Ahat", "Bhu", "Chat", "xhat0", "lastu0", "ry")}
    c0, c1 = lo // CHUNK, (lo + B - 1) // CHUNK
    for c in range(c0, c1 + 1):
        rng = np.random.default_rng([seed, c])
        A = _stable_A(rng, nx, CHUNK)
        Bu = rng.standard_normal((CHUNK, nx, nu)) / np.sqrt(nx)
        C = rng.standard_normal((CHUNK, ny, nx)) / np.sqrt(nx)
        Ah = np.zeros((CHUNK, nxh, nxh))
        Ah[:, :nx, :nx] = A
        Ah[:, nx:, nx:] = np.eye(ny)
        Bh = np.zeros((CHUNK, nxh, nu))
        Bh[:, :nx] = Bu
        Ch = np.concatenate([C, np.broadcast_to(np.eye(ny), (CHUNK, ny, ny))], axis=2)
        out["Ahat"].append(Ah)
        out["Bhu"].append(Bh)
        out["Chat"].append(Ch)
        out["xhat0"].append(rng.standard_normal((CHUNK, nxh)))
        out["lastu0"].append(rng.uniform(-0.5, 0.5, (CHUNK, nu)))
        out["ry"].append(2.0 * rng.standard_normal((CHUNK, ny)))
    a, b = lo - c0 * CHUNK, lo - c0 * CHUNK + B
    res = {k: np.ascontiguousarray(np.concatenate(v)[a:b]) for k, v in out.items()}
    res["cfg"] = cfg
    return res
