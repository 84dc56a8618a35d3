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


# ---------------------------------------------------------------------------------------------
# Linear MovingHorizonEstimator workloads (BASELINE.json configs[4])
@dataclass
class MheConfig:
    """Dimensions, covariances and per-channel hard bounds of one synthetic MHE workload.  The
    augmented model is a stable plant with one integrator per measured output
    (estimator/construct.jl:365-376), so nx̂ = nx + nym."""
    name: str
    nx: int
    nu: int
    nym: int
    nd: int
    He: int
    direct: bool = True
    sigmaQ: float = 0.1
    sigmaQint: float = 0.1
    sigmaR: float = 0.2
    sigmaP0: float = 1.0
    xabs: float = np.inf       # |x̂0| <= xabs on every state (arrival and window)
    wabs: float = np.inf       # |ŵ| <= wabs
    vabs: float = np.inf       # |v̂| <= vabs
    Cwt: float = np.inf        # weight of the slack ε (Inf: hard constraints only)

    @property
    def nxh(self):
        return self.nx + self.nym


# BASELINE.json configs[4]: He = 20, nx̂ = 12 (8 plant states + 4 output integrators), hard state bounds
C5 = MheConfig("C5: linear MHE nx̂=12 (8+4 integrators) nu=4 nym=4 He=20, hard x̂ bounds", nx=8, nu=4, nym=4, nd=0,
               He=20, xabs=1.5)
# the same workload with the state bounds relaxed by the slack ε (finite Cwt, softness 1 on x̂min / x̂max): the soft kernel variant
C5S = MheConfig("C5 soft: linear MHE nx̂=12 nu=4 nym=4 He=20, x̂ bounds relaxed by ε (Cwt = 1e5)", nx=8, nu=4, nym=4, nd=0,
                He=20, xabs=1.5, Cwt=1e5)
MHE_CONFIGS = {"C5": C5, "C5S": C5S}


def get_mhe_config(name: str) -> MheConfig:
    """"C5", or "nx,nu,nym,nd,He" for a C5-style workload of other dimensions."""
    if name in MHE_CONFIGS:
        return MHE_CONFIGS[name]
    nx, nu, nym, nd, He = (int(v) for v in name.split(","))
    return MheConfig(f"custom MHE: nx={nx} nu={nu} nym={nym} nd={nd} He={He}, hard x̂ bounds", nx=nx, nu=nu, nym=nym,
                     nd=nd, He=He, xabs=1.5)


def make_mhe_batch(cfg: MheConfig, B: int, seed: int = 0, lo: int = 0):
    """Estimators lo .. lo+B-1 of the seeded workload: Ahat (B,nx̂,nx̂) Bhu (B,nx̂,nu) Chm (B,nym,nx̂)
    Bhd (B,nx̂,nd) Dhdm (B,nym,nd) Qhat (B,nx̂,nx̂) Rhat (B,nym,nym) P0 (B,nx̂,nx̂), plus the plant
    (A, Bu, C, Bd) for make_mhe_data."""
    nx, nu, nym, nd, nxh = cfg.nx, cfg.nu, cfg.nym, cfg.nd, cfg.nxh
    keys = ("Ahat", "Bhu", "Chm", "Bhd", "Dhdm", "A", "Bu", "C", "Bd")
    out = {k: [] for k in keys}
    c0, c1 = lo // CHUNK, (lo + B - 1) // CHUNK
    for c in range(c0, c1 + 1):
        rng = np.random.default_rng([seed, c, 5])
        A = _stable_A(rng, nx, CHUNK)
        Bu = rng.standard_normal((CHUNK, nx, nu)) / np.sqrt(nx)
        C = rng.standard_normal((CHUNK, nym, nx)) / np.sqrt(nx)
        Bd = rng.standard_normal((CHUNK, nx, nd)) / np.sqrt(nx)
        Ah = np.zeros((CHUNK, nxh, nxh))
        Ah[:, :nx, :nx] = A
        Ah[:, nx:, nx:] = np.eye(nym)
        Bh = np.zeros((CHUNK, nxh, nu)); Bh[:, :nx] = Bu
        Bhd = np.zeros((CHUNK, nxh, nd)); Bhd[:, :nx] = Bd
        Ch = np.concatenate([C, np.broadcast_to(np.eye(nym), (CHUNK, nym, nym))], axis=2)
        for k, v in zip(keys, (Ah, Bh, Ch, Bhd, np.zeros((CHUNK, nym, nd)), A, Bu, C, Bd)):
            out[k].append(v)
    a, b = lo - c0 * CHUNK, lo - c0 * CHUNK + B
    res = {k: np.ascontiguousarray(np.concatenate(v)[a:b]) for k, v in out.items()}
    sq = np.concatenate([np.full(nx, cfg.sigmaQ), np.full(nym, cfg.sigmaQint)]) ** 2
    res["Qhat"] = np.broadcast_to(np.diag(sq), (B, nxh, nxh)).copy()
    res["Rhat"] = np.broadcast_to(np.eye(nym) * cfg.sigmaR ** 2, (B, nym, nym)).copy()
    res["P0"] = np.broadcast_to(np.eye(nxh) * cfg.sigmaP0 ** 2, (B, nxh, nxh)).copy()
    res["cfg"] = cfg
    return res


def make_mhe_data(cfg: MheConfig, bt, nper: int, seed: int = 0, lo: int = 0):
    """nper periods of plant data for every estimator of `bt`: Y (nper,B,nym), U (nper,B,nu), D (nper,B,nd)
    in deviation variables: x+ = A x + Bu u + Bd d + w, y = C x + bias + v with a slowly drifting output bias."""
    B = bt["A"].shape[0]
    rng = np.random.default_rng([seed, lo, 55])
    x = 0.5 * rng.standard_normal((B, cfg.nx))
    bias = 0.3 * rng.standard_normal((B, cfg.nym))
    u = 0.5 * rng.standard_normal((B, cfg.nu))
    Y, U, D = [], [], []
    for _ in range(nper):
        d = 0.5 * rng.standard_normal((B, cfg.nd))
        y = np.einsum("bij,bj->bi", bt["C"], x) + bias + cfg.sigmaR * rng.standard_normal((B, cfg.nym))
        Y.append(y); U.append(u.copy()); D.append(d)
        x = (np.einsum("bij,bj->bi", bt["A"], x) + np.einsum("bij,bj->bi", bt["Bu"], u)
             + np.einsum("bij,bj->bi", bt["Bd"], d) + cfg.sigmaQ * rng.standard_normal((B, cfg.nx)))
        bias = bias + cfg.sigmaQint * rng.standard_normal((B, cfg.nym))
        u = 0.9 * u + 0.3 * rng.standard_normal((B, cfg.nu))
    return np.array(Y), np.array(U), np.array(D)
