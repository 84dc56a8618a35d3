"""Device-resident closed loop of B controllers (SURVEY 8f-1 measurement): per control period
    preparestate! (k_kf_correct) -> moveinput! (k_step, warm-started) -> updatestate! (k_kf_predict)
with x̂0, Z̃, lastu0 resident in HBM; the plant (the augmented model itself + measurement noise) is
advanced with torch.bmm, which is plumbing.  Reports periods/s and the HBM rate of the two Kalman
kernels (HBM-bound: they stream Â, B̂u, Ĉ, K̂ once per call) against 8 TB/s."""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import mpcqp
from mpcqp import synth

cfg = synth.C3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
WARM_DUAL = (sys.argv[3] != "0") if len(sys.argv) > 3 else True      # MPCQP_FLAG_WARM_DUAL
bt = synth.make_batch(cfg, B, seed=0)
nxh, nu, ny, Hp, Hc = cfg.nxh, cfg.nu, cfg.ny, cfg.Hp, cfg.Hc
hd = mpcqp.Handle(B, nxh, nu, ny, 0, Hp, Hc, neps=1,
                  flags=mpcqp.FLAG_RY_CONSTANT | (mpcqp.FLAG_WARM_DUAL if WARM_DUAL else 0))   # warm start on
hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
hd.set_weights(np.full((B, hd.nY), cfg.Mwt), np.full((B, hd.nDU), cfg.Nwt), np.full((B, hd.nU), cfg.Lwt), np.full(B, cfg.Cwt))
hd.set_bounds(U0min=np.full((B, hd.nU), cfg.umin), U0max=np.full((B, hd.nU), cfg.umax), Y0max=np.full((B, hd.nY), cfg.ymax))
Q = np.diag(np.r_[np.full(cfg.nx, 1.0 / cfg.nx), np.ones(ny)] ** 2)
nK = min(B, 2048)                                         # DARE on the host for a sample, tiled
K = mpcqp.steady_kalman_gain(bt["Ahat"][:nK], bt["Chat"][:nK], Q, np.eye(ny))
K = np.tile(K, (B // nK + 1, 1, 1))[:B] if nK < B else K
if nK < B:   # gains must match the models: recompute exactly only for the sample, reuse models too
    for k in ("Ahat", "Bhu", "Chat"):
        bt[k] = np.tile(bt[k][:nK], (B // nK + 1, 1, 1))[:B]
    hd.set_model(mpcqp.colmajor(bt["Ahat"]), mpcqp.colmajor(bt["Bhu"]), mpcqp.colmajor(bt["Chat"]))
hd.kf_set(mpcqp.colmajor(K), np.arange(ny))
dev = torch.device("cuda", 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
A, Bu, C = T(bt["Ahat"]), T(bt["Bhu"]), T(bt["Chat"])
xp = T(bt["xhat0"]).unsqueeze(2)                          # plant state
xh = T(bt["xhat0"] * 0.0)                                 # estimate starts at 0
lu, ry = T(bt["lastu0"]), T(bt["ry"])
Z = torch.zeros((B, hd.nZ), dtype=torch.float64, device=dev)
u0 = torch.empty((B, nu), dtype=torch.float64, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream(); sp = s.cuda_stream
gen = torch.Generator(device=dev); gen.manual_seed(0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t_kf = t_step = 0.0; iters = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(N):
    y = torch.bmm(C, xp).squeeze(2) + 0.02 * torch.randn((B, ny), dtype=torch.float64, device=dev, generator=gen)
    ev[0].record(s)
    hd.kf_correct_device(xh.data_ptr(), y.data_ptr(), stream=sp)
    ev[1].record(s)
    hd.step_device(xh.data_ptr(), lu.data_ptr(), ry.data_ptr(), Z.data_ptr(), u0.data_ptr(), st.data_ptr(), iters=it.data_ptr(), stream=sp)
    ev[2].record(s)
    hd.kf_predict_device(xh.data_ptr(), u0.data_ptr(), stream=sp)
    ev[3].record(s)
    lu.copy_(u0)                                          # getinput!: lastu0 <- u - uop
    xp = torch.bmm(A, xp) + torch.bmm(Bu, u0.unsqueeze(2))
    torch.cuda.synchronize()
    t_kf += ev[0].elapsed_time(ev[1]) + ev[2].elapsed_time(ev[3]); t_step += ev[1].elapsed_time(ev[2])
    iters.append(float(it.double().mean()))
    if len(sys.argv) > 4:
        print(f"period {k}: mean {iters[-1]:.2f} max {int(it.max())} not-optimal {int((st != 0).sum())} iters>40 {int((it > 40).sum())} step {ev[1].elapsed_time(ev[2]):.2f} ms", file=sys.stderr)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
kf_bytes = 8 * B * (nxh * nxh + nxh * nu + ny * nxh + nxh * ny + 4 * nxh + ny + nu)   # both kernels, per period
out = {"workload": cfg.name, "batch": B, "periods": N, "warm_dual": WARM_DUAL, "periods_per_s": N / wall, "controller_steps_per_s": B * N / wall,
       "ms_per_period": {"kalman_correct+predict": t_kf / N, "moveinput": t_step / N, "wall": wall / N * 1e3},
       "ipm_iters_first_last": [iters[0], iters[-1]], "all_optimal_last": bool((st == 0).all()),
       "kalman_roofline": {"bound": "hbm", "achieved": kf_bytes / (t_kf / N * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                           "frac": kf_bytes / (t_kf / N * 1e-3) / 1e9 / 8000.0, "bytes_per_period": kf_bytes}}
print(json.dumps(out))
