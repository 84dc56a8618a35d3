# bench/ref_julia.jl -- the REFERENCE's CPU path on the bench workload (SURVEY 8d-3): B independent
# LinMPC controllers built from the same inputs as bench.py (exported by scripts/export_batch.py,
# because the seeded generator is NumPy's), `moveinput!` of all of them under Threads.@threads,
# JuMP/OSQP as the reference ships it.  Prints ONE JSON line shaped like bench.py's `cpu_baseline`
# with "kind": "reference".  Run it only where Julia and the package are installed:
#
#   python scripts/export_batch.py C3 2048 /tmp/c3_batch        # writes /tmp/c3_batch.{json,bin}
#   julia -t auto bench/ref_julia.jl /tmp/c3_batch [seconds]
#
# The shape follows the reference's own benchmark (benchmark/3_bench_predictive_control.jl:6-28:
# construct, preparestate!, then time moveinput!); its OSQP time limit is lifted the same way (:187,192).
using ModelPredictiveControl, JuMP, LinearAlgebra, Printf
import JSON

prefix = ARGS[1]
budget = length(ARGS) > 1 ? parse(Float64, ARGS[2]) : 20.0
hdr = JSON.parsefile(prefix * ".json")
B, nx, nu, ny, Hp, Hc = hdr["B"], hdr["nx"], hdr["nu"], hdr["ny"], hdr["Hp"], hdr["Hc"]
raw = Vector{Float64}(undef, hdr["doubles"])
read!(prefix * ".bin", raw)
pos = Ref(0)
take(dims...) = (n = prod(dims); a = reshape(raw[pos[]+1:pos[]+n], dims...); pos[] += n; a)
# plant models (B of them, row-major (B, n, m) in the file -> permute to Julia's column-major)
A  = permutedims(take(nx, nx, B), (2, 1, 3)); Bu = permutedims(take(nu, nx, B), (2, 1, 3))
C  = permutedims(take(nx, ny, B), (2, 1, 3))
x0 = take(nx + ny, B); lastu = take(nu, B); ry = take(ny, B)

function controller(b)
    model = LinModel(A[:, :, b], Bu[:, :, b], C[:, :, b], zeros(nx, 0), zeros(ny, 0), 1.0)
    mpc = LinMPC(model; Hp, Hc, Mwt=fill(hdr["Mwt"], ny), Nwt=fill(hdr["Nwt"], nu), Lwt=fill(hdr["Lwt"], nu),
                 Cwt=something(hdr["Cwt"], Inf), nint_ym=ones(Int, ny))
    kw = Dict{Symbol,Any}()
    hdr["umin"] !== nothing && (kw[:umin] = fill(hdr["umin"], nu)); hdr["umax"] !== nothing && (kw[:umax] = fill(hdr["umax"], nu))
    hdr["dumin"] !== nothing && (kw[:Δumin] = fill(hdr["dumin"], nu)); hdr["dumax"] !== nothing && (kw[:Δumax] = fill(hdr["dumax"], nu))
    hdr["ymin"] !== nothing && (kw[:ymin] = fill(hdr["ymin"], ny)); hdr["ymax"] !== nothing && (kw[:ymax] = fill(hdr["ymax"], ny))
    setconstraint!(mpc; kw...)
    unset_time_limit_sec(mpc.optim)
    setstate!(mpc.estim, x0[:, b])                  # the augmented estimate x̂ = [x; integrators]
    initstate!(mpc, lastu[:, b], mpc.estim(zeros(0)))
    mpc.estim.x̂0 .= x0[:, b]                         # cold start from the bench's estimate
    return mpc
end

mpcs = [controller(b) for b in 1:B]
step!(b) = (mpcs[b].Z̃ .= 0; mpcs[b].lastu0 .= lastu[:, b]; moveinput!(mpcs[b], ry[:, b]))   # cold start, like bench.py
Threads.@threads for b in 1:B; step!(b); end                                                   # warm-up / compilation
passes, t0 = 0, time()
while time() - t0 < budget
    Threads.@threads for b in 1:B; step!(b); end
    global passes += 1
end
dt = time() - t0
@printf("{\"value\": %.1f, \"unit\": \"solves/s\", \"cores\": %d, \"kind\": \"reference\", \"sample\": \"%d controllers x %d cold-start passes of moveinput! (JuMP/OSQP, reference defaults), %.1f s\"}\n",
        B * passes / dt, Threads.nthreads(), B, passes, dt)
