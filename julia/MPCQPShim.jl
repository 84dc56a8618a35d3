# MPCQPShim.jl -- the binding a ModelPredictiveControl.jl maintainer adds to run the `moveinput!`
# body of a BATCH of LinMPC controllers (identical dimensions) on MI355X through libmpcqp.so
# (C-ABI: include/mpcqp.h).  Everything else -- LinModel, estimators, LinMPC construction,
# setconstraint! bookkeeping, updatestate! -- stays the reference's Julia code.
#
# Julia is not part of the image this repository is built in, so this file is exercised by reading,
# not by CI: the same ABI is driven by modelpredictivecontrol.jl_amd/api.py (ctypes) and by
# tests/abi_c_client.c (plain C), whose call sequences this file mirrors one to one.
#
#   using ModelPredictiveControl, .MPCQPShim
#   mpcs  = [LinMPC(...) for _ in 1:B];  foreach(c -> setconstraint!(c; ...), mpcs)
#   batch = MPCQPShim.BatchLinMPC(mpcs)              # uploads models, weights, bounds; builds the kernel
#   u     = moveinput!(batch, ry)                    # (nu, B): one control period of all B controllers
module MPCQPShim

using ModelPredictiveControl
using LinearAlgebra: diag, Diagonal
import ModelPredictiveControl: moveinput!, LinMPC

const lib = get(ENV, "MPCQP_LIB", "libmpcqp.so")

const FLAG_RY_CONSTANT = 0x1          # Ry is (ny, B): ry held over Hp (the reference's default R̂y = repeat(ry, Hp))
const FLAG_COLD_START  = 0x2
const FLAG_WARM_DUAL   = 0x8
const STATUS_ITERATION_LIMIT, STATUS_ERROR = 1, 2

struct Dims                           # == mpcqp_dims
    batch::Cint; nxhat::Cint; nu::Cint; ny::Cint; nd::Cint; Hp::Cint; Hc::Cint
    nb::Ptr{Cint}; neps::Cint; device::Cint; flags::Cuint; max_iter::Cint
    gap_tol::Cdouble; res_tol::Cdouble; dual_reg::Cdouble
end

struct Bounds                         # == mpcqp_bounds: 16 pointers, C_NULL = group absent / default softness
    U0min::Ptr{Float64}; U0max::Ptr{Float64}; DUmin::Ptr{Float64}; DUmax::Ptr{Float64}
    Y0min::Ptr{Float64}; Y0max::Ptr{Float64}; x0min::Ptr{Float64}; x0max::Ptr{Float64}
    C_umin::Ptr{Float64}; C_umax::Ptr{Float64}; C_dumin::Ptr{Float64}; C_dumax::Ptr{Float64}
    C_ymin::Ptr{Float64}; C_ymax::Ptr{Float64}; c_x0min::Ptr{Float64}; c_x0max::Ptr{Float64}
end

mutable struct BatchLinMPC
    h::Ptr{Cvoid}
    mpcs::Vector{<:LinMPC}            # the logical controllers (same dimensions)
    Z̃::Matrix{Float64}                # (nZ̃, B) previous optima = warm start of the next period
    lastu0::Matrix{Float64}           # (nu, B)
    kernel::Int                       # MPCQP_KERNEL_* the steps run on
    flags::Cuint                      # MPCQP_FLAG_* given to the constructor (moveinput! only toggles RY_CONSTANT)
end

function check(rc::Integer)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:mpcqp_strerror, lib), Cstring, (Cint,), rc))
    rc == -6 && (msg *= ": " * unsafe_string(ccall((:mpcqp_last_hip_error, lib), Cstring, ())))
    error("mpcqp error $rc: $msg")
end

cat3(f, mpcs) = cat((f(c) for c in mpcs)...; dims=3)     # (rows, cols, B): the ABI layout, zero copy
cat2(f, mpcs) = reduce(hcat, (f(c) for c in mpcs))       # (n, B)
ptr_or_null(a) = a === nothing ? Ptr{Float64}(C_NULL) : pointer(a)

"""
    push_bounds!(h, mpcs)

`mpcqp_set_bounds` from the `ControllerConstraint` of every controller
(src/controller/construct.jl:126-199).  Bounds are the deviation vectors the reference stores
(`con.U0min = umin - Uop`, :359); `±Inf` = absent row (the `i_b` rule, transcription.jl:692-700).
Softness: the last column of the `A_*` blocks is `-C` (relaxU / relaxΔU, construct.jl:1029-1083);
`C_ymin, C_ymax, c_x̂min, c_x̂max` are stored as such.  A group whose bounds are all infinite for
every controller is passed as NULL (the kernel is specialised on the set of groups present).
"""
function push_bounds!(h::Ptr{Cvoid}, mpcs)
    soft = mpcs[1].nϵ == 1
    grp(f) = (a = cat2(c -> f(c.con), mpcs); all(isinf, a) ? nothing : a)
    sft(a, f) = (soft && a !== nothing) ? cat2(c -> f(c.con), mpcs) : nothing
    U0min, U0max = grp(c -> c.U0min), grp(c -> c.U0max)
    DUmin, DUmax = grp(c -> c.ΔUmin), grp(c -> c.ΔUmax)
    Y0min, Y0max = grp(c -> c.Y0min), grp(c -> c.Y0max)
    x0min, x0max = grp(c -> c.x̂0min), grp(c -> c.x̂0max)
    C_umin  = sft(U0min, c -> -Vector(c.A_Umin[:, end]));   C_umax  = sft(U0max, c -> -Vector(c.A_Umax[:, end]))
    C_dumin = sft(DUmin, c -> -Vector(c.A_ΔUmin[:, end]));  C_dumax = sft(DUmax, c -> -Vector(c.A_ΔUmax[:, end]))
    C_ymin  = sft(Y0min, c -> c.C_ymin);                    C_ymax  = sft(Y0max, c -> c.C_ymax)
    c_x0min = sft(x0min, c -> c.c_x̂min);                    c_x0max = sft(x0max, c -> c.c_x̂max)
    arrs = (U0min, U0max, DUmin, DUmax, Y0min, Y0max, x0min, x0max,
            C_umin, C_umax, C_dumin, C_dumax, C_ymin, C_ymax, c_x0min, c_x0max)
    GC.@preserve arrs begin
        b = Bounds(map(ptr_or_null, arrs)...)
        check(ccall((:mpcqp_set_bounds, lib), Cint, (Ptr{Cvoid}, Ref{Bounds}), h, b))
    end
    return nothing
end

"block-diagonal `M_Hp` (e.g. a terminal cost): blocks as (ny, ny, Hp, B), after mpcqp_set_weights"
function push_blockweight!(h::Ptr{Cvoid}, mpcs)
    ny, Hp = mpcs[1].estim.model.ny, mpcs[1].Hp
    Mb = Array{Float64}(undef, ny, ny, Hp, length(mpcs))
    for (b, c) in enumerate(mpcs), t in 1:Hp
        Mb[:, :, t, b] .= c.weights.M_Hp[(t-1)*ny+1:t*ny, (t-1)*ny+1:t*ny]
    end
    check(ccall((:mpcqp_set_output_weight_blocks, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, Mb))
end

isdiagonal(M) = all(iszero, M - Diagonal(diag(M)))

function BatchLinMPC(mpcs::Vector{<:LinMPC}; device::Integer=0, flags::Unsigned=FLAG_RY_CONSTANT)
    m = mpcs[1]; est = m.estim; model = est.model; B = length(mpcs)
    all(c -> (c.Hp, c.Hc, c.nϵ, c.nb) == (m.Hp, m.Hc, m.nϵ, m.nb), mpcs) ||
        throw(DimensionMismatch("all controllers of a batch need the same horizons"))
    nb = Cint.(m.nb)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve nb begin
        d = Dims(B, est.nx̂, model.nu, model.ny, model.nd, m.Hp, m.Hc, pointer(nb), m.nϵ, device,
                 flags, 0, 0.0, 0.0, 0.0)
        check(ccall((:mpcqp_create, lib), Cint, (Ref{Dims}, Ref{Ptr{Cvoid}}), d, h))
    end
    nd = model.nd
    Â, B̂u, Ĉ = cat3(c -> c.estim.Â, mpcs), cat3(c -> c.estim.B̂u, mpcs), cat3(c -> c.estim.Ĉ, mpcs)
    B̂d = nd > 0 ? cat3(c -> c.estim.B̂d, mpcs) : nothing
    D̂d = nd > 0 ? cat3(c -> c.estim.D̂d, mpcs) : nothing
    dop = cat2(c -> c.estim.f̂op - c.estim.x̂op, mpcs)
    GC.@preserve Â B̂u Ĉ B̂d D̂d dop check(ccall((:mpcqp_set_model, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        h[], Â, B̂u, Ĉ, ptr_or_null(B̂d), ptr_or_null(D̂d), dop))
    nΔU = model.nu * m.Hc
    Md = cat2(c -> diag(c.weights.M_Hp), mpcs)
    Nd = cat2(c -> diag(c.weights.Ñ_Hc)[1:nΔU], mpcs)
    Ld = cat2(c -> diag(c.weights.L_Hp), mpcs)
    Cw = Float64[m.nϵ == 1 ? c.weights.Ñ_Hc[end, end] : Inf for c in mpcs]
    GC.@preserve Md Nd Ld Cw check(ccall((:mpcqp_set_weights, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[], Md, Nd, Ld, Cw))
    isdiagonal(m.weights.M_Hp) || push_blockweight!(h[], mpcs)
    push_bounds!(h[], mpcs)
    # the specialised kernel of this shape and constraint pattern: built here (once per shape and
    # machine, cached), never inside a step -- the analogue of init_optimization! (linmpc.jl:303-339)
    # transcription of the controllers (LinMPC keyword, linmpc.jl:205-216): MultipleShooting runs on the stage-structured
    # kernel (include/mpcqp.h: mpcqp_set_transcription) when the handle qualifies, else the condensed kernels solve the
    # same problem
    if m.transcription isa MultipleShooting
        check(ccall((:mpcqp_set_transcription, lib), Cint, (Ptr{Cvoid}, Cint), h[], 1))
        if ccall((:mpcqp_transcription_supported, lib), Cint, (Ptr{Cvoid},), h[]) != 0
            @warn "MultipleShooting kernel not available for these controllers: the SingleShooting kernels solve the same problem"
            check(ccall((:mpcqp_set_transcription, lib), Cint, (Ptr{Cvoid}, Cint), h[], 0))
        end
    end
    kernel = ccall((:mpcqp_prepare, lib), Cint, (Ptr{Cvoid},), h[])
    kernel < 0 && check(kernel)
    b = BatchLinMPC(h[], mpcs, zeros(m.nϵ + nΔU, B), cat2(c -> c.lastu0, mpcs), kernel, Cuint(flags))
    finalizer(x -> ccall((:mpcqp_destroy, lib), Cint, (Ptr{Cvoid},), x.h), b)
    return b
end

"""
    moveinput!(b::BatchLinMPC, ry, d=...; R̂y=nothing, R̂u=nothing, D̂=nothing) -> u (nu, B)

Drop-in for `moveinput!` over the batch (src/controller/execute.jl:59-80): `ry` (ny, B), `d` (nd, B).
The estimates `x̂0` are read from the controllers' estimators (`preparestate!` stays Julia-side).
"""
function moveinput!(b::BatchLinMPC, ry::AbstractMatrix, d::AbstractMatrix=zeros(0, size(ry, 2));
                    R̂y=nothing, R̂u=nothing, D̂=nothing)
    m = b.mpcs[1]; model = m.estim.model; B = length(b.mpcs); Hp = m.Hp
    x̂0  = cat2(c -> c.estim.x̂0, b.mpcs)
    yop = cat2(c -> c.estim.model.yop, b.mpcs); uop = cat2(c -> c.estim.model.uop, b.mpcs)
    held = R̂y === nothing                                       # the held set point is sent once
    flags = held ? (b.flags | Cuint(FLAG_RY_CONSTANT)) : (b.flags & ~Cuint(FLAG_RY_CONSTANT))
    check(ccall((:mpcqp_set_flags, lib), Cint, (Ptr{Cvoid}, Cuint), b.h, flags))
    Ry0 = held ? Matrix{Float64}(ry .- yop) : Matrix{Float64}(R̂y .- repeat(yop, Hp))
    Ru0 = R̂u === nothing ? nothing : Matrix{Float64}(R̂u .- repeat(uop, Hp))
    d0 = D̂0 = nothing
    if model.nd > 0
        dop = cat2(c -> c.estim.model.dop, b.mpcs)
        d0 = Matrix{Float64}(d .- dop)
        D̂0 = D̂ === nothing ? repeat(d0, Hp) : Matrix{Float64}(D̂ .- repeat(dop, Hp))
    end
    u0 = similar(b.lastu0); status = zeros(Cint, B); iters = zeros(Cint, B)
    GC.@preserve x̂0 Ry0 Ru0 d0 D̂0 u0 status iters check(ccall((:mpcqp_step, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}),
        b.h, x̂0, b.lastu0, Ry0, ptr_or_null(Ru0), ptr_or_null(d0), ptr_or_null(D̂0), b.Z̃, u0, status, iters, C_NULL))
    any(==(STATUS_ERROR), status) && @error "MPC terminated without solution: returning last solution shifted"
    any(==(STATUS_ITERATION_LIMIT), status) && @warn "MPC termination status not OPTIMAL: keeping solution anyway"
    b.lastu0 .= u0                                              # getinput!: lastu0 <- u - uop
    for (i, c) in enumerate(b.mpcs)                             # keep the logical controllers in sync
        c.Z̃ .= @view b.Z̃[:, i]; c.lastu0 .= @view u0[:, i]
    end
    return u0 .+ uop
end

# ------------------------------------------------------------------------------------------------
# Batch of linear MovingHorizonEstimator objects (include/mpcqp_mhe.h; SURVEY 8 row f2).  The estimators keep
# their Julia fields; preparestate! / updatestate! of the BATCH run on the device.  Call sequence = the one of
# modelpredictivecontrol.jl_amd/mhe.py (BatchMHE), which the GPU tests drive.
import ModelPredictiveControl: preparestate!, updatestate!, MovingHorizonEstimator

struct MheDims                        # == mpcqp_mhe_dims
    batch::Cint; nxhat::Cint; nu::Cint; nym::Cint; nd::Cint; He::Cint; direct::Cint; device::Cint
    flags::Cuint; max_iter::Cint; gap_tol::Cdouble; res_tol::Cdouble; dual_reg::Cdouble
end

mutable struct BatchMHE
    h::Ptr{Cvoid}
    estims::Vector{MovingHorizonEstimator}
    x̂0::Matrix{Float64}              # (nx̂, B)
end

function BatchMHE(estims::Vector{<:MovingHorizonEstimator}; device::Integer=0)
    e = estims[1]; B = length(estims); model = e.model
    d = MheDims(B, e.nx̂, model.nu, e.nym, model.nd, e.He, e.direct ? 1 : 0, device, 0, 0, 0.0, 0.0, 0.0)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:mpcqp_mhe_create, lib), Cint, (Ref{MheDims}, Ref{Ptr{Cvoid}}), d, h))
    cat3(f) = cat((Matrix(f(c)) for c in estims)...; dims=3)
    cat2(f) = hcat((f(c) for c in estims)...)
    nd = model.nd
    check(ccall((:mpcqp_mhe_set_model, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
          Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[],
          cat3(c -> c.Â), cat3(c -> c.B̂u), cat3(c -> c.Ĉm), nd > 0 ? cat3(c -> c.B̂d) : C_NULL,
          nd > 0 ? cat3(c -> c.D̂dm) : C_NULL, cat2(c -> c.f̂op - c.x̂op), cat3(c -> c.cov.Q̂), cat3(c -> c.cov.R̂)))
    # bounds of setconstraint! as the reference keeps them: window-long vectors (con.x̂0min for the arrival state, then
    # con.X̂0min; con.Ŵmin; con.V̂min -- a per-channel keyword fills the whole vector, construct.jl:890-935), so the
    # window-long entry points are the faithful binding whatever mix of keywords the user called
    check(ccall((:mpcqp_mhe_set_bounds_window, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
          Ptr{Float64}, Ptr{Float64}), h[],
          cat2(c -> vcat(c.con.x̂0min, c.con.X̂0min)), cat2(c -> vcat(c.con.x̂0max, c.con.X̂0max)),
          cat2(c -> Vector(c.con.Ŵmin)), cat2(c -> Vector(c.con.Ŵmax)), cat2(c -> Vector(c.con.V̂min)), cat2(c -> Vector(c.con.V̂max))))
    if !isinf(e.C)         # soft constraints: Cwt and the softness of every row (first column of the A_* matrices = -C, construct.jl:964-1000)
        check(ccall((:mpcqp_mhe_set_softness_window, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
              Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[], [c.C for c in estims],
              cat2(c -> vcat(-Vector(c.con.A_x̂min[:, 1]), c.con.C_x̂min)), cat2(c -> vcat(-Vector(c.con.A_x̂max[:, 1]), c.con.C_x̂max)),
              cat2(c -> -Vector(c.con.A_Ŵmin[:, 1])), cat2(c -> -Vector(c.con.A_Ŵmax[:, 1])),
              cat2(c -> Vector(c.con.C_v̂min)), cat2(c -> Vector(c.con.C_v̂max))))
    end
    x̂0 = cat2(c -> c.x̂0)
    check(ccall((:mpcqp_mhe_init, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[],
          x̂0, cat3(c -> c.cov.P̂_0), nd > 0 ? cat2(c -> c.D0[1:nd]) : C_NULL, cat2(c -> c.lastu0)))
    return BatchMHE(h[], estims, x̂0)
end

function pull_state!(b::BatchMHE)
    check(ccall((:mpcqp_mhe_get, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), b.h, 0 #= MPCQP_MHE_XHAT0 =#, b.x̂0))
    for (i, c) in enumerate(b.estims); c.x̂0 .= @view b.x̂0[:, i]; end
    return b.x̂0 .+ hcat((c.x̂op for c in b.estims)...)
end

# preparestate!(estim, ym, d) over the batch (execute.jl:44-57): ym (nym, B), d (nd, B)
function preparestate!(b::BatchMHE, ym::AbstractMatrix, d::AbstractMatrix=zeros(0, size(ym, 2)))
    e = b.estims[1]
    y0m = ym .- hcat((c.model.yop[c.i_ym] for c in b.estims)...)
    d0 = d .- hcat((c.model.dop for c in b.estims)...)
    nbad = ccall((:mpcqp_mhe_prepare, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), b.h, y0m,
                 e.model.nd > 0 ? d0 : C_NULL)
    nbad < 0 && check(nbad)
    nbad > 0 && @warn "MHE termination status not OPTIMAL for $nbad estimators: keeping the open-loop estimate"
    return pull_state!(b)
end

# updatestate!(estim, u, ym, d) over the batch (execute.jl:76-88)
function updatestate!(b::BatchMHE, u::AbstractMatrix, ym::AbstractMatrix, d::AbstractMatrix=zeros(0, size(ym, 2)))
    e = b.estims[1]
    u0 = u .- hcat((c.model.uop for c in b.estims)...)
    y0m = ym .- hcat((c.model.yop[c.i_ym] for c in b.estims)...)
    d0 = d .- hcat((c.model.dop for c in b.estims)...)
    nbad = ccall((:mpcqp_mhe_update, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), b.h, u0, y0m,
                 e.model.nd > 0 ? d0 : C_NULL)
    nbad < 0 && check(nbad)
    return pull_state!(b)
end

end # module
