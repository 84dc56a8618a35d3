// mhe_bodies.h -- kernel bodies of the batched linear MovingHorizonEstimator, written
// against a wave interface W (gfx950: mhe_devwave.h; CPU emulator: tests/emu) like mpcqp_bodies.h.
//
// What the reference does per period (LinModel, SingleShooting; one estimator):
//   add_data_windows!        src/estimator/mhe/execute.jl:497-548   -> push_data()
//   initpred! (F, fx̄, H̃, q̃)  src/estimator/mhe/execute.jl:419-457   -> build_q()        (stage form, see below)
//   linconstraint! (b)       src/estimator/mhe/transcription.jl:732-782 -> the row residuals of the sweeps
//   optim_objective!         src/estimator/mhe/execute.jl:576-618   -> interior-point iterations
//   getstate!                src/estimator/mhe/execute.jl:629-643   -> write_outputs()
//   correct_cov!/update_cov! src/estimator/mhe/execute.jl:727-781   -> cov_body()       (KalmanFilter recursion,
//                            src/estimator/kalman.jl:1235-1264, 1275-1290)
//
// Stage form.  With states x(0..N) (x(0) = x̂0arr, N = Nk), g(j) = B̂u u0(j) + B̂d d0(j+p) + (f̂op - x̂op),
// e(i) = y0m(i) - D̂dm d0(i+1):   ŵ(j) = x(j+1) - Â x(j) - g(j),   v̂(i) = e(i) - Ĉm x(i+1-p)   and
//   J = (x(0)-x̄)' P̄⁻¹ (x(0)-x̄) + Σ ŵ' Q̂⁻¹ ŵ + Σ v̂' R̂⁻¹ v̂          (execute.jl:444-453 is its condensed form)
// so the Hessian in X is block tridiagonal:  diag blocks  [s=0] 2P̄⁻¹ + T1 (+T3 if p=1) | T1+T2+T3 | [s=N] T2 (+T3 if p=0),
// sub-diagonal blocks Oc = -2 Q̂⁻¹ Â,  T1 = 2 Â'Q̂⁻¹Â, T2 = 2 Q̂⁻¹, T3 = 2 Ĉm'R̂⁻¹Ĉm.
// Bounds (setconstraint!, construct.jl:858-1049, per channel): x̂min <= x(s) <= x̂max, ŵmin <= ŵ(j) <= ŵmax,
// v̂min <= v̂(i) <= v̂max; every row touches one stage (x̂, v̂) or two neighbours (ŵ), so
// Φ = H + G'D̃G keeps the block-tridiagonal pattern.  Newton systems are solved by the block Thomas
// recursion  S(s) = Φ(s,s) - O(s-1) S(s-1)⁻¹ O(s-1)',  Si(s) = S(s)⁻¹ (Gauss-Jordan in registers).
//
// Interior-point method: the same dual-regularised Mehrotra predictor-corrector as the LinMPC step
// (mpcqp_bodies.h Step::run; oracle/linmpc_ref.c restates it), without the active-set polish.
#pragma once
#include <math.h>

#include <type_traits>

#include "mhe_types.h"

// prefetch distance (stages) of the sweeps' scratch loads: factorisation sweep / solve sweeps / update pass
#ifndef MPCQP_MHE_DEPTH_F0
#define MPCQP_MHE_DEPTH_F0 1
#endif
#ifndef MPCQP_MHE_DEPTH
#define MPCQP_MHE_DEPTH 1
#endif
#ifndef MPCQP_MHE_DEPTH_U
#define MPCQP_MHE_DEPTH_U 2
#endif
#ifndef MPCQP_MHE_EXACT_PIVOT
#define MPCQP_MHE_EXACT_PIVOT 0
#endif
#ifndef MPCQP_MHE_GJ_DEFER
#define MPCQP_MHE_GJ_DEFER 1     // Gauss-Jordan inverse with the row scalings deferred to the end (Ops::gj)
#endif
#ifndef MPCQP_MHE_BMID_LDS
#define MPCQP_MHE_BMID_LDS 1     // middle diagonal block of the Hessian in LDS (18 KB per wave) or read from the constant block (12 KB)
#endif

namespace mpcqp {
namespace mhe {

// 1/x: v_rcp_f64 + one Newton step (2e-15 relative, see rcp() in mpcqp_bodies.h) / the raw instruction (5e-8) on the
// GPU; a plain division on the host (CPU wave emulator).
MPCQP_HD inline double recip(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
#else
    return 1.0 / x;
#endif
}
MPCQP_HD inline double recip_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    return 1.0 / x;
#endif
}

template <int I>
struct IC { static constexpr int v = I; };
template <int N, int I = 0, class F>
MPCQP_HD void sfor(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        sfor<N, I + 1>(f);
    }
}

// Row-lane linear algebra of one 16-lane group: every lane holds ITS row of each operand.
template <class W, int NX>
struct Ops {
    W& w;
    using Row = double[NX];

    // w.fmabc4<L0,L1,L2,L3>(acc, x0..x3, y0..y3): acc += sum_i (x_i of lane L_i of this row) * y_i -- four
    // v_fmac_f64_dpp on gfx950 (the row broadcast is the DPP modifier of the multiply-add itself)
    MPCQP_HD double mv(const Row& M, double v) const {            // (M v)[r]
        double acc = 0.0;
        sfor<NX / 4>([&](auto ij) {
            constexpr int c = 4 * decltype(ij)::v;
            w.template fmabc4<c, c + 1, c + 2, c + 3>(acc, v, v, v, v, M[c], M[c + 1], M[c + 2], M[c + 3]);
        });
        return acc;
    }
    MPCQP_HD void mm(const Row& X, const Row& Y, Row& C) const {  // C = X Y
        sfor<NX>([&](auto ic) {
            constexpr int c = decltype(ic)::v;
            double acc = 0.0;
            sfor<NX / 4>([&](auto ij) {
                constexpr int k = 4 * decltype(ij)::v;
                w.template fmabc4<k, k + 1, k + 2, k + 3>(acc, Y[c], Y[c], Y[c], Y[c], X[k], X[k + 1], X[k + 2], X[k + 3]);
            });
            C[c] = acc;
        });
    }
    MPCQP_HD void mmt_sub(const Row& X, const Row& Y, Row& C) const {   // C -= X Y', accumulated in place
        sfor<NX>([&](auto ic) {
            constexpr int c = decltype(ic)::v;
            sfor<NX / 4>([&](auto ij) {
                constexpr int k = 4 * decltype(ij)::v;
                w.template fmsbc4<c, c, c, c>(C[c], Y[k], Y[k + 1], Y[k + 2], Y[k + 3], X[k], X[k + 1], X[k + 2], X[k + 3]);
            });
        });
    }
    MPCQP_HD void mmt_acc(const Row& X, const Row& Y, Row& C, double sign) const {   // C += sign X Y'
        sfor<NX>([&](auto ic) {
            constexpr int c = decltype(ic)::v;
            double acc = 0.0;
            sfor<NX / 4>([&](auto ij) {
                constexpr int k = 4 * decltype(ij)::v;
                w.template fmabc4<c, c, c, c>(acc, Y[k], Y[k + 1], Y[k + 2], Y[k + 3], X[k], X[k + 1], X[k + 2], X[k + 3]);
            });
            C[c] = fma(sign, acc, C[c]);
        });
    }
    // in-place inverse of a symmetric positive definite matrix (Gauss-Jordan, no pivoting: the pivots
    // are those of the LDL' factorisation).  Returns false (for the whole group) on a bad pivot.
    // MPCQP_MHE_GJ_DEFER: the pivot row is NOT divided by its pivot when it is eliminated with -- a row scaling commutes
    // with the later eliminations (they see d_k times the scaled row and form d_k times its update) -- so the step is one
    // in-place multiply-add per entry (the pivot lane with factor 0) instead of a multiply and a multiply-add, and every
    // lane scales its row by its own 1/d once at the end: NX^2 + NX instead of 2 NX^2 FP64 instructions per inverse.
    MPCQP_HD bool gj(Row& a, int r) const {
        bool ok = true;
#if MPCQP_MHE_GJ_DEFER
        double mypinv = 0.0;
        sfor<NX>([&](auto ik) {
            constexpr int k = decltype(ik)::v;
            const double dk = w.template rowbc<k>(a[k]);      // (gjacc4 ends with the wait states a DPP read of its results needs)
            ok = ok && (dk > 1e-280) && (dk < 1e280);
#if MPCQP_MHE_EXACT_PIVOT
            const double pinv = 1.0 / dk;
#else
            const double pinv = recip(dk);
#endif
            // one = 1 on the pivot lane, 0 elsewhere (a select of the high word only); the three lane-dependent values follow
            // by arithmetic: g = x - one x (exactly 0 on the pivot lane), the lane's own 1/d, the new column k
            const double one = (r == k) ? 1.0 : 0.0;
            const double x = -a[k] * pinv;                   // other rows: a[c] - a[k] a_k[c] / dk
            const double g = fma(-one, x, x);
            mypinv = fma(one, pinv, mypinv);
            // (column k is computed too and then replaced: the four-element groups stay uniform)
            sfor<NX / 4>([&](auto ij) {
                constexpr int c = 4 * decltype(ij)::v;
                w.template gjacc4<k>(a[c], a[c + 1], a[c + 2], a[c + 3], g);
            });
            a[k] = g + one;
        });
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; a[c] *= mypinv; });
#else
        sfor<NX>([&](auto ik) {
            constexpr int k = decltype(ik)::v;
            const double dk = w.template rowbc<k>(a[k]);
            ok = ok && (dk > 1e-280) && (dk < 1e280);
#if MPCQP_MHE_EXACT_PIVOT
            const double pinv = 1.0 / dk;
#else
            const double pinv = recip(dk);
#endif
            const bool piv = (r == k);
            // a[c] <- m a[c] + g (a[c] of lane k):  pivot row (m = 0, g = 1/dk): a[c]/dk;  other rows (m = 1,
            // g = -a[k]/dk): a[c] - a[k] a_k[c] / dk
            const double g = piv ? pinv : -a[k] * pinv;
            const double m = piv ? 0.0 : 1.0;
            // (column k is computed too and then replaced: the four-element groups stay uniform)
            sfor<NX / 4>([&](auto ij) {
                constexpr int c = 4 * decltype(ij)::v;
                w.template gjrow4<k>(a[c], a[c + 1], a[c + 2], a[c + 3], m, g);
            });
            a[k] = g;
        });
#endif
        return ok;
    }
    MPCQP_HD static void ld(const double* p, int stride, Row& M) {
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; M[c] = p[(size_t)c * stride]; });
    }
    MPCQP_HD static void st(double* p, int stride, const Row& M) {
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; p[(size_t)c * stride] = M[c]; });
    }
    // element `idx` of an array with a wave-uniform base: the byte offset is formed in 32 bits, which lets the
    // access use the SGPR-base + 32-bit-VGPR-offset form of global_load / global_store (arrays < 4 GB)
    template <class T>
    MPCQP_HD static T* at(T* base, int idx) {
        return (T*)((char*)const_cast<typename std::remove_const<T>::type*>(base) + (uint32_t)((uint32_t)idx * (uint32_t)sizeof(T)));
    }
    // the same with a wave-uniform base pointer and a 32-bit per-lane element offset: the address is formed
    // at the access (SGPR base + VGPR offset) instead of living in a 64-bit register pair per array
    MPCQP_HD static void ldo(const double* base, int off, int stride, Row& M) {
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; M[c] = at(base, off)[c * stride]; });
    }
    MPCQP_HD static void sto(double* base, int off, int stride, const Row& M) {
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; at(base, off)[c * stride] = M[c]; });
    }
    MPCQP_HD static void add_diag(Row& M, int r, double v) {
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; M[c] += (r == c) ? v : 0.0; });
    }
};

// ---------------------------------------------------------------------------------------------
// set_model: the constant block of every estimator (once per model / covariance change)
template <class W, int NX>
MPCQP_HD void setup_body(W& w, const Dims& d, const Raw& in, double* cst_all, int wave_id) {
    using O = Ops<W, NX>;
    typename O::Row A, At, Qi, Cm, Ct, Ri, T, U;
    O op{w};
    const int lane = w.lane, r = lane & (RL - 1), g = lane >> 4;
    const CstMap cm = cst_map(NX, d.nu, d.nd);
    const int nx = d.nx, nym = d.nym, nu = d.nu, nd = d.nd;
    for (int wg = wave_id; wg * GPW < d.B; wg += d.nwaves) {
        const int bq = wg * GPW + g;
        const bool live = bq < d.B;
        const int b = live ? bq : d.B - 1;
        double* cst = cst_all + (size_t)b * cm.stride + r;
        const double* Ar = in.Ahat + (size_t)b * nx * nx;
        const double* Qr = in.Q + (size_t)b * nx * nx;
        const double* Cr = in.Cm + (size_t)b * nym * nx;
        const double* Rr = in.R + (size_t)b * nym * nym;
        sfor<NX>([&](auto ic) {
            constexpr int c = decltype(ic)::v;
            const bool in_x = r < nx && c < nx;
            A[c] = in_x ? Ar[c * nx + r] : 0.0;
            At[c] = in_x ? Ar[r * nx + c] : 0.0;
            Qi[c] = in_x ? Qr[c * nx + r] : (r == c ? 1.0 : 0.0);
            Cm[c] = (r < nym && c < nx) ? Cr[c * nym + r] : 0.0;          // row = measured output r
            Ct[c] = (r < nx && c < nym) ? Cr[r * nym + c] : 0.0;          // row = state r, column = output c
            Ri[c] = (r < nym && c < nym) ? Rr[c * nym + r] : (r == c ? 1.0 : 0.0);
        });
        if (live) {
            O::st(cst + cm.A, RL, A); O::st(cst + cm.At, RL, At);
            O::st(cst + cm.Cm, RL, Cm); O::st(cst + cm.Ct, RL, Ct);
            O::st(cst + cm.Q, RL, Qi); O::st(cst + cm.R, RL, Ri);
        }
        op.gj(Qi, r);
        op.gj(Ri, r);
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Qi[c] *= 2.0; Ri[c] *= 2.0; });    // 2 Q̂⁻¹, 2 R̂⁻¹
        typename O::Row Bm;
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Bm[c] = Qi[c]; });
        if (live) O::st(cst + cm.T2, RL, Qi);
        op.mm(Qi, A, T);                                   // 2 Q̂⁻¹ Â
        op.mm(At, T, U);                                   // T1 = Â' 2Q̂⁻¹ Â
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Bm[c] += U[c]; T[c] = -T[c]; });
        if (live) { O::st(cst + cm.T1, RL, U); O::st(cst + cm.Oc, RL, T); }
        op.mm(At, Qi, U);                                  // (Â' 2Q̂⁻¹) = -Oc'
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; U[c] = -U[c]; });
        if (live) O::st(cst + cm.OcT, RL, U);
        op.mm(Ri, Cm, T);                                  // 2R̂⁻¹ Ĉm
        op.mm(Ct, T, U);                                   // T3
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Bm[c] += U[c]; });
        if (live) { O::st(cst + cm.T3, RL, U); O::st(cst + cm.Bmid, RL, Bm); }
        op.mm(Ct, Ri, U);                                  // CR = Ĉm' 2R̂⁻¹
        if (live) O::st(cst + cm.CR, RL, U);
        if (live) {
            for (int c = 0; c < nu; ++c) cst[cm.Bu + c * RL] = r < nx ? in.Bu[(size_t)b * nx * nu + c * nx + r] : 0.0;
            for (int c = 0; c < nd; ++c) {
                cst[cm.Bd + c * RL] = r < nx ? in.Bd[(size_t)b * nx * nd + c * nx + r] : 0.0;
                cst[cm.Ddm + c * RL] = r < nym ? in.Ddm[(size_t)b * nym * nd + c * nym + r] : 0.0;
            }
            cst[cm.fx] = (r < nx && in.fx) ? in.fx[(size_t)b * nx + r] : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// arrival covariance: mode bit 0 = KalmanFilter correction, bit 1 = prediction (data independent),
// then Pi2 = 2 P̄⁻¹.  P row-lane [B][NX*RL]; mode 4: (re)load P from the ABI array P0 [B][nx*nx] first.
template <class W, int NX>
MPCQP_HD void cov_body(W& w, const Dims& d, const Args& a, int mode, const double* P0, double* Pout, int wave_id) {
    using O = Ops<W, NX>;
    typename O::Row P, A, Cm, X, Y, M;
    O op{w};
    const int lane = w.lane, r = lane & (RL - 1), g = lane >> 4;
    const CstMap cm = cst_map(NX, d.nu, d.nd);
    const int nx = d.nx;
    for (int wg = wave_id; wg * GPW < d.B; wg += d.nwaves) {
        const int bq = wg * GPW + g;
        const bool live = bq < d.B;
        const int b = live ? bq : d.B - 1;
        const double* cst = a.cst + (size_t)b * cm.stride + r;
        double* Pm = a.P + (size_t)b * NX * RL + r;
        bool good = true;
        if (mode & 4) {
            sfor<NX>([&](auto ic) {
                constexpr int c = decltype(ic)::v;
                P[c] = (r < nx && c < nx) ? P0[(size_t)b * nx * nx + c * nx + r] : (r == c ? 1.0 : 0.0);
            });
        } else {
            O::ld(Pm, RL, P);
        }
        if (mode & 1) {        // P <- P - P Ĉm' (Ĉm P Ĉm' + R̂)⁻¹ Ĉm P
            O::ld(cst + cm.Cm, RL, Cm);
            O::ld(cst + cm.R, RL, M);
            op.mm(Cm, P, X);                   // Ĉm P          (row = output)
            op.mmt_acc(X, Cm, M, 1.0);         // M = R̂ + (Ĉm P) Ĉm'
            good = op.gj(M, r) && good;
            op.mm(M, X, Y);                    // M⁻¹ Ĉm P
            sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; M[c] = 0.0; });
            op.mmt_acc(P, Cm, M, 1.0);         // P Ĉm'         (row = state, column = output)
            op.mm(M, Y, X);
            sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; P[c] -= X[c]; });
        }
        if (mode & 2) {        // P <- Â P Â' + Q̂
            O::ld(cst + cm.A, RL, A);
            op.mm(A, P, X);
            O::ld(cst + cm.Q, RL, P);
            // (padding rows/columns of Q̂ hold the identity: the padded block of P stays the identity)
            op.mmt_acc(X, A, P, 1.0);
        }
        // correct_cov! / update_cov! (mhe/execute.jl:729-780): a new P̄ that is not finite or not positive definite (or
        // not invertible) is dropped -- P̄ and 2 P̄⁻¹ keep their previous values.  (mode 4 loads the caller's P̂_0.)
        double fin = 1.0;
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; fin = (P[c] - P[c] == 0.0) ? fin : 0.0; });
        good = good && w.rmin(fin) > 0.5;
        typename O::Row Pi;
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Pi[c] = P[c]; });
        good = op.gj(Pi, r) && good;
        const bool commit = good || (mode & 4) || mode == 0;        // (mode 0 only reads P̄ back)
        if (live && commit) {
            O::st(Pm, RL, P);
            if (Pout && r < nx)
                sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; if (c < nx) Pout[(size_t)b * nx * nx + c * nx + r] = P[c]; });
            sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Pi[c] *= 2.0; });
            O::st(a.Pi2 + (size_t)b * NX * RL + r, RL, Pi);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// one inequality row: slack sv, multiplier lv, primal residual rp = g'z + s - h.
struct RowK {
    double Dt, c;
};
// Everything a row needs is ONE reciprocal, wi = 1/(s + δλ) (the row algebra of Step::row_step in mpcqp_bodies.h):
//   D̃ = λ wi,   w/s = wi,   c = wi rc - D̃ rp,   dλ = wi (λ a - rc),   ds = -wi (s a + δ rc),   a = rp + g'dz
// -- v_rcp_f64 + one Newton step (recip) instead of the three / four IEEE divisions (~12 instructions each) of the
// textbook formulas; algebraically the same quantities.
MPCQP_HD inline RowK row_rhs(bool has, double sv, double lv, double rp, double extra, double delta) {
    // D̃ = D/(1 + δD), c = w rc/s - D̃ rp with rc = s λ + extra (Step::run of mpcqp_bodies.h)
    RowK k;
    const double wi = recip(fma(delta, lv, sv));
    k.Dt = has ? lv * wi : 0.0;
    const double rc = fma(sv, lv, extra);
    k.c = has ? fma(wi, rc, -k.Dt * rp) : 0.0;
    return k;
}
MPCQP_HD inline void row_dir(bool has, double sv, double lv, double rp, double gd, double extra, double delta,
                             double& ds, double& dl) {
    const double wi = recip(fma(delta, lv, sv)), rc = fma(sv, lv, extra), a = rp + gd;
    dl = has ? wi * fma(lv, a, -rc) : 0.0;
    ds = has ? -wi * fma(sv, a, delta * rc) : 0.0;
}
// step to the boundary along dv < 0 (raw v_rcp_f64, 5e-8 relative: the fraction-to-the-boundary factor leaves 1e-4)
MPCQP_HD inline double ratio(double v, double dv) { return dv < 0.0 ? -v * recip_fast(dv) : 1e300; }

// ---------------------------------------------------------------------------------------------
// CM: compile-time set of bound classes the code is generated for (a handle whose classes are a subset runs it)
template <class W, int NX, unsigned CM = 7u>
struct Solver {
    using O = Ops<W, NX>;
    using Row = typename O::Row;
    W& w;
    const Dims& d;
    const Args& a;
    O op;
    const int lane, r, g;
    const CstMap cm;
    const SlotMap sm;
    int b;
    bool live;
    const double* cbase;   // constant blocks of this wavefront's first estimator (wave-uniform)
    int coff;           // this lane's offset into them: (b - first) * stride + r
    double* sb;      // scratch of this wavefront (wave-uniform base)
    // A slot holds one double per ACTIVE lane: the NX lanes of each of the four groups, packed (SW = 4 NX doubles;
    // 384 B instead of 512 B for NX = 12 -- a quarter of the HBM traffic of the sweeps would be padding).
    static constexpr int SW = GPW * NX;
    typename W::Buf sbuf;     // buffer resource over this wavefront's scratch
    unsigned lvo;             // this lane's byte offset inside a slot; out of range for the idle lanes (r >= NX)
    int first;       // first estimator of the wavefront's current group of four
    double* lds;     // this lane's column of the wave's LDS: Oc, OcT, Bmid rows
    int N, p;
    bool cX, cW, cV;
    double xlo, xhi, wlo, whi, vlo, vhi;
    bool hxlo, hxhi, hwlo, hwhi, hvlo, hvhi;
    bool cS;                                   // a slack variable ε >= 0 relaxes the rows with softness c > 0
    bool cL;                                   // window-long bounds (CLS_L): xlo .. vhi are re-read for every stage
    bool cC;                                   // window-long softness (CLS_C): cx0 .. cv1 are re-read for every stage
    double cx0, cx1, cw0, cw1, cv0, cv1;       // softness of this lane's rows (0: hard)

    MPCQP_HD Solver(W& w_, const Dims& d_, const Args& a_, double* smem, int wave_id)
        : w(w_), d(d_), a(a_), op{w_}, lane(w_.lane), r(w_.lane & (RL - 1)), g(w_.lane >> 4),
          cm(cst_map(NX, d_.nu, d_.nd)), sm(slot_map(NX, d_.He, d_.cls)) {
        sb = a.scratch + (size_t)wave_id * wave_scratch_doubles(NX, d.nslot);
        sbuf = w.make_buf(sb, wave_scratch_doubles(NX, d.nslot) * sizeof(double));
        lvo = r < NX ? (unsigned)((g * NX + r) * 8) : W::BUF_OOB;
        lds = smem + lane;
        N = d.N;
        p = d.direct ? 0 : 1;
        cX = (CM & CLS_X) && (d.cls & CLS_X); cW = (CM & CLS_W) && (d.cls & CLS_W); cV = (CM & CLS_V) && (d.cls & CLS_V);
        cS = (CM & CLS_S) && (d.cls & CLS_S);
        cL = (CM & (CLS_W | CLS_V)) && (d.cls & CLS_L);      // (served by the all-class variants only)
        cC = cS && (d.cls & CLS_C);
    }
    // Scratch accesses are raw buffer loads / stores: the slot offset is the instruction's scalar offset, the lane's
    // byte offset ONE register for the whole kernel, and the idle lanes carry an out-of-range offset -- the
    // hardware returns zero for their loads and drops their stores (no exec masking, no trash area).
    MPCQP_HD double Sld(int slot) { return w.bload(sbuf, lvo, slot * (SW * 8)); }
    MPCQP_HD void Sst(int slot, double v) { w.bstore(sbuf, lvo, slot * (SW * 8), v); }
    MPCQP_HD void Sld_rows(int slot, Row& M) {
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; M[c] = w.bload(sbuf, lvo, (slot + c) * (SW * 8)); });
    }
    MPCQP_HD void Sst_rows(int slot, const Row& M) {
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; w.bstore(sbuf, lvo, (slot + c) * (SW * 8), M[c]); });
    }
    MPCQP_HD const double* L_Oc() const { return lds; }
    MPCQP_HD const double* L_OcT() const { return lds + (size_t)NX * WAVE; }
    MPCQP_HD const double* L_Bmid() const { return lds + (size_t)2 * NX * WAVE; }
    MPCQP_HD int yslot(int i) const { return (d.hy + i) % d.He; }
    MPCQP_HD int dslot(int i) const { return (d.hd + i) % (d.He + 1); }
    // measurement attached to state s (p = 0: i = s-1, p = 1: i = s), -1: none
    MPCQP_HD int meas_of(int s) const { const int i = s - 1 + p; return (i >= 0 && i < N) ? i : -1; }

    // Window-long bounds (setconstraint!(estim; X̂min, ..., V̂max), construct.jl:858-935): the rows of stage s -- state
    // x(s) (s = 0: the arrival state's own bound), ŵ(s), v̂ of the measurement attached to s -- take their bounds from
    // block (He - N + .) of the He-long vectors: a window of N < He periods uses the LAST N blocks (trunc_bounds).
    MPCQP_HD void stage_bounds(int s) {
        if (cC) stage_softness(s);
        if (!cL) return;
        const int He = d.He, blk = He - N;
        auto at = [&](const double* p_, int nblk, int j, double dflt) {
            return p_ ? p_[((size_t)b * nblk + j) * RL + r] : dflt;
        };
        if (cX) {
            const int j = s == 0 ? 0 : 1 + blk + (s - 1);
            xlo = at(a.xmin, He + 1, j, -BIG); xhi = at(a.xmax, He + 1, j, BIG);
            hxlo = xlo > -BIG; hxhi = xhi < BIG;
        }
        if (cW && s < N) {
            wlo = at(a.wmin, He, blk + s, -BIG); whi = at(a.wmax, He, blk + s, BIG);
            hwlo = wlo > -BIG; hwhi = whi < BIG;
        }
        const int im = meas_of(s);
        if (cV && im >= 0) {
            vlo = at(a.vmin, He, blk + im, -BIG); vhi = at(a.vmax, He, blk + im, BIG);
            hvlo = vlo > -BIG; hvhi = vhi < BIG;
        }
    }

    // Window-long softness (setconstraint!(estim; C_x̂min, ..., C_v̂max), construct.jl:937-1020).  The softness is a column of
    // the constraint matrices, which the reference does NOT truncate while the window grows (only the bound vectors go
    // through trunc_bounds, transcription.jl:750-752): the rows of window entry j take softness block j, whatever Nk.
    MPCQP_HD void stage_softness(int s) {
        const int He = d.He;
        auto at = [&](const double* p_, int nblk, int j) { return p_ ? p_[((size_t)b * nblk + j) * RL + r] : 0.0; };
        if (cX) { cx0 = at(a.cxmin, He + 1, s); cx1 = at(a.cxmax, He + 1, s); }        // block 0: the arrival state
        if (cW && s < N) { cw0 = at(a.cwmin, He, s); cw1 = at(a.cwmax, He, s); }
        const int im = meas_of(s);
        if (cV && im >= 0) { cv0 = at(a.cvmin, He, im); cv1 = at(a.cvmax, He, im); }
    }

    // O(j) = Oc - D̃w(j) Â: sub-diagonal block (j+1, j) of the Newton matrix (D̃w(j) from the forward sweep of phase 0)
    MPCQP_HD void load_O(int j, Row& Ob) {
        O::ld(L_Oc(), WAVE, Ob);
        if (cW) {
            const double Dp = Sld(sm.WD + j);
            Row Ap;
            O::ldo(w.uniform(cbase + cm.A), coff, RL, Ap);
            sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Ob[c] -= Dp * Ap[c]; });
        }
    }

    // diagonal block of the Hessian at stage s
    MPCQP_HD void base_block(int s, Row& Bs) {
        if (s > 0 && s < N) {
#if MPCQP_MHE_BMID_LDS
            O::ld(L_Bmid(), WAVE, Bs);
#else
            O::ldo(w.uniform(cbase + cm.Bmid), coff, RL, Bs);      // once per stage and iteration: not worth LDS
#endif
            return;
        }
        Row T;
        if (s == 0) {
            O::ldo(a.Pi2 + (size_t)first * NX * RL, (b - first) * NX * RL + r, RL, Bs);
            O::ldo(w.uniform(cbase + cm.T1), coff, RL, T);
        } else {
            sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Bs[c] = 0.0; });
            O::ldo(w.uniform(cbase + cm.T2), coff, RL, T);
        }
        sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Bs[c] += T[c]; });
        if (meas_of(s) >= 0) {
            O::ldo(w.uniform(cbase + cm.T3), coff, RL, T);
            sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Bs[c] += T[c]; });
        }
    }

    // add_data_windows! (execute.jl:497-548): this period's data become window entry N-1 (ring slots)
    MPCQP_HD void push_data() {
        if (!a.y0m_new) return;
        const int nx = d.nx, nu = d.nu, nd = d.nd, nym = d.nym, He = d.He;
        if (live) {
            const int e = yslot(N - 1);
            if (r < nym) a.Y0m[((size_t)b * He + e) * nym + r] = a.y0m_new[(size_t)b * nym + r];
            if (r < nx) a.X0old[((size_t)b * He + e) * nx + r] = a.xhat0[(size_t)b * nx + r];
            for (int c = r; c < nu; c += RL) a.U0[((size_t)b * He + e) * nu + c] = a.u0_new[(size_t)b * nu + c];
            const int ed = dslot(N);
            for (int c = r; c < nd; c += RL) a.D0[((size_t)b * (He + 1) + ed) * nd + c] = a.d0_new[(size_t)b * nd + c];
        }
        w.sync();
    }

    MPCQP_HD double g_of(int j) {      // g(j)[r] = B̂u u0(j) + B̂d d0(j+p) + (f̂op - x̂op)
        double acc = *O::at(cbase + cm.fx, coff);
        const double* u = a.U0 + ((size_t)b * d.He + yslot(j)) * d.nu;
        for (int c = 0; c < d.nu; ++c) acc = fma(O::at(cbase + cm.Bu, coff)[c * RL], u[c], acc);
        const double* dd = a.D0 + ((size_t)b * (d.He + 1) + dslot(j + p)) * d.nd;
        for (int c = 0; c < d.nd; ++c) acc = fma(O::at(cbase + cm.Bd, coff)[c * RL], dd[c], acc);
        return acc;
    }
    MPCQP_HD double e_of(int i) {      // e(i)[r] = y0m(i) - D̂dm d0(i+1)        (lanes r < nym)
        double acc = r < d.nym ? a.Y0m[((size_t)b * d.He + yslot(i)) * d.nym + r] : 0.0;
        const double* dd = a.D0 + ((size_t)b * (d.He + 1) + dslot(i + 1)) * d.nd;
        for (int c = 0; c < d.nd; ++c) acc = fma(-O::at(cbase + cm.Ddm, coff)[c * RL], dd[c], acc);
        return acc;
    }

    // ---- everything one stage of a sweep reads from the scratch.  The sweeps load it DEPTH stages ahead of its
    // use (software pipeline in registers): a stage of the light sweeps computes for ~0.3 us, a scratch access
    // (HBM / MALL, the working set of the resident wavefronts is ~0.5 GB) takes a few us.
    struct RowIn { double s0, l0, s1, l1; };
    struct StageIn {
        double x;        // forward: X(s+1); backward, update: X(s)
        double a0;       // F0: Q(s), F1: RD(s), B0/B1: T(s), U: DX(s)
        double dxa;      // DXA(s)
        RowIn xr, wr, vr;
        double ww, wga, wg, wd;      // ŵ(s), affine / final direction of the ŵ rows, D̃w(s)
        double vv, vga, vg;          // v̂(im), directions of the v̂ rows
        double gs, es;               // g(s), e(im)
        double phi, tp, psi;         // slack column φ(s) of the Newton matrix, its forward solve, ψ(s) = (Φxx⁻¹φ)(s)
        Row Si;
    };
    enum { K_F0, K_F1, K_B0, K_B1, K_U, K_R0, K_R1 };      // K_R*: row pass of the soft variant
    MPCQP_HD void load_rows(int slot, RowIn& q) {
        q.s0 = Sld(slot + 0); q.l0 = Sld(slot + 1); q.s1 = Sld(slot + 2); q.l1 = Sld(slot + 3);
    }
    template <int K>
    MPCQP_HD void load_stage(int s, StageIn& in) {
        constexpr bool fwd = K == K_F0 || K == K_F1, ph1 = K == K_F1 || K == K_B1 || K == K_R1;
        constexpr bool rowpass = K == K_R0 || K == K_R1;
        const int im = meas_of(s);
        in.x = fwd ? (s < N ? Sld(sm.X + s + 1) : 0.0) : Sld(sm.X + s);
        in.a0 = Sld((K == K_F0 ? sm.Q : K == K_F1 ? sm.RD : (K == K_U || K == K_R1) ? sm.DX : K == K_R0 ? sm.DXA : sm.T) + s);
        if (K == K_F1 || K == K_B1 || K == K_U || K == K_R1) in.dxa = Sld(sm.DXA + s);
        if (cS) {
            if (K == K_B0 || K == K_B1) in.phi = Sld(sm.PHI + s);
            if (K == K_B0) in.tp = Sld(sm.TP + s);
            if (rowpass) in.psi = Sld(sm.PSI + s);
        }
        // (the backward solve sweeps of the soft variant do no row work: the rows wait for dε, see run())
        const bool rows = !(cS && (K == K_B0 || K == K_B1));
        if (cX && rows) load_rows(sm.XR + 4 * s, in.xr);
        if (cW && s < N) {
            if (rows) {
                load_rows(sm.WR + 4 * s, in.wr);
                if (K == K_F0) in.gs = Sld(sm.G + s);
                else in.ww = Sld(sm.WW + s);
                if (ph1 || K == K_U) in.wga = Sld(sm.WGA + s);
                if (K == K_U) in.wg = Sld(sm.WG + s);
            }
            if (K == K_B0 || K == K_B1) in.wd = Sld(sm.WD + s);
        }
        if (cV && im >= 0 && rows) {
            load_rows(sm.VR + 4 * im, in.vr);
            if (K == K_F0) in.es = Sld(sm.E + im);
            else in.vv = Sld(sm.VV + im);
            if (ph1 || K == K_U) in.vga = Sld(sm.VGA + im);
            if (K == K_U) in.vg = Sld(sm.VG + im);
        }
        if (K == K_F1 || K == K_B0 || K == K_B1) Sld_rows(sm.SI + s * NX, in.Si);
    }
    // stages 0..N (forward) or N..0 with the inputs of stage s +- DEPTH already in flight while stage s computes
    template <int K, int DEPTH, class F>
    MPCQP_HD void sweep(bool forward, F&& comp) {
        StageIn c, n1, n2;
        const int s0 = forward ? 0 : N, ds = forward ? 1 : -1;
        if (DEPTH == 0) {
            for (int k = 0; k <= N; ++k) {
                load_stage<K>(s0 + k * ds, c);
                comp(s0 + k * ds, c);
            }
            return;
        }
        load_stage<K>(s0, c);
        if (DEPTH >= 2 && N >= 1) load_stage<K>(s0 + ds, n1);
        for (int k = 0; k <= N; ++k) {
            const int s = s0 + k * ds;
            if (DEPTH >= 2) {
                if (k + 2 <= N) load_stage<K>(s + 2 * ds, n2);
            } else {
                if (k + 1 <= N) load_stage<K>(s + ds, n1);
            }
            comp(s, c);
            c = n1;
            if (DEPTH >= 2) n1 = n2;
        }
    }

    MPCQP_HD void run() {
        const int nx = d.nx, nym = d.nym;
        push_data();
        // ---- constants to LDS
        {
            Row T;
            O::ldo(w.uniform(cbase + cm.Oc), coff, RL, T); O::st(lds, WAVE, T);
            O::ldo(w.uniform(cbase + cm.OcT), coff, RL, T); O::st(lds + (size_t)NX * WAVE, WAVE, T);
#if MPCQP_MHE_BMID_LDS
            O::ldo(w.uniform(cbase + cm.Bmid), coff, RL, T); O::st(lds + (size_t)2 * NX * WAVE, WAVE, T);
#endif
        }
        auto bnd = [&](const double* p_, bool on, int n, double dflt) { return (on && p_ && r < n && !cL) ? p_[(size_t)b * RL + r] : dflt; };
        xlo = bnd(a.xmin, cX, nx, -BIG); xhi = bnd(a.xmax, cX, nx, BIG);
        wlo = bnd(a.wmin, cW, nx, -BIG); whi = bnd(a.wmax, cW, nx, BIG);
        vlo = bnd(a.vmin, cV, nym, -BIG); vhi = bnd(a.vmax, cV, nym, BIG);
        hxlo = xlo > -BIG; hxhi = xhi < BIG; hwlo = wlo > -BIG; hwhi = whi < BIG; hvlo = vlo > -BIG; hvhi = vhi < BIG;
        auto sft = [&](const double* p_, bool on, int n) { return (cS && on && p_ && r < n && !cC) ? p_[(size_t)b * RL + r] : 0.0; };
        cx0 = sft(a.cxmin, cX, nx); cx1 = sft(a.cxmax, cX, nx); cw0 = sft(a.cwmin, cW, nx); cw1 = sft(a.cwmax, cW, nx);
        cv0 = sft(a.cvmin, cV, nym); cv1 = sft(a.cvmax, cV, nym);
        const double Cw = cS ? a.Cwt[b] : 0.0;
        const double lam0 = 10.0;
        // the slack ε (starts at 0), the row ε >= 0 and the slack component of the Newton steps (group-uniform scalars)
        double eps = 0.0, se = 1.0, le = lam0, dea = 0.0, de = 0.0, dsa_e = 0.0, dla_e = 0.0, ds_e = 0.0, dl_e = 0.0, ppsi = 0.0;
        double sel_keep = 0.0, rd_e = 0.0;
        const double xbar = r < nx ? a.X0old[((size_t)b * d.He + yslot(0)) * nx + r] : 0.0;     // x̂0arr_old

        // ---- stage data: g, e, q, starting point (x(0) = x̄, ŵ = 0), slacks and multipliers
        double nh_l = 1.0;
        int m_l = 0;
        {
            Row T;
            double xc = xbar, gprev = 0.0;
            for (int s = 0; s <= N; ++s) {
                stage_bounds(s);
                const double gs = s < N ? g_of(s) : 0.0;
                if (s < N) Sst(sm.G + s, gs);
                double q = 0.0;
                if (s == 0) {
                    O::ldo(a.Pi2 + (size_t)first * NX * RL, (b - first) * NX * RL + r, RL, T);
                    q -= op.mv(T, xbar);
                }
                if (s > 0) {
                    O::ldo(w.uniform(cbase + cm.T2), coff, RL, T);
                    q -= op.mv(T, gprev);
                }
                if (s < N) {
                    O::ld(L_OcT(), WAVE, T);
                    q -= op.mv(T, gs);                  // + Â' 2Q̂⁻¹ g(s)
                }
                const int i = meas_of(s);
                double ei = 0.0;
                if (i >= 0) {
                    ei = e_of(i);
                    Sst(sm.E + i, ei);
                    O::ldo(w.uniform(cbase + cm.CR), coff, RL, T);
                    q -= op.mv(T, ei);
                }
                Sst(sm.Q + s, q);
                Sst(sm.X + s, xc);
                // rows of this stage
                if (cX) {
                    const double s0 = fmax(xc - xlo, 1.0), s1 = fmax(xhi - xc, 1.0);
                    Sst(sm.XR + 4 * s + 0, s0); Sst(sm.XR + 4 * s + 1, lam0 * recip(s0));
                    Sst(sm.XR + 4 * s + 2, s1); Sst(sm.XR + 4 * s + 3, lam0 * recip(s1));
                    m_l += (hxlo ? 1 : 0) + (hxhi ? 1 : 0);
                    if (hxlo) nh_l = fmax(nh_l, fabs(xlo) + 1.0);
                    if (hxhi) nh_l = fmax(nh_l, fabs(xhi) + 1.0);
                }
                if (cW && s < N) {          // ŵ(s) = 0 at the starting point
                    const double s0 = fmax(0.0 - wlo, 1.0), s1 = fmax(whi - 0.0, 1.0);
                    Sst(sm.WR + 4 * s + 0, s0); Sst(sm.WR + 4 * s + 1, lam0 * recip(s0));
                    Sst(sm.WR + 4 * s + 2, s1); Sst(sm.WR + 4 * s + 3, lam0 * recip(s1));
                    m_l += (hwlo ? 1 : 0) + (hwhi ? 1 : 0);
                    if (hwlo) nh_l = fmax(nh_l, fabs(wlo) + 1.0);
                    if (hwhi) nh_l = fmax(nh_l, fabs(whi) + 1.0);
                }
                if (cV && i >= 0) {
                    O::ldo(w.uniform(cbase + cm.Cm), coff, RL, T);
                    const double vv = ei - op.mv(T, xc);
                    const double s0 = fmax(vv - vlo, 1.0), s1 = fmax(vhi - vv, 1.0);
                    Sst(sm.VR + 4 * i + 0, s0); Sst(sm.VR + 4 * i + 1, lam0 * recip(s0));
                    Sst(sm.VR + 4 * i + 2, s1); Sst(sm.VR + 4 * i + 3, lam0 * recip(s1));
                    m_l += (hvlo ? 1 : 0) + (hvhi ? 1 : 0);
                    if (hvlo) nh_l = fmax(nh_l, fabs(vlo) + 1.0);
                    if (hvhi) nh_l = fmax(nh_l, fabs(vhi) + 1.0);
                } else if (cV && i < 0) {
                    // (no v̂ row at this state)
                }
                if (s < N) {                 // x(s+1) = Â x(s) + g(s)
                    O::ldo(w.uniform(cbase + cm.A), coff, RL, T);
                    xc = op.mv(T, xc) + gs;
                }
                gprev = gs;
            }
        }
        const double nh = w.rmax(nh_l);
        const double mrows = w.rsum((double)m_l) + (cS ? 1.0 : 0.0);
        const bool norows = !(mrows > 0.0);
        const double delta = d.dual_reg;
        // ŵ rows couple neighbouring stages: an active one enters S(s+1) as D̃ - (D̃Â)(Â'D̃Â + ..)⁻¹(Â'D̃), a difference of
        // D̃-sized terms.  Their multiplier steps are therefore regularised with δw = 1e-8 (D̃ <= 1e8: the cancellation
        // costs 8 digits, not all 16); δ vanishes from the converged solution either way.
        const double delta_w = fmax(delta, 1e-8);

        int st = 1, it = 0;
        bool done = false;
        double laststep = 1e300, rdn_prev = 1e300, rpn_prev = 1e300, lastscale = 1.0, rpn = 0.0, rd_best = 1e300;
        int nflat = 0;

        for (int pass = 0; pass < d.max_iter; ++pass) {
            double mu = 0.0, smu = 0.0, alpha = 1.0;
            bool ok = true;
            for (int phase = 0; phase < 2; ++phase) {
                // ---------------- forward sweep: residuals (phase 0), right-hand side, factorisation (phase 0), t = S⁻¹ b̃
                double rpn_l = 0.0, mu_l = 0.0, rdn_l = 0.0, ndd_l = 0.0;
                double see_l = 0.0, sel_l = 0.0, sec_l = 0.0;      // slack column: Σ D̃c², Σ cλ, Σ c·c̃ over this lane's soft rows
                {
                    Row Si, Oprev, Bs, U;
                    double xm = 0.0, xc = Sld(sm.X + 0), tprev = 0.0, tpprev = 0.0;
                    double dd_carry = 0.0, gl_carry = 0.0, cr_carry = 0.0, ph_carry = 0.0;
                    auto stage = [&](int s, StageIn& in) {
                        stage_bounds(s);
                        const double xp = in.x;
                        double gl = gl_carry, dd = dd_carry, cr = cr_carry, ph = ph_carry;
                        dd_carry = gl_carry = cr_carry = ph_carry = 0.0;
                        double Dtw = 0.0;
                        // slack column of a row pair with softness (c0, c1): φ part f = D̃0 c0 - D̃1 c1 (in the pair's own space)
                        auto soft = [&](const RowK& k0, const RowK& k1, double l0, double l1, bool h0, bool h1, double c0, double c1) {
                            see_l += k0.Dt * c0 * c0 + k1.Dt * c1 * c1;
                            sec_l += c0 * k0.c + c1 * k1.c;
                            if (!phase) sel_l += (h0 ? c0 * l0 : 0.0) + (h1 ? c1 * l1 : 0.0);
                            return k0.Dt * c0 - k1.Dt * c1;
                        };
                        // phase 1: the rows' complementarity target carries the second-order term of the affine step
                        auto extra = [&](bool has, double sv, double lv, double rp, double gda, double dlt) {
                            double ds, dl;
                            row_dir(has, sv, lv, rp, gda, 0.0, dlt, ds, dl);
                            return ds * dl - smu;
                        };
                        auto tally = [&](bool has, double sv, double lv, double rp) {
                            if (has) { rpn_l = fmax(rpn_l, fabs(rp)); mu_l += sv * lv; }
                        };
                        if (cX) {
                            const RowIn& q = in.xr;
                            const double rp0 = -xc + q.s0 + xlo - cx0 * eps, rp1 = xc + q.s1 - xhi - cx1 * eps;
                            double e0 = 0.0, e1 = 0.0;
                            if (phase) { e0 = extra(hxlo, q.s0, q.l0, rp0, -in.dxa - cx0 * dea, delta); e1 = extra(hxhi, q.s1, q.l1, rp1, in.dxa - cx1 * dea, delta); }
                            const RowK k0 = row_rhs(hxlo, q.s0, q.l0, rp0, e0, delta), k1 = row_rhs(hxhi, q.s1, q.l1, rp1, e1, delta);
                            gl += (hxhi ? q.l1 : 0.0) - (hxlo ? q.l0 : 0.0);
                            dd += k0.Dt + k1.Dt;
                            cr += k1.c - k0.c;
                            if (cS) ph += soft(k0, k1, q.l0, q.l1, hxlo, hxhi, cx0, cx1);
                            if (!phase) { tally(hxlo, q.s0, q.l0, rp0); tally(hxhi, q.s1, q.l1, rp1); }
                        }
                        Row A;
                        if (cW && s < N) {
                            O::ldo(w.uniform(cbase + cm.A), coff, RL, A);
                            double wv;
                            if (!phase) { wv = xp - op.mv(A, xc) - in.gs; Sst(sm.WW + s, wv); }
                            else wv = in.ww;
                            const RowIn& q = in.wr;
                            const double rp0 = -wv + q.s0 + wlo - cw0 * eps, rp1 = wv + q.s1 - whi - cw1 * eps;
                            double e0 = 0.0, e1 = 0.0;
                            if (phase) { e0 = extra(hwlo, q.s0, q.l0, rp0, -in.wga - cw0 * dea, delta_w); e1 = extra(hwhi, q.s1, q.l1, rp1, in.wga - cw1 * dea, delta_w); }
                            const RowK k0 = row_rhs(hwlo, q.s0, q.l0, rp0, e0, delta_w), k1 = row_rhs(hwhi, q.s1, q.l1, rp1, e1, delta_w);
                            const double lw = (hwhi ? q.l1 : 0.0) - (hwlo ? q.l0 : 0.0), cw = k1.c - k0.c;
                            Dtw = k0.Dt + k1.Dt;
                            if (!phase) Sst(sm.WD + s, Dtw);
                            Row At;
                            O::ldo(w.uniform(cbase + cm.At), coff, RL, At);
                            gl -= op.mv(At, lw);
                            cr -= op.mv(At, cw);
                            gl_carry = lw; cr_carry = cw; dd_carry = Dtw;
                            if (cS) {      // ŵ = x(s+1) - Â x(s) - g: the column part f enters stage s+1 as is and stage s through -Â'
                                const double f = soft(k0, k1, q.l0, q.l1, hwlo, hwhi, cw0, cw1);
                                ph -= op.mv(At, f);
                                ph_carry = f;
                            }
                            if (!phase) { tally(hwlo, q.s0, q.l0, rp0); tally(hwhi, q.s1, q.l1, rp1); }
                        }
                        const int im = meas_of(s);
                        double Dtv = 0.0;
                        if (cV && im >= 0) {
                            Row Cm;
                            O::ldo(w.uniform(cbase + cm.Cm), coff, RL, Cm);
                            double vv;
                            if (!phase) { vv = in.es - op.mv(Cm, xc); Sst(sm.VV + im, vv); }
                            else vv = in.vv;
                            const RowIn& q = in.vr;
                            const double rp0 = -vv + q.s0 + vlo - cv0 * eps, rp1 = vv + q.s1 - vhi - cv1 * eps;
                            double e0 = 0.0, e1 = 0.0;
                            if (phase) { e0 = extra(hvlo, q.s0, q.l0, rp0, -in.vga - cv0 * dea, delta); e1 = extra(hvhi, q.s1, q.l1, rp1, in.vga - cv1 * dea, delta); }
                            const RowK k0 = row_rhs(hvlo, q.s0, q.l0, rp0, e0, delta), k1 = row_rhs(hvhi, q.s1, q.l1, rp1, e1, delta);
                            const double lv = (hvhi ? q.l1 : 0.0) - (hvlo ? q.l0 : 0.0), cv = k1.c - k0.c;
                            Dtv = k0.Dt + k1.Dt;
                            if (!phase) Sst(sm.VD + im, Dtv);
                            Row Ct;
                            O::ldo(w.uniform(cbase + cm.Ct), coff, RL, Ct);
                            gl -= op.mv(Ct, lv);            // v̂ = e - Ĉm x: the rows' gradient is -Ĉm'
                            cr -= op.mv(Ct, cv);
                            if (cS) ph -= op.mv(Ct, soft(k0, k1, q.l0, q.l1, hvlo, hvhi, cv0, cv1));
                            if (!phase) { tally(hvlo, q.s0, q.l0, rp0); tally(hvhi, q.s1, q.l1, rp1); }
                        }
                        double rd;
                        if (!phase) {
                            base_block(s, Bs);
                            double hz = op.mv(Bs, xc);
                            if (s > 0) { O::ld(L_Oc(), WAVE, U); hz += op.mv(U, xm); }
                            if (s < N) { O::ld(L_OcT(), WAVE, U); hz += op.mv(U, xp); }
                            const double q = in.a0;
                            rd = hz + q + gl;
                            Sst(sm.RD + s, rd);
                            rdn_l = fmax(rdn_l, fabs(rd));
                            ndd_l = fmax(ndd_l, fmax(fabs(q), fmax(fabs(hz), fabs(gl))));
                        } else {
                            rd = in.a0;
                        }
                        const double rhs = -rd + cr;
                        MPCQP_SCHED_FENCE();
                        // ---- O(s-1) = Oc - D̃w(s-1) Â (re-materialised from LDS: not carried in registers)
                        double otp = 0.0;
                        if (s > 0) {
                            load_O(s - 1, Oprev);
                            otp = op.mv(Oprev, tprev);
                        }
                        // ---- S(s) = Φ(s,s) - O(s-1) Si(s-1) O(s-1)'
                        if (!phase) {
                            O::add_diag(Bs, r, dd);
                            if (cW && s < N) {            // + Â' D̃w Â
                                Row At;
                                O::ldo(w.uniform(cbase + cm.At), coff, RL, At);
                                sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; U[c] = Dtw * A[c]; });
                                sfor<NX>([&](auto ic) {
                                    constexpr int c = decltype(ic)::v;
                                    double acc = 0.0;
                                    sfor<NX / 4>([&](auto ij) { constexpr int k = 4 * decltype(ij)::v; w.template fmabc4<k, k + 1, k + 2, k + 3>(acc, U[c], U[c], U[c], U[c], At[k], At[k + 1], At[k + 2], At[k + 3]); });
                                    Bs[c] += acc;
                                });
                            }
                            if (cV && im >= 0) {          // + Ĉm' D̃v Ĉm
                                Row Cm, Ct;
                                O::ldo(w.uniform(cbase + cm.Cm), coff, RL, Cm);
                                O::ldo(w.uniform(cbase + cm.Ct), coff, RL, Ct);
                                sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; U[c] = Dtv * Cm[c]; });
                                sfor<NX>([&](auto ic) {
                                    constexpr int c = decltype(ic)::v;
                                    double acc = 0.0;
                                    sfor<NX / 4>([&](auto ij) { constexpr int k = 4 * decltype(ij)::v; w.template fmabc4<k, k + 1, k + 2, k + 3>(acc, U[c], U[c], U[c], U[c], Ct[k], Ct[k + 1], Ct[k + 2], Ct[k + 3]); });
                                    Bs[c] += acc;
                                });
                            }
                            if (s > 0) {
                                op.mm(Oprev, Si, U);
                                MPCQP_SCHED_FENCE();
                                op.mmt_sub(U, Oprev, Bs);
                                MPCQP_SCHED_FENCE();
                            }
                            ok = op.gj(Bs, r) && ok;
                            sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Si[c] = Bs[c]; });
                            Sst_rows(sm.SI + s * NX, Si);
                        } else {
                            sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Si[c] = in.Si[c]; });
                        }
                        const double t = op.mv(Si, rhs - otp);
                        Sst(sm.T + s, t);
                        tprev = t;
                        if (cS && !phase) {        // the same forward recursion for the slack column φ
                            const double otpp = s > 0 ? op.mv(Oprev, tpprev) : 0.0;
                            const double tp = op.mv(Si, ph - otpp);
                            Sst(sm.PHI + s, ph);
                            Sst(sm.TP + s, tp);
                            tpprev = tp;
                        }
                        MPCQP_SCHED_FENCE();
                        xm = xc; xc = xp;
                    };
                    if (!phase) sweep<K_F0, MPCQP_MHE_DEPTH_F0>(true, stage); else sweep<K_F1, MPCQP_MHE_DEPTH>(true, stage);
                }
                // ---- slack variable: its row of the Newton system (group-uniform scalars)
                double r_eps = 0.0, phi_ee = 1.0, rp_e = 0.0, ex_e = 0.0;
                if (cS) {
                    const double see = w.rsum(see_l), sec = w.rsum(sec_l);
                    if (!phase) sel_keep = w.rsum(sel_l);
                    rp_e = -eps + se;                                   // row -ε <= 0
                    if (phase) ex_e = dsa_e * dla_e - smu;
                    const RowK ke = row_rhs(true, se, le, rp_e, ex_e, delta);
                    rd_e = 2.0 * Cw * eps - sel_keep - le;
                    phi_ee = 2.0 * Cw + see + ke.Dt;
                    r_eps = -rd_e - sec - ke.c;
                }
                if (!phase) {
                    rpn = w.rmax(rpn_l);
                    double rdn = w.rmax(rdn_l), ndd = w.rmax(ndd_l) + 1.0;
                    double musum = w.rsum(mu_l);
                    if (cS) {
                        rpn = fmax(rpn, fabs(rp_e));
                        rdn = fmax(rdn, fabs(rd_e));
                        ndd = fmax(ndd, fmax(fabs(2.0 * Cw * eps), fabs(sel_keep) + fabs(le)) + 1.0);
                        musum += se * le;
                    }
                    mu = norows ? 0.0 : musum / mrows;
#ifdef MHE_DEBUG_PRINT
                    if (r == 0) printf("[b%d] pass %d mu %.3e rpn %.3e rdn %.3e ndd %.3e laststep %.3e ok %d done %d\n", b, pass, mu, rpn, rdn, ndd, laststep, (int)ok, (int)done);
#endif
                    if (!done) {
                        it = pass;
                        if (!(mu == mu) || !(rdn == rdn)) { st = 2; done = true; }
                        const bool stalled = rdn >= 0.5 * rdn_prev && lastscale <= 0.1;
                        rdn_prev = rdn;
                        // ... or it has not come below half of its best value for four iterations (at the floor the
                        // directions are noise: the steps get short and the rule above, which asks for a nearly
                        // full step, never fires -- a soft-bound family ran into the iteration limit that way)
                        nflat = rdn >= 0.5 * rd_best ? nflat + 1 : 0;
                        rd_best = fmin(rd_best, rdn);
                        const bool flat = nflat >= 4;
                        const bool pstalled = rpn >= 0.5 * rpn_prev && lastscale <= 0.1 && rpn <= 1e-9 * nh;
                        rpn_prev = rpn;
                        if (!done && mu <= d.gap_tol && (rdn <= d.res_tol * ndd || stalled || flat) &&
                            (rpn <= 10.0 * d.res_tol * nh || pstalled) && laststep <= 1e-6) { st = 0; done = true; }
                        if (!done && !ok) { st = 2; done = true; }
                    }
                    if (!w.any(!done)) break;
                }
                // ---------------- backward sweep: dx(s) = t(s) - Si(s) O(s)' dx(s+1); step ratios of the rows
                // (soft variant: the sweep yields y = Φxx⁻¹ r_x and, in phase 0, ψ = Φxx⁻¹ φ; dε = (r_ε - φ'y)/(Φεε - φ'ψ),
                //  dx = y - ψ dε, and the rows are visited by a pass of their own once dε is known)
                double amin_l = 1e300, q1_l = 0.0, q2_l = 0.0, py_l = 0.0, ppsi_l = 0.0;
                const int sDX = phase ? sm.DX : sm.DXA;
                double dcur = 0.0;          // dε of the solve in progress
                auto rows2 = [&](bool h0, bool h1, const RowIn& q, double val, double lo, double hi, double gda, double gd, double c0, double c1, double dlt) {
                    // both rows of one bounded quantity `val` (lo <= val <= hi) with direction gd
                    const double rp0 = -val + q.s0 + lo - c0 * eps, rp1 = val + q.s1 - hi - c1 * eps;
                    double e0 = 0.0, e1 = 0.0, ds, dl;
                    if (phase) {
                        row_dir(h0, q.s0, q.l0, rp0, -gda - c0 * dea, 0.0, dlt, ds, dl); e0 = ds * dl - smu;
                        row_dir(h1, q.s1, q.l1, rp1, gda - c1 * dea, 0.0, dlt, ds, dl); e1 = ds * dl - smu;
                    }
                    row_dir(h0, q.s0, q.l0, rp0, -gd - c0 * dcur, e0, dlt, ds, dl);
                    amin_l = fmin(amin_l, fmin(ratio(q.s0, ds), ratio(q.l0, dl)));
                    q1_l += q.s0 * dl + q.l0 * ds; q2_l += ds * dl;
                    row_dir(h1, q.s1, q.l1, rp1, gd - c1 * dcur, e1, dlt, ds, dl);
                    amin_l = fmin(amin_l, fmin(ratio(q.s1, ds), ratio(q.l1, dl)));
                    q1_l += q.s1 * dl + q.l1 * ds; q2_l += ds * dl;
                };
                // rows of stage s for the direction dx(s) (dxn = dx(s+1)); stores the ŵ / v̂ row directions
                auto rows_stage = [&](int s, StageIn& in, double dx, double dxn) {
                    stage_bounds(s);
                    if (cX) rows2(hxlo, hxhi, in.xr, in.x, xlo, xhi, in.dxa, dx, cx0, cx1, delta);
                    if (cW && s < N) {
                        Row A;
                        O::ldo(w.uniform(cbase + cm.A), coff, RL, A);
                        const double gd = dxn - op.mv(A, dx);
                        Sst((phase ? sm.WG : sm.WGA) + s, gd);
                        rows2(hwlo, hwhi, in.wr, in.ww, wlo, whi, in.wga, gd, cw0, cw1, delta_w);
                    }
                    const int im = meas_of(s);
                    if (cV && im >= 0) {
                        Row Cm;
                        O::ldo(w.uniform(cbase + cm.Cm), coff, RL, Cm);
                        const double gd = -op.mv(Cm, dx);
                        Sst((phase ? sm.VG : sm.VGA) + im, gd);
                        rows2(hvlo, hvhi, in.vr, in.vv, vlo, vhi, in.vga, gd, cv0, cv1, delta);
                    }
                };
                {
                    Row U;
                    double dxn = 0.0, psn = 0.0;
                    auto stage = [&](int s, StageIn& in) {
                        double dx = in.a0, ps = cS && !phase ? in.tp : 0.0;
                        if (s < N) {
                            O::ld(L_OcT(), WAVE, U);
                            double u = op.mv(U, dxn), up = cS && !phase ? op.mv(U, psn) : 0.0;
                            if (cW) {
                                Row At;
                                O::ldo(w.uniform(cbase + cm.At), coff, RL, At);
                                u -= op.mv(At, in.wd * dxn);
                                if (cS && !phase) up -= op.mv(At, in.wd * psn);
                            }
                            dx -= op.mv(in.Si, u);
                            if (cS && !phase) ps -= op.mv(in.Si, up);
                        }
                        Sst(sDX + s, dx);
                        if (cS) {
                            py_l += in.phi * dx;
                            if (!phase) { Sst(sm.PSI + s, ps); ppsi_l += in.phi * ps; }
                        } else {
                            rows_stage(s, in, dx, dxn);
                        }
                        dxn = dx; psn = ps;
                    };
                    if (!phase) sweep<K_B0, MPCQP_MHE_DEPTH>(false, stage); else sweep<K_B1, MPCQP_MHE_DEPTH>(false, stage);
                }
                if (cS) {
                    if (!phase) ppsi = w.rsum(ppsi_l);
                    dcur = (r_eps - w.rsum(py_l)) / (phi_ee - ppsi);
                    if (!phase) dea = dcur; else de = dcur;
                    // row pass: dx = y - ψ dε, then the rows
                    double dxn = 0.0;
                    auto stage = [&](int s, StageIn& in) {
                        const double dx = in.a0 - in.psi * dcur;
                        Sst(sDX + s, dx);
                        rows_stage(s, in, dx, dxn);
                        dxn = dx;
                    };
                    if (!phase) sweep<K_R0, MPCQP_MHE_DEPTH>(false, stage); else sweep<K_R1, MPCQP_MHE_DEPTH>(false, stage);
                    // the row -ε <= 0 itself
                    double ds, dl;
                    row_dir(true, se, le, rp_e, -dcur, ex_e, delta, ds, dl);
                    if (r == 0) {          // (one lane of the group carries it through the reductions)
                        amin_l = fmin(amin_l, fmin(ratio(se, ds), ratio(le, dl)));
                        q1_l += se * dl + le * ds; q2_l += ds * dl;
                    }
                    if (!phase) { dsa_e = ds; dla_e = dl; } else { ds_e = ds; dl_e = dl; }
                }
                const double amin = w.rmin(amin_l);
                const double q1 = w.rsum(q1_l), q2 = w.rsum(q2_l);
                if (!phase) {
                    // centring: mean complementarity after the affine step of length aaff, a quadratic in aaff whose
                    // coefficients the sweep accumulated -- no extra pass over the rows
                    const double aaff = fmin(1.0, amin);
                    const double mas = mu * mrows + aaff * (q1 + aaff * q2);
                    const double sig = (mas / mrows) / mu;
                    smu = sig * sig * sig * mu;
                } else {
                    // fraction to the boundary 0.9999 (the wide-neighbourhood safeguard of the LinMPC kernels costs a
                    // pass over the rows per iteration here and did not lower the iteration count on MHE problems).
                    // An unguarded 0.9999 can jam a rare instance into a cycle (seen once in 65536 C2 controllers,
                    // mpcqp_small_bodies.h): a solve still running after 20 iterations continues with 0.99.
                    alpha = fmin(1.0, (pass >= 20 ? 0.99 : 0.9999) * amin);
                }
            }
            if (!w.any(!done)) break;
            // ---------------- update (a finished estimator of the wavefront keeps its iterate)
            {
                const double al = done ? 0.0 : alpha;
                double zm_l = 1.0, dm_l = 0.0;
                auto upd2 = [&](bool h0, bool h1, int slot, const RowIn& q, double val, double lo, double hi, double gda, double gd, double c0, double c1, double dlt) {
                    const double rp0 = -val + q.s0 + lo - c0 * eps, rp1 = val + q.s1 - hi - c1 * eps;
                    double ds, dl;
                    if (h0) {
                        row_dir(true, q.s0, q.l0, rp0, -gda - c0 * dea, 0.0, dlt, ds, dl);
                        const double e = ds * dl - smu;
                        row_dir(true, q.s0, q.l0, rp0, -gd - c0 * de, e, dlt, ds, dl);
                        Sst(slot + 0, q.s0 + al * ds);
                        Sst(slot + 1, q.l0 + al * dl);
                    }
                    if (h1) {
                        row_dir(true, q.s1, q.l1, rp1, gda - c1 * dea, 0.0, dlt, ds, dl);
                        const double e = ds * dl - smu;
                        row_dir(true, q.s1, q.l1, rp1, gd - c1 * de, e, dlt, ds, dl);
                        Sst(slot + 2, q.s1 + al * ds);
                        Sst(slot + 3, q.l1 + al * dl);
                    }
                };
                auto stage = [&](int s, StageIn& in) {
                    stage_bounds(s);
                    const double xc = in.x, dx = in.a0;
                    if (cX) upd2(hxlo, hxhi, sm.XR + 4 * s, in.xr, xc, xlo, xhi, in.dxa, dx, cx0, cx1, delta);
                    if (cW && s < N) upd2(hwlo, hwhi, sm.WR + 4 * s, in.wr, in.ww, wlo, whi, in.wga, in.wg, cw0, cw1, delta_w);
                    const int im = meas_of(s);
                    if (cV && im >= 0) upd2(hvlo, hvhi, sm.VR + 4 * im, in.vr, in.vv, vlo, vhi, in.vga, in.vg, cv0, cv1, delta);
                    zm_l = fmax(zm_l, fabs(xc));
                    dm_l = fmax(dm_l, fabs(al * dx));
                    Sst(sm.X + s, xc + al * dx);
                };
                sweep<K_U, MPCQP_MHE_DEPTH_U>(true, stage);
                if (cS) { eps += al * de; se += al * ds_e; le += al * dl_e; }
                const double dm = w.rmax(dm_l), zm = w.rmax(zm_l);
                if (!done) {
                    laststep = dm / zm;
                    lastscale = 1.0 - alpha;
                    if (norows) { st = 0; done = true; it = 0; }
                }
            }
            if (!w.any(!done)) break;
        }
        if (st == 1 && !(rpn <= 1e-6 * nh)) st = 2;
        if (live && cS && a.eps_out && r == 0) a.eps_out[b] = st == 2 ? 0.0 : eps;
        write_outputs(st, it, xbar);
    }

    // getstate! (execute.jl:629-643): x̂0(k+p) = x(N); Z̃ = [x̂0arr; Ŵ]; optional V̂, X̂
    MPCQP_HD void write_outputs(int st, int it, double xbar) {
        const int nx = d.nx, nym = d.nym, He = d.He;
        Row A, Cm;
        O::ldo(w.uniform(cbase + cm.A), coff, RL, A);
        O::ldo(w.uniform(cbase + cm.Cm), coff, RL, Cm);
        const bool bad = st == 2;
        // a failed solve keeps the open-loop window: x(0) = x̄, ŵ = 0 (the starting point)
        double xc = bad ? xbar : Sld(sm.X + 0);
        if (live && a.Zt && r < nx) a.Zt[(size_t)b * (nx + He * nx) + r] = xc;
        for (int s = 0; s < N; ++s) {
            const double gs = Sld(sm.G + s);
            const double ax = op.mv(A, xc) + gs;
            const double xn = bad ? ax : Sld(sm.X + s + 1);
            if (live && r < nx) {
                if (a.Zt) a.Zt[(size_t)b * (nx + He * nx) + nx + s * nx + r] = xn - ax;
                if (a.Xhat) a.Xhat[(size_t)b * He * nx + s * nx + r] = xn;
            }
            xc = xn;
            if (a.Vhat) {
                const int i = p ? s : s;       // v̂(i) pairs with state i+1-p
                (void)i;
            }
        }
        if (live && a.Zt && r < nx)
            for (int s = N; s < He; ++s) a.Zt[(size_t)b * (nx + He * nx) + nx + s * nx + r] = 0.0;
        if (a.Vhat) {
            double xs = bad ? xbar : Sld(sm.X + 0);
            for (int s = 0; s <= N; ++s) {
                const int i = meas_of(s);
                const double cx = op.mv(Cm, xs);
                if (i >= 0 && live && r < nym) a.Vhat[(size_t)b * He * nym + i * nym + r] = Sld(sm.E + i) - cx;
                if (s < N) {
                    // (the row product in EVERY group, failed or not: `bad` is per estimator, and a cross-lane operation under a
                    //  per-group condition makes the groups of one wavefront disagree on the sequence of cross-lane operations --
                    //  harmless for the row-local DPP broadcast of the device, fatal for the emulator's barriers: the "stack
                    //  smashing" of ADVICE r4 was this line, reached when one estimator of a wavefront failed and another did not)
                    const double ax = op.mv(A, xs) + Sld(sm.G + s);
                    xs = bad ? ax : Sld(sm.X + s + 1);
                }
            }
        }
        if (live) {
            if (r < nx) a.xhat0[(size_t)b * nx + r] = xc;
            if (r == 0) {
                a.status[b] = st;
                if (a.iters) a.iters[b] = it;
            }
        }
    }
};

template <class W, int NX, unsigned CM = 7u>
MPCQP_HD void step_body(W& w, const Dims& d, const Args& a, int wave_id, double* smem) {
    Solver<W, NX, CM> sv(w, d, a, smem, wave_id);
    const CstMap cm = cst_map(NX, d.nu, d.nd);
    for (int wg = wave_id; wg * GPW < d.B; wg += d.nwaves) {
        const int bq = wg * GPW + sv.g;
        sv.live = bq < d.B;
        sv.b = sv.live ? bq : d.B - 1;
        sv.first = wg * GPW;
        sv.cbase = a.cst + (size_t)sv.first * cm.stride;
        sv.coff = (sv.b - sv.first) * cm.stride + sv.r;
        sv.run();
        w.sync();
    }
}

MPCQP_HD inline size_t step_lds_doubles(int NX) { return (size_t)(MPCQP_MHE_BMID_LDS ? 3 : 2) * NX * WAVE; }

}  // namespace mhe
}  // namespace mpcqp
