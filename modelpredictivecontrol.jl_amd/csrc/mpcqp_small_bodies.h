// mpcqp_small_bodies.h -- the LinMPC step for SMALL problems (nZ̃ <= 16): four controllers per wavefront, one per
// 16-lane DPP row, with the row-per-lane register algebra of the MovingHorizonEstimator kernel (mhe_bodies.h: Ops).
//
// One QP per wavefront leaves most of the wave idle when nZ̃ is small (BASELINE configs[1], "C2": nZ̃ = 11 -> 11 of 64
// lanes busy in the factorisation).  Here lane r of a row owns variable r, row r of H̃ / Φ (nZ̃ registers), its two
// variable bounds and the merged input-bound rows of its (move-blocking interval, channel); Φ⁻¹ comes from the
// in-register Gauss-Jordan sweep, G v / Gᵀw are mat-vecs with the lane's rows of the input-bound matrices.
//
// Same step as Step::build / Step::run of mpcqp_bodies.h (initpred!, linconstraint!, optim_objective!, getinput!:
// src/controller/execute.jl:247-277, 466-505, 536-546; transcription.jl:811-848, 997-1007) for the handles that
// qualify (small_eligible() in mpcqp_launch.h): constraint groups box (hard ΔU bounds, ϵ >= 0), U (hard or soft
// input bounds, one merged row per interval and channel with its multiplicity as barrier weight) and -- variant
// HASY, up to 64 dense rows -- Y (hard or soft output bounds, any horizon-long pattern with +-Inf holes: setconstraint!(ymin,
// ymax, Ymin, Ymax, c_ymin, ...), construct.jl:324-509) and the terminal rows (x̂min, x̂max, c_x̂min, c_x̂max on x̂(k+Hp): the
// rows ex̂ z of transcription.jl:815-821 appended to the same dense block), diagonal weights, no custom rows.  Dual-regularised Mehrotra
// predictor-corrector and active-set polish as everywhere else (one polish attempt per wavefront, see there).
//
// Y rows (HASY): row r = (step t, output a) of  -E z - c0 ϵ <= -Y0min + F,  E z - c1 ϵ <= Y0max - F  belongs to lane r % 16
// of its controller, slot r / 16 (up to four slots per lane: s, λ of both sides in registers).  The dense E (nY x NX,
// block (t, j) = Σ_(t - j_j), transcription.jl:134-165) is laid out once per step in LDS: E v reads the lane's own rows
// against row-broadcasts of v, Eᵀw and EᵀD̃E read its columns against LDS vectors of the row values.
#pragma once
#include <math.h>

#include "mhe_bodies.h"
#include "mpcqp_types.h"

#ifndef MPCQP_SMALL_POLISH
#define MPCQP_SMALL_POLISH 1      // active-set polish of the interior-point iterate, one attempt per wavefront (step_small_body<.., POL = true>):
                                  // C2, 65536 controllers: 13.7 -> 9.9 factorisations, 1.09 -> 0.98 ms.  Not in the one-wave-per-SIMD variant: a launch of
                                  // <= 4096 controllers lasts as long as its slowest wavefront, and that one pays for failed attempts on top of its
                                  // interior-point passes (1024 controllers: 0.088 ms without, 0.109 ms with polish)
#endif
#ifndef MPCQP_SMALL_POLISH_WAIT
#define MPCQP_SMALL_POLISH_WAIT 0  // passes a ready controller waits for the others of its wavefront before the attempt is made without them (0: for ever)
#endif
#ifndef MPCQP_SMALL_POLISH_Y
#define MPCQP_SMALL_POLISH_Y 0     // ... of the variant with dense rows: measured slower (C2 shapes with soft ymax, 65536: 4.55 -> 5.09 ms although 12.4 -> 10.7 iterations: a round costs two passes over the dense rows in LDS)
#endif
#ifndef MPCQP_POLISH_MU
#define MPCQP_POLISH_MU 1e-7
#endif
#ifndef MPCQP_POLISH_RP
#define MPCQP_POLISH_RP 1e-6
#endif

namespace mpcqp {

constexpr int SMALL_GPW = 4, SMALL_RL = 16, SMALL_KY = 4;      // SMALL_KY: Y-row slots per lane (nY <= 64)

// per group: M(F - R̂y) and F (nY each) and the optimum (16) for the optional Ŷ output; with Y rows the dense E
// (nY x NX, NX = nZ̃ rounded up to a multiple of four) and two row vectors
// dense rows of the variant with rows: the output-bound rows (nY, when the handle has any) and the terminal rows (nx̂)
MPCQP_HD inline int small_dense_rows(const Dims& d) {
    return (((d.gmask >> (2 * P_Y)) & 3u) ? d.nY : 0) + (((d.gmask >> (2 * P_X)) & 3u) ? d.nxh : 0);
}
MPCQP_HD inline int small_row_slots(const Dims& d) { const int n = small_dense_rows(d); return n <= 32 ? 2 : n <= 48 ? 3 : 4; }      // KYS
MPCQP_HD inline size_t small_group_doubles(const Dims& d, bool hasy) {
    const int NXv = 4 * ((d.nZ + 3) / 4), nR = small_dense_rows(d);
    return (size_t)(2 * d.nY + SMALL_RL) + (hasy ? (size_t)nR * NXv + 2 * nR + 4 * small_row_slots(d) * SMALL_RL + d.nxh : 0);
}
MPCQP_HD inline size_t small_lds_doubles(const Dims& d, bool hasy = false) { return (size_t)SMALL_GPW * small_group_doubles(d, hasy); }

// KYS: Y-row slots per lane (0: the variant without output-bound rows; 2, 3, 4: nY <= 16 KYS)
// POL: with the active-set polish (the throughput variant k_step_small; the grids of at most one wavefront per SIMD and the dense-row
// variant run without it, see MPCQP_SMALL_POLISH)
template <class W, int NX, int KYS = 0, bool POL = false>
MPCQP_HD void step_small_body(W& w, const Dims& d, const Model& m, const StepIO& io, int wg, double* smem) {
    constexpr bool HASY = KYS > 0;
    using O = mhe::Ops<W, NX>;
    using Row = typename O::Row;
    O op{w};
    const int lane = w.lane, l = lane & (SMALL_RL - 1), g = lane >> 4;
    const int bq = wg * SMALL_GPW + g;
    const bool live = bq < d.B;
    const int b = live ? bq : d.B - 1;
    const int nx = d.nxh, nu = d.nu, ny = d.ny, nd = d.nd, nY = d.nY, nDU = d.nDU, nZ = d.nZ, Hp = d.Hp, Hc = d.Hc;
    const int e = nDU;                         // index of ϵ (when neps)
    const bool isvar = l < nZ, isdu = l < nDU, iseps = d.neps && l == e;
    double* cyv = smem + (size_t)g * small_group_doubles(d, HASY);
    double* Fv = cyv + nY;          // F, kept for the optional Ŷ output
    double* zv = Fv + nY;           // the optimum, for the same
    // (HASY) dense rows: the output-bound rows (nYr = nY or 0) followed by the terminal rows ex̂ z (nXr = nx̂ or 0)
    const int nYr = (HASY && ((d.gmask >> (2 * P_Y)) & 3u)) ? nY : 0, nXr = (HASY && ((d.gmask >> (2 * P_X)) & 3u)) ? nx : 0, nR = nYr + nXr;
    double* Ed = zv + SMALL_RL;     // (HASY) E[r][c], r < nR, c < NX (the ϵ column and the pad columns are zero)
    double* dv = Ed + (HASY ? (size_t)nR * NX : 0);      // (HASY) row vector: D̃lo + D̃hi
    double* wv = dv + (HASY ? nR : 0);                   // (HASY) row vector: multiplier-like values of the rows
    double* ykb = wv + (HASY ? nR : 0);                  // (HASY) row constants [4 KYS][16] per lane: h lower, h upper, c lower, c upper
    double* yk = ykb + l;
    double* fxv = ykb + (HASY ? 4 * KYS * SMALL_RL : 0); // (HASY) terminal free response fx̂ (nx̂)
    const double* x0 = io.xhat0 + (size_t)b * nx;
    const double* lu = io.lastu0 + (size_t)b * nu;
    const double* Stab = m.Stab + (size_t)b * Hp * ny * nu;          // Σ_t [Hp][ny][nu]
    auto jl = [&](int j) { return m.jl[j]; };

    // ---- F = B + K x̂0 + V lastu0 (+ G d0 + J D̂0), cy = M (F - R̂y)             execute.jl:249-275
    {
        const double* K = m.Ktab + (size_t)b * nx * nY;
        const double* Bv = m.Bvec + (size_t)b * nY;
        const double* Md = m.Mdiag + (size_t)b * nY;
        const bool rconst = d.flags & 1u;
        for (int r = l; r < nY; r += SMALL_RL) {
            const int t = r / ny, a = r - t * ny;
            // (loads in batches of four, requested before the first of them is used: a dependent chain of single loads costs
            //  one memory latency each -- the set-up was 20 of the 90 us of a C2 step at 1024 controllers)
            double acc = Bv[r];
            {
                int k = 0;
                for (; k + 4 <= nx; k += 4) {
                    const double k0 = K[(size_t)k * nY + r], k1 = K[(size_t)(k + 1) * nY + r], k2 = K[(size_t)(k + 2) * nY + r], k3 = K[(size_t)(k + 3) * nY + r];
                    acc += k0 * x0[k]; acc += k1 * x0[k + 1]; acc += k2 * x0[k + 2]; acc += k3 * x0[k + 3];
                }
                for (; k < nx; ++k) acc += K[(size_t)k * nY + r] * x0[k];
            }
            for (int cc = 0; cc < nu; ++cc) acc += Stab[(t * ny + a) * nu + cc] * lu[cc];
            if (nd > 0) {
                const double* Gd = m.Gdtab + (size_t)b * Hp * ny * nd;
                const double* dd0 = io.d0 + (size_t)b * nd;
                const double* Dh = io.Dhat0 + (size_t)b * d.nD;
                const double* Dd = m.Dd + (size_t)b * ny * nd;
                for (int q = 0; q < nd; ++q) {
                    acc += Gd[(t * ny + a) * nd + q] * dd0[q];
                    acc += Dd[a + ny * q] * Dh[t * nd + q];
                    for (int j = 0; j < t; ++j) acc += Gd[((t - j - 1) * ny + a) * nd + q] * Dh[j * nd + q];
                }
            }
            const double ry = rconst ? io.Ry[(size_t)b * ny + a] : io.Ry[(size_t)b * nY + r];
            cyv[r] = Md[r] * (acc - ry);
            Fv[r] = acc;
        }
    }
    if constexpr (HASY) {
        const double* exT = nXr ? m.exT + (size_t)b * Hc * nx * nu : nullptr;          // ex̂ block j: [nx̂][nu]
        for (int idx = l; idx < nR * NX; idx += SMALL_RL) {
            const int r = idx / NX, c = idx - r * NX;
            double val = 0.0;
            if (c < nDU) {
                const int jc = c / nu, cc = c - jc * nu;
                if (r < nYr) {
                    const int t = r / ny, a = r - t * ny, t0 = jl(jc);
                    if (t0 <= t) val = Stab[((t - t0) * ny + a) * nu + cc];
                } else {
                    val = exT[(jc * nx + (r - nYr)) * nu + cc];
                }
            }
            Ed[idx] = val;
        }
        // terminal free response fx̂ = bx̂ + kx̂ x̂0 + vx̂ lastu0 (+ gx̂ d0 + jx̂ D̂0)  (transcription.jl:815-821; Step::build)
        for (int i = l; i < nXr; i += SMALL_RL) {
            const double* kx = m.kxT + (size_t)b * nx * nx;
            double acc = m.bxv[(size_t)b * nx + i];
            for (int k = 0; k < nx; ++k) acc += kx[i + nx * k] * x0[k];
            for (int cc = 0; cc < nu; ++cc) acc += exT[i * nu + cc] * lu[cc];              // block j = 0 is vx̂ (j_0 = 0)
            if (nd > 0) {
                const double* Xd = m.Xdtab + (size_t)b * Hp * nx * nd;
                const double* dd0 = io.d0 + (size_t)b * nd;
                const double* Dh = io.Dhat0 + (size_t)b * d.nD;
                for (int q = 0; q < nd; ++q) {
                    acc += Xd[((Hp - 1) * nx + i) * nd + q] * dd0[q];
                    for (int j = 1; j < Hp; ++j) acc += Xd[((Hp - j - 1) * nx + i) * nd + q] * Dh[(j - 1) * nd + q];
                }
            }
            fxv[i] = acc;
        }
    }
    w.sync();
    // ---- q̃ = 2[(M Ẽ)'(F - R̂y) + (L P̃u)'(Tu lastu0 - R̂u)]
    const int jme = isdu ? l / nu : 0, cme = isdu ? l - jme * nu : 0;
    double qv = 0.0;
    if (isdu) {
        const int t0 = jl(jme);
        double acc = 0.0;
        {
            // rows (t, a), t >= t0, are the contiguous range i = (t - t0) ny + a of the table column and of cy
            const int n = (Hp - t0) * ny;
            const double* Sp = Stab + cme;
            const double* cp = cyv + t0 * ny;
            double a1_ = 0.0, a2_ = 0.0, a3_ = 0.0;
            int i = 0;
            for (; i + 8 <= n; i += 8) {
                const double v0 = Sp[i * nu], v1 = Sp[(i + 1) * nu], v2 = Sp[(i + 2) * nu], v3 = Sp[(i + 3) * nu];
                const double v4 = Sp[(i + 4) * nu], v5 = Sp[(i + 5) * nu], v6 = Sp[(i + 6) * nu], v7 = Sp[(i + 7) * nu];
                acc = fma(v0, cp[i], acc); a1_ = fma(v1, cp[i + 1], a1_); a2_ = fma(v2, cp[i + 2], a2_); a3_ = fma(v3, cp[i + 3], a3_);
                acc = fma(v4, cp[i + 4], acc); a1_ = fma(v5, cp[i + 5], a1_); a2_ = fma(v6, cp[i + 6], a2_); a3_ = fma(v7, cp[i + 7], a3_);
            }
            for (; i < n; ++i) acc = fma(Sp[i * nu], cp[i], acc);
            acc += (a1_ + a2_) + a3_;
        }
        const double* Ld = m.Ldiag + (size_t)b * d.nU + cme;
        const double* Rup = io.Ru ? io.Ru + (size_t)b * d.nU + cme : nullptr;
        {
            int t = t0;
            for (; t + 4 <= Hp; t += 4) {
                const double e0 = Ld[t * nu], e1 = Ld[(t + 1) * nu], e2 = Ld[(t + 2) * nu], e3 = Ld[(t + 3) * nu];
                double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
                if (Rup) { r0 = Rup[t * nu]; r1 = Rup[(t + 1) * nu]; r2 = Rup[(t + 2) * nu]; r3 = Rup[(t + 3) * nu]; }
                acc += e0 * (lu[cme] - r0); acc += e1 * (lu[cme] - r1); acc += e2 * (lu[cme] - r2); acc += e3 * (lu[cme] - r3);
            }
            for (; t < Hp; ++t) acc += Ld[t * nu] * (lu[cme] - (Rup ? Rup[t * nu] : 0.0));
        }
        qv = 2.0 * acc;
    }
    // ---- H̃ (row l), the input-bound matrices: rows of the lower / upper merged rows and of their transposes
    Row H, GU, GUt;          // P̃u (held cumulative sum) row of this lane's (interval, channel) and the row of its transpose
    const double* Hpk = m.Hpk + (size_t)b * d.npk;
    // softness of this lane's merged rows: the one of the interval's first step (Step::soft_init)
    double cs0 = 0.0, cs1 = 0.0;
    if (isdu && d.neps) {
        if (m.C_umin) cs0 = m.C_umin[(size_t)b * d.nU + jl(jme) * nu + cme];
        if (m.C_umax) cs1 = m.C_umax[(size_t)b * d.nU + jl(jme) * nu + cme];
    }
    mhe::sfor<NX>([&](auto ic) {
        constexpr int c = decltype(ic)::v;
        const bool in = isvar && c < nZ;
        H[c] = in ? (c <= l ? Hpk[pk(l, c)] : Hpk[pk(c, l)]) : (l == c ? 1.0 : 0.0);
        const int jc = c / nu, cc = c - jc * nu;
        const bool same = isdu && c < nDU && cc == cme;
        const double gu = (same && jc <= jme) ? 1.0 : 0.0;           // P̃u row of (interval, channel) l: held cumulative sum
        const double gut = (same && jc >= jme) ? 1.0 : 0.0;
        GU[c] = gu; GUt[c] = gut;
    });
    // The merged rows are  -P̃u z - cs0 ϵ <= h2  and  P̃u z - cs1 ϵ <= h3: their ΔU part shares P̃u, the ϵ part is the
    // lane's softness -- applied by hand (gmul / gtmul below) instead of carrying four register rows.
    auto epsof = [&](double v) { return d.neps ? w.rsum(iseps ? v : 0.0) : 0.0; };            // component ϵ of a vector
    // G v for the two merged rows of this lane
    auto gmul = [&](double v, double& glo, double& ghi) {
        const double pu = op.mv(GU, v), ve = epsof(v);
        glo = -pu - cs0 * ve; ghi = pu - cs1 * ve;
    };
    // (Gu' w)[l] for multipliers / coefficients w2 (lower rows), w3 (upper rows) of every lane
    auto gtmul = [&](double w2, double w3) {
        const double x = op.mv(GUt, w3 - w2);
        const double se_ = d.neps ? w.rsum(cs0 * w2 + cs1 * w3) : 0.0;
        return iseps ? -se_ : x;
    };
    // ---- Y rows of this lane (HASY): slot q holds row r = l + 16 q
    constexpr int KYM = HASY ? KYS : 1;
    bool yp0[KYM], yp1[KYM];                   // lower / upper row present (finite bound)
    // right-hand sides and softness of the lane's rows: read-only after the set-up, kept in LDS (64 registers otherwise)
    auto yh0 = [&](int q) { return yk[(0 * KYM + q) * SMALL_RL]; };
    auto yh1 = [&](int q) { return yk[(1 * KYM + q) * SMALL_RL]; };
    auto yc0 = [&](int q) { return yk[(2 * KYM + q) * SMALL_RL]; };
    auto yc1 = [&](int q) { return yk[(3 * KYM + q) * SMALL_RL]; };
    double ys0[KYM], ys1[KYM], yl0[KYM], yl1[KYM];       // slacks, multipliers
    int yr[KYM];
    mhe::sfor<KYM>([&](auto iq) {
        constexpr int q = decltype(iq)::v;
        const int r = l + SMALL_RL * q;
        const bool has = HASY && r < nR;
        yr[q] = has ? r : 0;
        double h0_ = 2.0 * BIG, h1_ = 2.0 * BIG, c0_ = 0.0, c1_ = 0.0;
        if (has && r < nYr) {
            const size_t o = (size_t)b * nY + r;
            if (m.Y0min) h0_ = -m.Y0min[o] + Fv[r];
            if (m.Y0max) h1_ = m.Y0max[o] - Fv[r];
            if (d.neps) { c0_ = m.C_ymin ? m.C_ymin[o] : 1.0; c1_ = m.C_ymax ? m.C_ymax[o] : 1.0; }
        } else if (has) {                  // terminal rows  -ex̂ z - c ϵ <= -x̂0min + fx̂,  ex̂ z - c ϵ <= x̂0max - fx̂
            const int i = r - nYr;
            const size_t o = (size_t)b * nx + i;
            if (m.x0min) h0_ = -m.x0min[o] + fxv[i];
            if (m.x0max) h1_ = m.x0max[o] - fxv[i];
            if (d.neps) { c0_ = m.c_x0min ? m.c_x0min[o] : 1.0; c1_ = m.c_x0max ? m.c_x0max[o] : 1.0; }
        }
        if constexpr (HASY) {
            yk[(0 * KYM + q) * SMALL_RL] = h0_; yk[(1 * KYM + q) * SMALL_RL] = h1_;
            yk[(2 * KYM + q) * SMALL_RL] = c0_; yk[(3 * KYM + q) * SMALL_RL] = c1_;
        }
        yp0[q] = fabs(h0_) < BIG && h0_ == h0_; yp1[q] = fabs(h1_) < BIG && h1_ == h1_;
        ys0[q] = ys1[q] = 1.0; yl0[q] = yl1[q] = 0.0;
    });
    // (E v)[r] for the lane's rows: v is one entry per lane, the row reads it through row broadcasts
    auto ymul = [&](double v, double (&gy)[KYM]) {
        if constexpr (HASY) {
            double acc[KYM];
            mhe::sfor<KYM>([&](auto iq) { acc[decltype(iq)::v] = 0.0; });
            mhe::sfor<NX>([&](auto ic) {
                constexpr int c = decltype(ic)::v;
                const double vc = w.template rowbc<c>(v);
                mhe::sfor<KYM>([&](auto iq) { constexpr int q = decltype(iq)::v; acc[q] = fma(Ed[yr[q] * NX + c], vc, acc[q]); });
            });
            mhe::sfor<KYM>([&](auto iq) { constexpr int q = decltype(iq)::v; gy[q] = acc[q]; });
        }
    };
    // (Ey' w)[l]: w = whi - wlo on the ΔU part, -(c0 wlo + c1 whi) summed on ϵ  (absent rows pass zeros)
    auto ytmul = [&](const double (&wlo)[KYM], const double (&whi)[KYM]) {
        if constexpr (HASY) {
            double se = 0.0;
            mhe::sfor<KYM>([&](auto iq) {
                constexpr int q = decltype(iq)::v;
                if (l + SMALL_RL * q < nR) wv[yr[q]] = whi[q] - wlo[q];
                se += yc0(q) * wlo[q] + yc1(q) * whi[q];
            });
            w.sync();
            // (one wave per SIMD: nothing hides an LDS round trip but the loop's own independent work -- four rows in flight)
            double acc4[4] = {0.0, 0.0, 0.0, 0.0};
            const double* Ec = Ed + (isvar ? l : 0);
            int r = 0;
            for (; r + 4 <= nR; r += 4) {
                const double e0_ = Ec[r * NX], e1_ = Ec[(r + 1) * NX], e2_ = Ec[(r + 2) * NX], e3_ = Ec[(r + 3) * NX];
                const double v0_ = wv[r], v1_ = wv[r + 1], v2_ = wv[r + 2], v3_ = wv[r + 3];
                acc4[0] = fma(e0_, v0_, acc4[0]); acc4[1] = fma(e1_, v1_, acc4[1]); acc4[2] = fma(e2_, v2_, acc4[2]); acc4[3] = fma(e3_, v3_, acc4[3]);
            }
            for (; r < nR; ++r) acc4[0] = fma(Ec[r * NX], wv[r], acc4[0]);
            const double acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            const double set = d.neps ? w.rsum(se) : 0.0;
            w.sync();
            return iseps ? -set : (isdu ? acc : 0.0);
        } else {
            return 0.0;
        }
    };
    // ---- rows of this lane: 0 box lower, 1 box upper, 2 merged Umin, 3 merged Umax          (i_b: finite only)
    double h0 = 2.0 * BIG, h1 = 2.0 * BIG, h2 = 2.0 * BIG, h3 = 2.0 * BIG, wt = 1.0;
    if (isdu) {
        const size_t o = (size_t)b * nDU + l;
        if (m.DUmin && (!d.neps || !m.C_dumin || m.C_dumin[o] == 0.0)) h0 = -m.DUmin[o];
        if (m.DUmax && (!d.neps || !m.C_dumax || m.C_dumax[o] == 0.0)) h1 = m.DUmax[o];
        const int t0 = jl(jme), t1 = (jme + 1 < Hc) ? jl(jme + 1) : Hp;
        wt = (double)(t1 - t0);                   // multiplicity of the merged row (barrier weight)
        if (m.U0min) {
            const double* p = m.U0min + (size_t)b * d.nU + cme;
            double v = -INFINITY;
            int t = t0;
            for (; t + 4 <= t1; t += 4) {
                const double v0 = p[t * nu], v1 = p[(t + 1) * nu], v2 = p[(t + 2) * nu], v3 = p[(t + 3) * nu];
                v = fmax(fmax(v, v0), fmax(fmax(v1, v2), v3));
            }
            for (; t < t1; ++t) v = fmax(v, p[t * nu]);
            h2 = -v + lu[cme];
        }
        if (m.U0max) {
            const double* p = m.U0max + (size_t)b * d.nU + cme;
            double v = INFINITY;
            int t = t0;
            for (; t + 4 <= t1; t += 4) {
                const double v0 = p[t * nu], v1 = p[(t + 1) * nu], v2 = p[(t + 2) * nu], v3 = p[(t + 3) * nu];
                v = fmin(fmin(v, v0), fmin(fmin(v1, v2), v3));
            }
            for (; t < t1; ++t) v = fmin(v, p[t * nu]);
            h3 = v - lu[cme];
        }
    } else if (iseps) {
        h0 = 0.0;                                 // ϵ >= 0
    }
    auto fin = [](double h) { return fabs(h) < BIG && h == h; };
    const bool p0 = fin(h0), p1 = fin(h1), p2 = fin(h2), p3 = fin(h3);
    const double w0 = 1.0, w1 = 1.0, w2 = wt, w3 = wt;
    double ycnt = 0.0, yhm = 0.0;
    mhe::sfor<KYM>([&](auto iq) {
        constexpr int q = decltype(iq)::v;
        ycnt += (yp0[q] ? 1.0 : 0.0) + (yp1[q] ? 1.0 : 0.0);
        yhm = fmax(yhm, fmax(yp0[q] ? fabs(yh0(q)) : 0.0, yp1[q] ? fabs(yh1(q)) : 0.0));
    });
    const double wsum = w.rsum((p0 ? w0 : 0.0) + (p1 ? w1 : 0.0) + (p2 ? w2 : 0.0) + (p3 ? w3 : 0.0) + ycnt);
    const bool norows = !(wsum > 0.0);
    const double nh = 1.0 + w.rmax(fmax(fmax(fmax(p0 ? fabs(h0) : 0.0, p1 ? fabs(h1) : 0.0), fmax(p2 ? fabs(h2) : 0.0, p3 ? fabs(h3) : 0.0)), yhm));

    // ---- warm start  Z̃s = [Z̃prev[nu+1:nΔU]; 0; ϵprev]                             transcription.jl:1001-1004
    const double* Zg = io.Z + (size_t)b * nZ;
    const bool cold = d.flags & 2u;
    double z = 0.0;
    if (isvar && !cold) z = (l < nDU - nu) ? Zg[l + nu] : (l >= nDU ? Zg[l] : 0.0);
    const double zws = z;
    // starting point: s = max(h - G z, 1), λ = 10 w / s
    double s0, s1, s2, s3, l0, l1, l2, l3;
    {
        double g2, g3;
        gmul(z, g2, g3);
        s0 = fmax(h0 + z, 1.0); s1 = fmax(h1 - z, 1.0); s2 = fmax(h2 - g2, 1.0); s3 = fmax(h3 - g3, 1.0);
        if (!p0) s0 = 1.0; if (!p1) s1 = 1.0; if (!p2) s2 = 1.0; if (!p3) s3 = 1.0;
        l0 = p0 ? 10.0 * w0 * mhe::recip(s0) : 0.0; l1 = p1 ? 10.0 * w1 * mhe::recip(s1) : 0.0; l2 = p2 ? 10.0 * w2 * mhe::recip(s2) : 0.0; l3 = p3 ? 10.0 * w3 * mhe::recip(s3) : 0.0;
        if constexpr (HASY) {
            double gy[KYM];
            ymul(z, gy);
            const double ze = epsof(z);
            mhe::sfor<KYM>([&](auto iq) {
                constexpr int q = decltype(iq)::v;
                ys0[q] = yp0[q] ? fmax(yh0(q) + gy[q] + yc0(q) * ze, 1.0) : 1.0;
                ys1[q] = yp1[q] ? fmax(yh1(q) - gy[q] + yc1(q) * ze, 1.0) : 1.0;
                yl0[q] = yp0[q] ? 10.0 * mhe::recip(ys0[q]) : 0.0;
                yl1[q] = yp1[q] ? 10.0 * mhe::recip(ys1[q]) : 0.0;
            });
        }
    }
    const double delta = d.dual_reg;
    int st = 1, it = 0, npol = 0, waited = 0;
    bool done = false;
    double polmu_next = MPCQP_POLISH_MU, polished = 0.0;
    double laststep = 1e300, rdn_prev = 1e300, rpn_prev = 1e300, lastscale = 1.0, rpn = 0.0;
    double mu_seen = 0.0, rd_seen = 0.0;      // what the convergence test saw last (audit record)
    // one reciprocal per row, wi = 1/(s + δλ): D̃ = λ wi, w/s = wi (mhe::row_rhs)
    struct RowD { double Dt, wi; };
    auto rowd = [&](bool has, double sv, double lv) {
        RowD r;
        r.wi = mhe::recip(fma(delta, lv, sv));
        r.Dt = has ? lv * r.wi : 0.0;
        return r;
    };
    // Φ = H̃ + Gᵀ D G for row factors D of this lane's rows (the interior-point iterations: D̃ = λ / (s + δλ); the polish: ρ on its
    // active rows)
    const double meps = iseps ? 1.0 : 0.0;
    auto build_phi = [&](double D0, double D1, double D2, double D3, const double (&yD0)[KYM], const double (&yD1)[KYM], Row& Phi) {
        // P̃u' (D̃2 + D̃3) P̃u without forming the product: entry (l, c) of two variables of the same input channel is the
        // sum of D̃ over the intervals from the later of the two on, i.e. the suffix sum `suf` of the later one -- the
        // lane's own for the columns up to its own, the column's lane's (row broadcast) for the later columns.
        // (GU = 1 on the columns c <= l of the lane's channel, GUt = 1 on the columns c >= l: both include c = l.)
        const double suf = op.mv(GUt, D2 + D3);
        // (Φ[l][c] = H̃ + GU[c] suf(l) + GUt[c] suf(c): the second product takes its left factor from lane c -- the row
        //  broadcast is the multiply-add's own DPP modifier; both count the diagonal, taken out again with the box rows' D̃)
        mhe::sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Phi[c] = fma(GU[c], suf, H[c]); });
        mhe::sfor<NX / 4>([&](auto ij) {
            constexpr int c = 4 * decltype(ij)::v;
            w.template rank1bc4<c, c + 1, c + 2, c + 3>(Phi[c], Phi[c + 1], Phi[c + 2], Phi[c + 3], suf, GUt[c], GUt[c + 1], GUt[c + 2], GUt[c + 3]);
        });
        O::add_diag(Phi, l, D0 + D1 - (isdu ? suf : 0.0));
        // ϵ row and column of a group of soft rows: the row (lane ϵ) takes the column vector `col` of the ΔU lanes across the
        // row -- zero on the other lanes by construction --, every ΔU lane its own entry into column ϵ, lane ϵ the diagonal
        auto eps_border = [&](double col, double dee) {
            mhe::sfor<NX / 4>([&](auto ij) {
                constexpr int c = 4 * decltype(ij)::v;
                w.template rank1bc4<c, c + 1, c + 2, c + 3>(Phi[c], Phi[c + 1], Phi[c + 2], Phi[c + 3], col, meps, meps, meps, meps);
            });
            const double xe = iseps ? dee : (isdu ? col : 0.0);
            mhe::sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Phi[c] += (c == e) ? xe : 0.0; });
        };
        if (d.neps) {      // ϵ column / row: Φ[k][ϵ] = sum_j P̃u[j][k] (D̃2 cs0 - D̃3 cs1)_j,  Φ[ϵ][ϵ] += sum_j D̃2 cs0² + D̃3 cs1²
            const double col = op.mv(GUt, D2 * cs0 - D3 * cs1);
            const double dee = w.rsum(D2 * cs0 * cs0 + D3 * cs1 * cs1);
            eps_border(col, dee);
        }
        if constexpr (HASY) {      // + Ey' (D̃lo + D̃hi) Ey, the ϵ column Ey'(D̃lo c0 - D̃hi c1) and Φ[ϵ][ϵ] += sum D̃lo c0² + D̃hi c1²
            double dee = 0.0;
            mhe::sfor<KYM>([&](auto iq) {
                constexpr int q = decltype(iq)::v;
                if (l + SMALL_RL * q < nR) {
                    dv[yr[q]] = yD0[q] + yD1[q];
                    wv[yr[q]] = yD0[q] * yc0(q) - yD1[q] * yc1(q);
                }
                dee += yD0[q] * yc0(q) * yc0(q) + yD1[q] * yc1(q) * yc1(q);
            });
            w.sync();
            double col = 0.0;
            const int lc = isvar ? l : 0;
            auto rank1 = [&](int r, double el, double dr, double wr, const Row& Er) {
                el = isdu ? el : 0.0;
                const double tt = el * dr;
                col = fma(el, wr, col);
                mhe::sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Phi[c] = fma(tt, Er[c], Phi[c]); });
            };
            int r = 0;
            for (; r + 2 <= nR; r += 2) {          // two rows in flight: the loads of both are issued before either update
                Row Ea, Eb;
                mhe::sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Ea[c] = Ed[r * NX + c]; Eb[c] = Ed[(r + 1) * NX + c]; });
                const double ela = Ed[r * NX + lc], elb = Ed[(r + 1) * NX + lc], da = dv[r], db = dv[r + 1], wa = wv[r], wb = wv[r + 1];
                rank1(r, ela, da, wa, Ea);
                rank1(r + 1, elb, db, wb, Eb);
            }
            for (; r < nR; ++r) {
                Row Ea;
                mhe::sfor<NX>([&](auto ic) { constexpr int c = decltype(ic)::v; Ea[c] = Ed[r * NX + c]; });
                rank1(r, Ed[r * NX + lc], dv[r], wv[r], Ea);
            }
            if (d.neps) eps_border(col, w.rsum(dee));
            w.sync();
        }
    };
    for (int pass = 0; pass < d.max_iter; ++pass) {
        // ---- residuals
        double g2, g3;
        gmul(z, g2, g3);
        const double rp0 = -z + s0 - h0, rp1 = z + s1 - h1, rp2 = g2 + s2 - h2, rp3 = g3 + s3 - h3;
        const double hz = op.mv(H, z);
        double yrp0[KYM], yrp1[KYM];             // primal residuals of the Y rows:  G z + s - h
        double yrpm = 0.0, ymus = 0.0;
        if constexpr (HASY) {
            double gy[KYM];
            ymul(z, gy);
            const double ze = epsof(z);
            mhe::sfor<KYM>([&](auto iq) {
                constexpr int q = decltype(iq)::v;
                yrp0[q] = -gy[q] - yc0(q) * ze + ys0[q] - yh0(q);
                yrp1[q] = gy[q] - yc1(q) * ze + ys1[q] - yh1(q);
                yrpm = fmax(yrpm, fmax(yp0[q] ? fabs(yrp0[q]) : 0.0, yp1[q] ? fabs(yrp1[q]) : 0.0));
                ymus += (yp0[q] ? ys0[q] * yl0[q] : 0.0) + (yp1[q] ? ys1[q] * yl1[q] : 0.0);
            });
        }
        const double gl = (p1 ? l1 : 0.0) - (p0 ? l0 : 0.0) + gtmul(p2 ? l2 : 0.0, p3 ? l3 : 0.0) + ytmul(yl0, yl1);
        const double rd = isvar ? hz + qv + gl : 0.0;
        rpn = w.rmax(fmax(fmax(fmax(p0 ? fabs(rp0) : 0.0, p1 ? fabs(rp1) : 0.0), fmax(p2 ? fabs(rp2) : 0.0, p3 ? fabs(rp3) : 0.0)), yrpm));
        const double rdn = w.rmax(fabs(rd));
        const double ndd = w.rmax(isvar ? fmax(fabs(qv), fmax(fabs(hz), fabs(gl))) : 0.0) + 1.0;
        const double musum = w.rsum((p0 ? s0 * l0 : 0.0) + (p1 ? s1 * l1 : 0.0) + (p2 ? s2 * l2 : 0.0) + (p3 ? s3 * l3 : 0.0) + ymus);
        const double mu = norows ? 0.0 : musum / wsum;
#ifdef MHE_DEBUG_PRINT
        if (l == 0) printf("[b%d] pass %d mu %.3e rpn %.3e rdn %.3e ndd %.3e laststep %.3e lastscale %.3e done %d\n", b, pass, mu, rpn, rdn, ndd, laststep, lastscale, (int)done);
#endif
        if (!done) {
            it = pass;
            mu_seen = mu; rd_seen = rdn / ndd;
            if (!(mu == mu) || !(rdn == rdn)) { st = 2; done = true; }
            const bool stalled = rdn >= 0.5 * rdn_prev && lastscale <= 0.1;
            rdn_prev = rdn;
            const bool pstalled = rpn >= 0.5 * rpn_prev && lastscale <= 0.1 && rpn <= 1e-9 * nh;
            rpn_prev = rpn;
            if (!done && mu <= d.gap_tol && (rdn <= d.res_tol * ndd || stalled) && (rpn <= 10.0 * d.res_tol * nh || pstalled) &&
                laststep <= 1e-6) { st = 0; done = true; }
        }
        if (!w.any(!done)) break;
#if MPCQP_SMALL_POLISH
        // ---- active-set polish (Step::polish of mpcqp_bodies.h, oracle/linmpc_ref.c): once the gap is below MPCQP_POLISH_MU the rows
        // with λ > s are taken as the active set A and the equality-constrained QP on A is solved by Newton steps on its
        // augmented Lagrangian (ρ = 1e10, exact residuals every round); the point is accepted with the KKT conditions of the
        // inequality QP.  The four controllers of a wavefront share the instruction stream: an attempt is made when every
        // unfinished one is ready for it (a ready controller iterates on while it waits: measured on C2, an attempt costs two
        // passes, so a wavefront should pay for one attempt, not for one per controller).
        {
            const bool want = POL && (MPCQP_SMALL_POLISH_Y || !HASY) && !done && !norows && !(d.flags & 16u) && mu <= polmu_next && rpn <= MPCQP_POLISH_RP * nh && npol < 4;
            if (w.any(want) && (!w.any(!done && !want) || (MPCQP_SMALL_POLISH_WAIT > 0 && w.any(want && ++waited >= MPCQP_SMALL_POLISH_WAIT)))) {
                constexpr double rho = 1e10;
                const bool A0 = want && p0 && l0 > s0, A1 = want && p1 && l1 > s1, A2 = want && p2 && l2 > s2, A3 = want && p3 && l3 > s3;
                bool yA0[KYM], yA1[KYM];
                double lp0 = A0 ? l0 : 0.0, lp1 = A1 ? l1 : 0.0, lp2 = A2 ? l2 : 0.0, lp3 = A3 ? l3 : 0.0;
                double ylp0[KYM], ylp1[KYM], yr0[KYM], yr1[KYM];
                mhe::sfor<KYM>([&](auto iq) {
                    constexpr int q = decltype(iq)::v;
                    yA0[q] = HASY && want && yp0[q] && yl0[q] > ys0[q]; yA1[q] = HASY && want && yp1[q] && yl1[q] > ys1[q];
                    ylp0[q] = yA0[q] ? yl0[q] : 0.0; ylp1[q] = yA1[q] ? yl1[q] : 0.0;
                    yr0[q] = yA0[q] ? rho : 0.0; yr1[q] = yA1[q] ? rho : 0.0;
                });
                Row Pp;
                build_phi(A0 ? rho : 0.0, A1 ? rho : 0.0, A2 ? rho : 0.0, A3 ? rho : 0.0, yr0, yr1, Pp);
                const bool run = op.gj(Pp, l) && want;
                double zp = z, pg2 = 0.0, pg3 = 0.0, pgy[KYM], pze = 0.0, rpa_prev = 1e300;
                mhe::sfor<KYM>([&](auto iq) { pgy[decltype(iq)::v] = 0.0; });
                bool okp = false, gaveup = false;
                for (int round = 0; round <= 8; ++round) {
                    gmul(zp, pg2, pg3);
                    if constexpr (HASY) { ymul(zp, pgy); pze = epsof(zp); }
                    const double ra0 = A0 ? -zp - h0 : 0.0, ra1 = A1 ? zp - h1 : 0.0, ra2 = A2 ? pg2 - h2 : 0.0, ra3 = A3 ? pg3 - h3 : 0.0;
                    double yra0[KYM], yra1[KYM], yram = 0.0;
                    mhe::sfor<KYM>([&](auto iq) {
                        constexpr int q = decltype(iq)::v;
                        yra0[q] = yA0[q] ? -pgy[q] - yc0(q) * pze - yh0(q) : 0.0;
                        yra1[q] = yA1[q] ? pgy[q] - yc1(q) * pze - yh1(q) : 0.0;
                        yram = fmax(yram, fmax(fabs(yra0[q]), fabs(yra1[q])));
                    });
                    const double rpan = w.rmax(fmax(fmax(fmax(fabs(ra0), fabs(ra1)), fmax(fabs(ra2), fabs(ra3))), yram));
                    const double hzp = op.mv(H, zp);
                    const double glp = (lp1 - lp0) + gtmul(lp2, lp3) + ytmul(ylp0, ylp1);
                    const double gtv = isvar ? hzp + qv + glp : 0.0;
                    const double rdn2 = w.rmax(fabs(gtv));
                    const double ndd2 = w.rmax(isvar ? fmax(fabs(qv), fmax(fabs(hzp), fabs(glp))) : 0.0);
                    if (run && !okp && !gaveup && rpan <= 1e-13 * nh && rdn2 <= 1e-14 * (1.0 + ndd2)) okp = true;
                    // (a round that no longer contracts the residual of the active rows: dependent active rows, wrong set)
                    if (round >= 2 && !(rpan < 0.25 * rpa_prev) && rpan > 1e-13 * nh) gaveup = true;
                    rpa_prev = rpan;
                    const bool go = run && !okp && !gaveup;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_DEBUG_POLISH)
                    if (l == 0) printf("  small polish wg %d g %d pass %d round %d want %d run %d rpan %.3e rdn2 %.3e (tol %.1e) okp %d gaveup %d\n", wg, g, pass, round, (int)want, (int)run, rpan, rdn2, 1e-14 * (1.0 + ndd2), (int)okp, (int)gaveup);
#endif
                    if (round == 8 || !w.any(go)) break;
                    const double c0 = rho * ra0, c1 = rho * ra1, c2 = rho * ra2, c3 = rho * ra3;
                    double yc0v[KYM], yc1v[KYM];
                    mhe::sfor<KYM>([&](auto iq) { constexpr int q = decltype(iq)::v; yc0v[q] = rho * yra0[q]; yc1v[q] = rho * yra1[q]; });
                    const double gtc = gtmul(c2, c3) + ytmul(yc0v, yc1v);       // (every lane takes part in the mat-vecs)
                    const double rhs = isvar ? -gtv - ((c1 - c0) + gtc) : 0.0;
                    const double dzp = op.mv(Pp, rhs);
                    double gd2, gd3, gdy[KYM], dze = 0.0;
                    gmul(dzp, gd2, gd3);
                    mhe::sfor<KYM>([&](auto iq) { gdy[decltype(iq)::v] = 0.0; });
                    if constexpr (HASY) { ymul(dzp, gdy); dze = epsof(dzp); }
                    if (go) {
                        zp += dzp;
                        if (A0) lp0 += rho * (ra0 - dzp);
                        if (A1) lp1 += rho * (ra1 + dzp);
                        if (A2) lp2 += rho * (ra2 + gd2);
                        if (A3) lp3 += rho * (ra3 + gd3);
                        mhe::sfor<KYM>([&](auto iq) {
                            constexpr int q = decltype(iq)::v;
                            if (yA0[q]) ylp0[q] += rho * (yra0[q] - gdy[q] - yc0(q) * dze);
                            if (yA1[q]) ylp1[q] += rho * (yra1[q] + gdy[q] - yc1(q) * dze);
                        });
                    }
                }
                // KKT conditions of the inequality QP: multipliers >= 0 on A, the other rows feasible (G zp of the last round)
                {
                    double lm = fmax(fmax(fabs(lp0), fabs(lp1)), fmax(fabs(lp2), fabs(lp3)));
                    mhe::sfor<KYM>([&](auto iq) { constexpr int q = decltype(iq)::v; lm = fmax(lm, fmax(fabs(ylp0[q]), fabs(ylp1[q]))); });
                    const double ltol = -1e-12 * (1.0 + w.rmax(lm)), stol = -1e-11 * nh;
                    bool bad = (A0 && lp0 < ltol) || (A1 && lp1 < ltol) || (A2 && lp2 < ltol) || (A3 && lp3 < ltol);
                    bad = bad || (p0 && !A0 && h0 + zp < stol) || (p1 && !A1 && h1 - zp < stol) || (p2 && !A2 && h2 - pg2 < stol) || (p3 && !A3 && h3 - pg3 < stol);
                    mhe::sfor<KYM>([&](auto iq) {
                        constexpr int q = decltype(iq)::v;
                        bad = bad || (yA0[q] && ylp0[q] < ltol) || (yA1[q] && ylp1[q] < ltol);
                        bad = bad || (yp0[q] && !yA0[q] && yh0(q) + pgy[q] + yc0(q) * pze < stol) || (yp1[q] && !yA1[q] && yh1(q) - pgy[q] + yc1(q) * pze < stol);
                    });
                    const bool anybad = w.rmax(bad ? 1.0 : 0.0) > 0.0;
                    okp = okp && !anybad;
                }
                if (want) { ++npol; polmu_next = 1e-2 * mu; waited = 0; }
                if (okp) { z = zp; st = 0; done = true; it = pass + npol; polished = 1.0; }
                if (!w.any(!done)) break;
            }
        }
#endif
        // ---- Φ = H̃ + Gᵀ D̃ G, Φ⁻¹
        const RowD d0 = rowd(p0, s0, l0), d1 = rowd(p1, s1, l1), d2 = rowd(p2, s2, l2), d3 = rowd(p3, s3, l3);
        Row Phi;
        RowD yd0[KYM], yd1[KYM];
        double ydt0[KYM], ydt1[KYM];
        mhe::sfor<KYM>([&](auto iq) {
            constexpr int q = decltype(iq)::v;
            yd0[q] = rowd(HASY && yp0[q], ys0[q], yl0[q]); yd1[q] = rowd(HASY && yp1[q], ys1[q], yl1[q]);
            ydt0[q] = yd0[q].Dt; ydt1[q] = yd1[q].Dt;
        });
        build_phi(d0.Dt, d1.Dt, d2.Dt, d3.Dt, ydt0, ydt1, Phi);
        const bool ok = op.gj(Phi, l);
        if (!done && !ok) { st = 2; done = true; }
        // ---- predictor, corrector
        double smu = 0.0, alpha = 1.0, dz = 0.0;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;            // Δs Δλ of the affine step
        double ya0[KYM], ya1[KYM], yds0[KYM], yds1[KYM], ydl0[KYM], ydl1[KYM];
        mhe::sfor<KYM>([&](auto iq) { constexpr int q = decltype(iq)::v; ya0[q] = ya1[q] = yds0[q] = yds1[q] = ydl0[q] = ydl1[q] = 0.0; });
        double ds0 = 0, ds1 = 0, ds2 = 0, ds3 = 0, dl0 = 0, dl1 = 0, dl2 = 0, dl3 = 0;
        for (int phase = 0; phase < 2; ++phase) {
            auto cof = [&](bool has, const RowD& rr, double sv, double lv, double rp, double ex) {
                return has ? fma(rr.wi, fma(sv, lv, ex), -rr.Dt * rp) : 0.0;
            };
            const double e0 = phase ? a0 - w0 * smu : 0.0, e1 = phase ? a1 - w1 * smu : 0.0;
            const double e2 = phase ? a2 - w2 * smu : 0.0, e3 = phase ? a3 - w3 * smu : 0.0;
            const double c0 = cof(p0, d0, s0, l0, rp0, e0), c1 = cof(p1, d1, s1, l1, rp1, e1);
            const double c2 = cof(p2, d2, s2, l2, rp2, e2), c3 = cof(p3, d3, s3, l3, rp3, e3);
            double ye0[KYM], ye1[KYM], ycf0[KYM], ycf1[KYM];
            mhe::sfor<KYM>([&](auto iq) {
                constexpr int q = decltype(iq)::v;
                ye0[q] = phase ? ya0[q] - smu : 0.0; ye1[q] = phase ? ya1[q] - smu : 0.0;
                ycf0[q] = HASY ? cof(yp0[q], yd0[q], ys0[q], yl0[q], yrp0[q], ye0[q]) : 0.0;
                ycf1[q] = HASY ? cof(yp1[q], yd1[q], ys1[q], yl1[q], yrp1[q], ye1[q]) : 0.0;
            });
            const double gtc = gtmul(c2, c3) + ytmul(ycf0, ycf1);       // (every lane takes part in the mat-vecs)
            const double rhs = isvar ? -rd + (c1 - c0) + gtc : 0.0;
            dz = op.mv(Phi, rhs);
            double gd2, gd3;
            gmul(dz, gd2, gd3);
            double gdy[KYM];
            double dze = 0.0;
            if constexpr (HASY) { ymul(dz, gdy); dze = epsof(dz); }
            auto dir = [&](bool has, const RowD& rr, double sv, double lv, double rp, double gd, double ex, double& ds, double& dl) {
                const double rc = fma(sv, lv, ex), a = rp + gd;
                dl = has ? rr.wi * fma(lv, a, -rc) : 0.0;
                ds = has ? -rr.wi * fma(sv, a, delta * rc) : 0.0;
            };
            dir(p0, d0, s0, l0, rp0, -dz, e0, ds0, dl0); dir(p1, d1, s1, l1, rp1, dz, e1, ds1, dl1);
            dir(p2, d2, s2, l2, rp2, gd2, e2, ds2, dl2); dir(p3, d3, s3, l3, rp3, gd3, e3, ds3, dl3);
            double am = 1e300;
            if constexpr (HASY) {
                mhe::sfor<KYM>([&](auto iq) {
                    constexpr int q = decltype(iq)::v;
                    dir(yp0[q], yd0[q], ys0[q], yl0[q], yrp0[q], -gdy[q] - yc0(q) * dze, ye0[q], yds0[q], ydl0[q]);
                    dir(yp1[q], yd1[q], ys1[q], yl1[q], yrp1[q], gdy[q] - yc1(q) * dze, ye1[q], yds1[q], ydl1[q]);
                    am = fmin(am, fmin(fmin(mhe::ratio(ys0[q], yds0[q]), mhe::ratio(yl0[q], ydl0[q])),
                                       fmin(mhe::ratio(ys1[q], yds1[q]), mhe::ratio(yl1[q], ydl1[q]))));
                });
            }
            am = fmin(am, fmin(fmin(mhe::ratio(s0, ds0), mhe::ratio(l0, dl0)), fmin(mhe::ratio(s1, ds1), mhe::ratio(l1, dl1))));
            am = fmin(am, fmin(fmin(mhe::ratio(s2, ds2), mhe::ratio(l2, dl2)), fmin(mhe::ratio(s3, ds3), mhe::ratio(l3, dl3))));
            const double amin = w.rmin(am);
            if (!phase) {
                const double aaff = fmin(1.0, amin);
                double ymas = 0.0;
                mhe::sfor<KYM>([&](auto iq) {
                    constexpr int q = decltype(iq)::v;
                    ymas += (yp0[q] ? (ys0[q] + aaff * yds0[q]) * (yl0[q] + aaff * ydl0[q]) : 0.0) +
                            (yp1[q] ? (ys1[q] + aaff * yds1[q]) * (yl1[q] + aaff * ydl1[q]) : 0.0);
                    ya0[q] = yds0[q] * ydl0[q]; ya1[q] = yds1[q] * ydl1[q];
                });
                const double mas = w.rsum((s0 + aaff * ds0) * (l0 + aaff * dl0) * (p0 ? 1.0 : 0.0) + (s1 + aaff * ds1) * (l1 + aaff * dl1) * (p1 ? 1.0 : 0.0) +
                                          (s2 + aaff * ds2) * (l2 + aaff * dl2) * (p2 ? 1.0 : 0.0) + (s3 + aaff * ds3) * (l3 + aaff * dl3) * (p3 ? 1.0 : 0.0) + ymas);
                const double sig = (mas / wsum) / mu;
                smu = sig * sig * sig * mu;
                a0 = ds0 * dl0; a1 = ds1 * dl1; a2 = ds2 * dl2; a3 = ds3 * dl3;
            } else {
                // fraction to the boundary: 0.9999 if the iterate it leads to stays in the wide neighbourhood
                // min s_i λ_i / w_i >= 0.01 μ, else 0.99 (an unguarded 0.9999 jams about one instance in 50000 into a
                // cycle; all the row state is in registers here, so the test costs two reductions)
                alpha = fmin(1.0, 0.9999 * amin);
                auto pr = [&](bool has, double sv, double ds, double lv, double dl) { return has ? (sv + alpha * ds) * (lv + alpha * dl) : 0.0; };
                const double q0 = pr(p0, s0, ds0, l0, dl0), q1 = pr(p1, s1, ds1, l1, dl1), q2 = pr(p2, s2, ds2, l2, dl2), q3 = pr(p3, s3, ds3, l3, dl3);
                double yps = 0.0, ypm = 1e300;
                mhe::sfor<KYM>([&](auto iq) {
                    constexpr int q = decltype(iq)::v;
                    const double y0_ = pr(yp0[q], ys0[q], yds0[q], yl0[q], ydl0[q]), y1_ = pr(yp1[q], ys1[q], yds1[q], yl1[q], ydl1[q]);
                    yps += y0_ + y1_;
                    ypm = fmin(ypm, fmin(yp0[q] ? y0_ : 1e300, yp1[q] ? y1_ : 1e300));
                });
                const double psum = w.rsum(q0 + q1 + q2 + q3 + yps);
                const double iwt = mhe::recip_fast(wt);         // (w0 = w1 = 1, w2 = w3 = wt)
                const double pmin = w.rmin(fmin(fmin(fmin(p0 ? q0 : 1e300, p1 ? q1 : 1e300), fmin(p2 ? q2 * iwt : 1e300, p3 ? q3 * iwt : 1e300)), ypm));
                if (!(pmin * wsum >= 0.01 * psum)) alpha = fmin(1.0, 0.99 * amin);
            }
        }
        // ---- update (a finished controller of the wavefront keeps its iterate)
        {
            const double al = done ? 0.0 : alpha;
            if (p0) { s0 += al * ds0; l0 += al * dl0; }
            if (p1) { s1 += al * ds1; l1 += al * dl1; }
            if (p2) { s2 += al * ds2; l2 += al * dl2; }
            if (p3) { s3 += al * ds3; l3 += al * dl3; }
            mhe::sfor<KYM>([&](auto iq) {
                constexpr int q = decltype(iq)::v;
                if (yp0[q]) { ys0[q] += al * yds0[q]; yl0[q] += al * ydl0[q]; }
                if (yp1[q]) { ys1[q] += al * yds1[q]; yl1[q] += al * ydl1[q]; }
            });
            const double zm = w.rmax(isdu ? fmax(1.0, fabs(z)) : 1.0), dm = w.rmax(isdu ? fabs(al * dz) : 0.0);
            if (isvar && !done) z += al * dz;
            if (!done) {
                laststep = alpha >= 0.5 ? dm / zm : 1e300;      // (a blocked step says nothing about convergence: Step::run)
                lastscale = 1.0 - alpha;
                if (norows) { st = 0; done = true; it = 0; }
            }
        }
        if (!w.any(!done)) break;
    }
    // never primal-feasible (or NaN) => the reference's error branch (execute.jl:484-489): the shifted warm start
    if (st == 1 && !(rpn <= 1e-6 * nh)) st = 2;
    if (st == 2) z = zws;
    if (io.Yhat0) {          // predict! (transcription.jl:1136-1145): Ŷ0 = Ẽ Z̃ + F
        zv[l] = isdu ? z : 0.0;
        w.sync();
        for (int r = l; r < nY; r += SMALL_RL) {
            const int t = r / ny, a = r - t * ny;
            double acc = Fv[r];
            for (int j = 0; j < Hc && jl(j) <= t; ++j)
                for (int cc = 0; cc < nu; ++cc) acc += Stab[((t - jl(j)) * ny + a) * nu + cc] * zv[j * nu + cc];
            if (live) io.Yhat0[(size_t)b * nY + r] = acc;
        }
    }
    if (live) {
        if (isvar) io.Z[(size_t)b * nZ + l] = z;
        if (l < nu) io.u0[(size_t)b * nu + l] = z + lu[l];          // getinput!: u0 = lastu0 + ΔU[1:nu]
        if (l == 0) {
            io.status[b] = st;
            if (io.iters) io.iters[b] = it;                          // factorisations (0: closed form, no finite row)
            if (io.audit) {
                double* au = io.audit + (size_t)b * 4;
                au[0] = mu_seen; au[1] = rd_seen; au[2] = rpn / nh; au[3] = polished;
            }
        }
    }
}

}  // namespace mpcqp
