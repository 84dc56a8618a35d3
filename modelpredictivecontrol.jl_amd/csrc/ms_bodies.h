// ms_bodies.h -- one control period of a LinMPC with the MultipleShooting transcription, one controller per
// wavefront, stage-structured: the condensed matrices E, H~ are NEVER formed.
//
//   decision vector Z = [dU; X^0(k+1..k+Hp)] (+ eps)          /root/reference/src/controller/transcription.jl:5-7
//   prediction matrices E = [0 diag(C^)], J = diag(D^d)       src/controller/transcription.jl:196-240
//   the model as equality constraints  E_S Z + F_S = 0         src/controller/transcription.jl:303-414, 913-928
//   same objective, bounds, softness, slack as SingleShooting  src/controller/construct.jl:837-845, 999-1234
//   recommended when cond(H~) is large (unstable plants, long horizons)   src/controller/construct.jl:855-866
//
// The QP is solved in its stage form.  With xi_t = [x^0(k+t); u0(k+t-1)] (ns = nx^ + nu) the model is
//     xi_{t+1} = Abar xi_t + Bbar du_t + g_t,   Abar = [A^ B^u; 0 I],  Bbar = [B^u; I],   g_t = [B^d d0(k+t) + f^op - x^op; 0]
// with a free move du_t only at the first step of a move-blocking interval (construct.jl:597-660); every inequality row
// touches ONE stage (input bounds: the u part of xi_{t+1}; output bounds: C^ x^0(k+t+1); terminal bounds: x^0(k+Hp);
// increment bounds: du_t) plus the slack eps.  Interior-point method: the dual-regularised Mehrotra predictor-corrector
// of the condensed kernels (mpcqp_bodies.h: Step::run -- same starting point, row algebra with one reciprocal per row,
// step-length rules and termination test) on the iterate (Z, nu, s, lam) with nu the multipliers of the model equations.
// Every Newton system
//     [Phi A_eq'; A_eq 0] [dZ; nu+] = -[g^; c],   Phi = blkdiag_t(Q_t, R_t) + arrow border of eps
// is solved by a RICCATI recursion over the horizon (block elimination of the KKT matrix in stage order):
//     S = Abar' P_{t+1} Abar,  Lam = R_t + S_uu,  K_t = -Lam^-1 S_u.,  P_t = Q_t + S + S_.u K_t          (backward, matrices)
//     w = P_{t+1} c_t + p_{t+1},  k_t = -Lam^-1 (r_t + (Abar'w)_u),  p_t = q_t + Abar'w + K_t'(...)        (backward, vectors)
//     du_t = K_t dxi_t + k_t,  dxi_{t+1} = Abar dxi_t + Bbar du_t + c_t,  nu+_{t+1} = P_{t+1} dxi_{t+1} + p_{t+1}   (forward)
// (Bbar is the u-column block of Abar, so Bbar'P Abar = S_u. and Bbar'P Bbar = S_uu: one congruence per stage.)  The
// forward sweep runs on the CLOSED-LOOP dynamics Abar + Bbar K_t, which is what keeps the recursion accurate for unstable
// plants where powers of A^ (the entries of the condensed E) overflow float64's digits; the defects c_t of the model
// equations are part of every Newton step, so rounding in X^0 does not accumulate.  The slack couples all soft rows: its
// column phi of Phi goes through the same recursion once per factorisation (psi = -Phi^-1 phi) and deps follows from the
// scalar Schur complement, like the arrow border of the MHE kernel (mhe_bodies.h).
//
// Written against the wave interface W of mpcqp_bodies.h (gfx950: DevWave; tests/emu: 64 host threads).
#pragma once
#include <math.h>

#include "mpcqp_bodies.h"
#include "mpcqp_types.h"

#ifndef MPCQP_MS_LDS_SHARE
#define MPCQP_MS_LDS_SHARE (40 * 1024)   // horizon-long data stay in LDS up to this many bytes per wavefront (160 KB / 4: measured
                                         // on C2 shapes, 24 KB: 29 ms in LDS against 70 ms through the HBM scratch for 8192 controllers)
#endif
#ifndef MPCQP_MS_ADAPT_DELTA
#define MPCQP_MS_ADAPT_DELTA 0      // (experiment: delta / 10 when r_p stalls at small mu -- fewer flagged solves on C3, but randomised family 1 then wanders at mu < 1e-13: off)
#endif
#ifndef MPCQP_MS_POLISH
#define MPCQP_MS_POLISH 1         // active-set polish of the interior-point iterate (MsStep::polish)
#endif
#ifndef MPCQP_MS_POLISH_RHO
#define MPCQP_MS_POLISH_RHO 1e8   // penalty of the polish's method of multipliers.  (Step::polish uses 1e10 on the condensed Phi; the cost-to-go of
                                  // the Riccati sweep sums the penalty over the stages and Lam = R + B'P B hides the O(0.04) curvature of a free move
                                  // behind it: at 1e10 the rounds of randomised family 60 (nu = 4 > ny = 2, Hp = 22) stopped contracting at a dual
                                  // residual of 2e-4; at 1e8 they reach 7e-13 in three rounds, at 1e6 r_A contracts by 7e-3 only.)
#endif
#ifndef MPCQP_MS_POLISH_FACTS
#define MPCQP_MS_POLISH_FACTS 3   // working sets (factorisations) per polish attempt
#endif
#ifndef MPCQP_MS_POLISH_BUDGET
#define MPCQP_MS_POLISH_BUDGET 6  // no new polish attempt once this many polish factorisations are spent
#endif
#ifndef MPCQP_MS_POLISH_RD
#define MPCQP_MS_POLISH_RD 1e-13  // relative dual residual the polished point is accepted at
#endif
#ifndef MPCQP_MS_REFINE
#define MPCQP_MS_REFINE 0         // steps of iterative refinement per Newton solve (MsStep::newton; measured: no gain, see there)
#endif

namespace mpcqp {

// optional outputs of the MultipleShooting step
struct MsIO {
    double* Xhat;     // [B][Hp][nxh]  X^0(k+1..k+Hp) at the optimum (the second block of Z), may be null
    double* defect;   // [B]           max |E_S Z + F_S| at the returned point, may be null
    double* scratch;  // [nslots][big] horizon-long data of the resident wavefronts (null: they live in LDS)
    int nslots;
    int* next;        // work counter of the persistent grid (HBM placement)
};

enum { MS_UMIN = 0, MS_UMAX, MS_DUMIN, MS_DUMAX, MS_YMIN, MS_YMAX, MS_XMIN, MS_XMAX, MS_EPS, MS_NGROUP };
constexpr int MS_NROWARR = 8;     // h, s, lam, rp, gd, pp, cs, wi

struct MsCarve {
    int A, Bu, C;                       // model: A^ (nx,nx) col-major, B^u (nx,nu) col-major, C^ (ny,nx) col-major
    int X, V, DU;                       // iterate: x^0(k+t+1), u0(k+t) for t = 0..Hp-1; free moves
    int NX, NV;                         // multipliers of the model equations (costates nu_{t+1})
    int dX, dV, dDU, nX, nV;            // Newton direction and its nu+
    int pX, pV, pDU, qX, qV;            // psi = -Phi^-1 phi and its nu
    int gX, gV, gDU;                    // gradient of the current solve
    int fX, fV, fDU;                    // border column phi
    int cX, cV;                         // defects c_t
    int eX, eV, eDU, mX, mV, hDU;       // correction of a Newton solve (iterative refinement), its nu; the solve's own g_u
    int gv;                             // g_t (x part), [Hp][nx]
    int ry, ru;                         // targets: C^ x - ry[t] with ry = R^y - D^d d^ (nY); u - ru (nU)
    int QY, QV, RD;                     // stage Hessian diagonals: output weight 2M + D_Y (nY), 2L + D_U (nU), 2N + D_dU (nDU)
    int CX, CD;                         // C^ x of the iterate / of a direction (nY)
    int XT, bDU;                        // D~ of the terminal rows per state (nx); dU kept while the polish runs (nDU)
    int P, K, Li, Lm, pv, kk;               // factor data the sweeps read: P_{t+1} c_t [Hp][ns], K_t [Hc][nu][ns], Lam^-1, Lam [Hc][nu][nu] (pv, kk: unused)
    int S, T, wv, av, Pl, Pl2, Kl, Ll, Lml, pl, xl, kkl, ul, stg;   // stage work in LDS: S, T (ns x ns), w, a (ns), P_{t+1} / P_t, K_t, Lam^-1, Lam, sweep carries
    int x0, lu;                         // x^0(k), u0(k-1)
    int rows[MS_NROWARR];
    int rowoff[MS_NGROUP + 1];
    int jl, ctrl;                       // int tables: first step of block j [Hc+1]; block that starts at step t or -1 [Hp]
    int nrows, total;
    int small, big;                     // doubles of the always-in-LDS block / of the horizon-long data
    bool big_in_lds;
};

MPCQP_HD inline int ms_group_count(const Dims& d, int g) {
    switch (g) {
        case MS_UMIN: case MS_UMAX: return d.nU;
        case MS_DUMIN: case MS_DUMAX: return d.nDU;
        case MS_YMIN: case MS_YMAX: return d.nY;
        case MS_XMIN: case MS_XMAX: return d.nxh;
        default: return 1;
    }
}
MPCQP_HD inline bool ms_group_on(const Dims& d, const Model& m, int g) {
    switch (g) {
        case MS_UMIN: return m.U0min != nullptr;
        case MS_UMAX: return m.U0max != nullptr;
        case MS_DUMIN: return m.DUmin != nullptr;
        case MS_DUMAX: return m.DUmax != nullptr;
        case MS_YMIN: return m.Y0min != nullptr;
        case MS_YMAX: return m.Y0max != nullptr;
        case MS_XMIN: return m.x0min != nullptr;
        case MS_XMAX: return m.x0max != nullptr;
        default: return d.neps != 0;
    }
}

MPCQP_HD inline MsCarve make_ms_carve(const Dims& d, const Model& m) {
    MsCarve c{};
    const int nx = d.nxh, nu = d.nu, ny = d.ny, ns = nx + nu, Hp = d.Hp, Hc = d.Hc;
    const int nX = nx * Hp, nV = nu * Hp, nDU = d.nDU, nY = d.nY, npk = ns * (ns + 1) / 2;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 1) & ~1; return r; };
    // -- always in LDS (offsets from the LDS base): the model, the work matrices of a stage, the horizon tables
    c.A = take(nx * nx); c.Bu = take(nx * nu); c.C = take(ny * nx);
    c.S = take(ns * ns); c.T = take(ns * ns); c.wv = take(ns > ny ? ns : ny); c.av = take(ns);
    c.Pl = take(ns * ns); c.Pl2 = take(ns * ns);   // cost-to-go of the stage in flight, full storage: P_{t+1} (in), P_t (out), ping-pong
    c.Kl = take(nu * ns); c.Ll = take(nu * nu); c.Lml = take(nu * nu);   // gain, Lam^-1, Lam of the stage in flight
    c.pl = take(2 * ns); c.xl = take(2 * ns); c.kkl = take(nDU); c.ul = take(nu > ny ? nu : ny);
    c.stg = take(nu * ns + 2 * nu * nu + nu + 3 * ns + ny);   // read-only inputs of the stage in flight, staged from the horizon-long arrays in one go
    c.x0 = take(nx); c.lu = take(nu);
    c.jl = take((Hc + 2) / 2 + 1);
    c.ctrl = take((Hp + 1) / 2 + 1);
    c.small = o;
    // -- the horizon-long data (offsets from the "big" base): behind the small block in LDS when everything fits a share
    //    of the LDS that keeps eight wavefronts per CU resident, else in a per-wavefront scratch in HBM (make_ms_carve's
    //    `big_in_lds`; the kernels are instantiated for both placements)
    o = 0;
    c.X = take(nX); c.V = take(nV); c.DU = take(nDU);
    c.NX = take(nX); c.NV = take(nV);
    c.dX = take(nX); c.dV = take(nV); c.dDU = take(nDU); c.nX = take(nX); c.nV = take(nV);
    c.pX = take(nX); c.pV = take(nV); c.pDU = take(nDU); c.qX = take(nX); c.qV = take(nV);
    c.gX = take(nX); c.gV = take(nV); c.gDU = take(nDU);
    c.fX = take(nX); c.fV = take(nV); c.fDU = take(nDU);
    c.cX = take(nX); c.cV = take(nV);
    { const int R_ = MPCQP_MS_REFINE ? 1 : 0; c.eX = take(R_ * nX); c.eV = take(R_ * nV); c.eDU = take(R_ * nDU); c.mX = take(R_ * nX); c.mV = take(R_ * nV); c.hDU = take(R_ * nDU); }
    c.gv = take(nX);
    c.ry = take(nY); c.ru = take(nV);
    c.QY = take(nY); c.QV = take(nV); c.RD = take(nDU);
    c.CX = take(nY); c.CD = take(nY);
    c.XT = take(nx); c.bDU = take(nDU);
    c.P = take(Hp * ns);                 // the vectors P_{t+1} c_t of the iterate's defects: all the sweeps need of the cost-to-go
    c.K = take(Hc * nu * ns); c.Li = take(Hc * nu * nu); c.Lm = take(Hc * nu * nu); c.pv = take(0); c.kk = take(0);
    int r = 0;
    for (int g = 0; g < MS_NGROUP; ++g) {
        c.rowoff[g] = r;
        if (ms_group_on(d, m, g)) r += ms_group_count(d, g);
    }
    c.rowoff[MS_NGROUP] = r;
    c.nrows = r;
    for (int a = 0; a < MS_NROWARR; ++a) c.rows[a] = take(r);
    c.big = o;
    c.big_in_lds = (size_t)(c.small + c.big) * sizeof(double) <= MPCQP_MS_LDS_SHARE;
    c.total = c.small + (c.big_in_lds ? c.big : 0);      // doubles of LDS per wavefront
    return c;
}

template <class W>
struct MsStep {
    W& w;
    double* bg;         // base of the horizon-long data: LDS behind the small block, or this wavefront's HBM scratch
    const Dims& d;
    const Model& m;
    const StepIO& io;
    const int b;
    double* sm;
    const MsCarve c;
    const int nx, nu, ny, nd, ns, Hp, Hc, nDU, nY, nXt, nVt, npk;
    int *jlt, *ctrl;
    double *A, *Bu, *Cm;
    double *rh, *rs, *rl, *rrp, *rgd, *rpp, *rcs, *rwi;
    double eps = 0.0, deps = 0.0, delta, nh = 1.0, wsum = 0.0;
    bool use_defect = false;        // the defects of the iterate enter the Newton systems (set by defects())
    double prof_[8] = {0};          // -DMPCQP_MS_PROFILE: cycles per phase (residuals, stage data, factor, psi sweep, newton, update, polish, run)
    int mact = 0;

    // block-diagonal M_Hp (terminal costs, construct.jl:45-93; mpcqp_set_output_weight_blocks): [Hp][ny][ny] symmetric blocks
    // of this controller, or null (diagonal Mdiag)
    MPCQP_HD const double* Mblk_() const { return m.Mblk ? m.Mblk + (size_t)b * d.Hp * d.ny * d.ny : nullptr; }
    MPCQP_HD MsStep(W& w_, const Dims& d_, const Model& m_, const StepIO& io_, int b_, double* sm_, double* big_)
        : w(w_), bg(big_), d(d_), m(m_), io(io_), b(b_), sm(sm_), c(make_ms_carve(d_, m_)), nx(d_.nxh), nu(d_.nu), ny(d_.ny), nd(d_.nd),
          ns(d_.nxh + d_.nu), Hp(d_.Hp), Hc(d_.Hc), nDU(d_.nDU), nY(d_.nY), nXt(d_.nxh * d_.Hp), nVt(d_.nu * d_.Hp),
          npk((d_.nxh + d_.nu) * (d_.nxh + d_.nu + 1) / 2) {
        jlt = reinterpret_cast<int*>(sm + c.jl);
        ctrl = reinterpret_cast<int*>(sm + c.ctrl);
        A = sm + c.A; Bu = sm + c.Bu; Cm = sm + c.C;
        rh = bg + c.rows[0]; rs = bg + c.rows[1]; rl = bg + c.rows[2]; rrp = bg + c.rows[3];
        rgd = bg + c.rows[4]; rpp = bg + c.rows[5]; rcs = bg + c.rows[6]; rwi = bg + c.rows[7];
        delta = d.dual_reg;
    }

    MPCQP_HD static long long clk() {
#if defined(MPCQP_MS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        return clock64();
#else
        return 0;
#endif
    }
    MPCQP_HD bool on(int g) const { return c.rowoff[g + 1] > c.rowoff[g]; }
    MPCQP_HD static int pidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
    MPCQP_HD bool fin(int r) const { return rh[r] < BIG; }

    // softness of row k of group g (reference defaults: 0 for u and du, 1 for y and x^end; construct.jl:909-913)
    MPCQP_HD double softness(int g, int k) const {
        if (!d.neps) return 0.0;
        const double* p = nullptr;
        double def = 0.0;
        size_t n = 0;
        switch (g) {
            case MS_UMIN: p = m.C_umin; n = d.nU; break;
            case MS_UMAX: p = m.C_umax; n = d.nU; break;
            case MS_DUMIN: p = m.C_dumin; n = d.nDU; break;
            case MS_DUMAX: p = m.C_dumax; n = d.nDU; break;
            case MS_YMIN: p = m.C_ymin; n = d.nY; def = 1.0; break;
            case MS_YMAX: p = m.C_ymax; n = d.nY; def = 1.0; break;
            case MS_XMIN: p = m.c_x0min; n = d.nxh; def = 1.0; break;
            case MS_XMAX: p = m.c_x0max; n = d.nxh; def = 1.0; break;
            default: return 0.0;
        }
        return p ? p[(size_t)b * n + k] : def;
    }

    // fn(group, k, row index) for every row slot owned by this lane
    template <class Fn>
    MPCQP_HD void for_rows(Fn fn) {
        for (int g = 0; g < MS_NGROUP; ++g) {
            const int r0 = c.rowoff[g], n = c.rowoff[g + 1] - r0;
            for (int k = w.lane; k < n; k += WAVE) fn(g, k, r0 + k);
        }
    }

    // ---- set-up: model, tables, references, bounds ---------------------------------------------------
    MPCQP_HD void load() {
        const double* gA = m.Ahat + (size_t)b * nx * nx;
        const double* gB = m.Bu + (size_t)b * nx * nu;
        const double* gC = m.C + (size_t)b * ny * nx;
        for (int i = w.lane; i < nx * nx; i += WAVE) A[i] = gA[i];
        for (int i = w.lane; i < nx * nu; i += WAVE) Bu[i] = gB[i];
        for (int i = w.lane; i < ny * nx; i += WAVE) Cm[i] = gC[i];
        for (int i = w.lane; i <= Hc; i += WAVE) jlt[i] = d.default_nb ? (i < Hc ? i : Hp) : (i < Hc ? m.jl[i] : Hp);
        for (int t = w.lane; t < Hp; t += WAVE) ctrl[t] = -1;
        for (int i = w.lane; i < nx; i += WAVE) sm[c.x0 + i] = io.xhat0[(size_t)b * nx + i];
        for (int i = w.lane; i < nu; i += WAVE) sm[c.lu + i] = io.lastu0[(size_t)b * nu + i];
        w.sync();
        if (io.kf_y0m) {
            // fused control period (mpcqp_loop_device, round 6 on this kernel too): the SteadyKalmanFilter correction first
            // (correct_estimate_obsv!, src/estimator/kalman.jl:284-295; same arithmetic order as kf_correct_lane and step_body)
            double* xh = sm + c.x0;
            const double* K = io.kf_K + (size_t)b * io.kf_nym * nx;
            double acc[4];                                     // rows i = lane + 64 q (nx^ <= 256)
            int nq = 0;
            for (int i = w.lane; i < nx; i += WAVE, ++nq) {
                double a_ = xh[i];
                for (int mm = 0; mm < io.kf_nym; ++mm) {
                    const int a = io.kf_iym[mm];
                    double v = io.kf_y0m[(size_t)b * io.kf_nym + mm];
                    for (int k = 0; k < nx; ++k) v -= gC[a + ny * k] * xh[k];
                    for (int e = 0; e < nd; ++e) v -= m.Dd[(size_t)b * ny * nd + a + ny * e] * io.d0[(size_t)b * nd + e];
                    a_ += K[i + nx * mm] * v;
                }
                acc[nq] = a_;
            }
            w.sync();
            nq = 0;
            for (int i = w.lane; i < nx; i += WAVE, ++nq) {
                xh[i] = acc[nq];
                if (!io.kf_predict) io.xhat0_out[(size_t)b * nx + i] = acc[nq];
            }
            w.sync();
        }
        for (int j = w.lane; j < Hc; j += WAVE) ctrl[jlt[j]] = j;
        // g_t = B^d d0(k+t) + (f^op - x^op): d0(k) for t = 0, D^0 block t-1 after (transcription.jl:386-389)
        for (int i = w.lane; i < nXt; i += WAVE) {
            const int t = i / nx, r = i - t * nx;
            double acc = m.dop ? m.dop[(size_t)b * nx + r] : 0.0;
            for (int e = 0; e < nd; ++e) {
                const double de = t == 0 ? io.d0[(size_t)b * nd + e] : io.Dhat0[(size_t)b * d.nD + (t - 1) * nd + e];
                acc += m.Bd[(size_t)b * nx * nd + r + nx * e] * de;
            }
            bg[c.gv + i] = acc;
        }
        // output target of stage t: R^y - Yop - D^d d^0(k+t+1)  (F = J D^0, transcription.jl:232; execute.jl:262-266)
        const bool rconst = d.flags & 1u;
        for (int r = w.lane; r < nY; r += WAVE) {
            const int t = r / ny, a = r - t * ny;
            double acc = rconst ? io.Ry[(size_t)b * ny + a] : io.Ry[(size_t)b * nY + r];
            for (int e = 0; e < nd; ++e) acc -= m.Dd[(size_t)b * ny * nd + a + ny * e] * io.Dhat0[(size_t)b * d.nD + t * nd + e];
            bg[c.ry + r] = acc;
        }
        for (int r = w.lane; r < nVt; r += WAVE) bg[c.ru + r] = io.Ru ? io.Ru[(size_t)b * d.nU + r] : 0.0;
        w.sync();
    }

    // bounds of the rows.  A row acts on the stage variable itself, so (unlike the condensed form) its right-hand side
    // is the bound: U0min/U0max on u0(k+t); Y0min/Y0max - D^d d^0 on C^ x^0(k+t+1); x^0min/x^0max on x^0(k+Hp).
    MPCQP_HD void build_rows() {
        int cnt = 0;
        double hmax = 0.0;
        for_rows([&](int g, int k, int r) {
            double bound = INFINITY;
            switch (g) {
                case MS_UMIN: bound = -m.U0min[(size_t)b * d.nU + k]; break;
                case MS_UMAX: bound = m.U0max[(size_t)b * d.nU + k]; break;
                case MS_DUMIN: bound = -m.DUmin[(size_t)b * nDU + k]; break;
                case MS_DUMAX: bound = m.DUmax[(size_t)b * nDU + k]; break;
                case MS_YMIN:
                case MS_YMAX: {
                    const int t = k / ny, a = k - t * ny;
                    double dterm = 0.0;
                    for (int e = 0; e < nd; ++e) dterm += m.Dd[(size_t)b * ny * nd + a + ny * e] * io.Dhat0[(size_t)b * d.nD + t * nd + e];
                    bound = g == MS_YMIN ? -(m.Y0min[(size_t)b * nY + k] - dterm) : m.Y0max[(size_t)b * nY + k] - dterm;
                    break;
                }
                case MS_XMIN: bound = -m.x0min[(size_t)b * nx + k]; break;
                case MS_XMAX: bound = m.x0max[(size_t)b * nx + k]; break;
                default: bound = 0.0; break;          // -eps <= 0
            }
            const bool ok = fabs(bound) < BIG && bound == bound;
            rh[r] = ok ? bound : 2.0 * BIG;
            rs[r] = 1.0; rl[r] = ok ? 1.0 : 0.0; rrp[r] = 0.0; rgd[r] = 0.0; rpp[r] = 0.0; rwi[r] = 0.0;
            rcs[r] = softness(g, k);
            if (ok) { ++cnt; hmax = fmax(hmax, fabs(bound)); }
        });
        mact = w.isum(cnt);
        wsum = (double)mact;
        nh = 1.0 + w.maxv(hmax);
        w.sync();
    }

    // ---- model operators ---------------------------------------------------------------------------------
    // state of stage t (t = -1: the given x^0(k), u0(k-1); directions: zero)
    MPCQP_HD double xat(const double* Xv, int t, int i, bool dir) const { return t >= 0 ? Xv[t * nx + i] : (dir ? 0.0 : sm[c.x0 + i]); }
    MPCQP_HD double vat(const double* Vv, int t, int cc, bool dir) const { return t >= 0 ? Vv[t * nu + cc] : (dir ? 0.0 : sm[c.lu + cc]); }

    // X, V <- the model rolled out from DU (sequential over the horizon, lanes over the state)
    MPCQP_HD void rollout(const double* DU, double* Xv, double* Vv) {
        for (int t = 0; t < Hp; ++t) {
            const int j = ctrl[t];
            for (int cc = w.lane; cc < nu; cc += WAVE) Vv[t * nu + cc] = vat(Vv, t - 1, cc, false) + (j >= 0 ? DU[j * nu + cc] : 0.0);
            w.sync();
            for (int i = w.lane; i < nx; i += WAVE) {
                double acc = bg[c.gv + t * nx + i];
                for (int k = 0; k < nx; ++k) acc += A[i + nx * k] * xat(Xv, t - 1, k, false);
                for (int cc = 0; cc < nu; ++cc) acc += Bu[i + nx * cc] * Vv[t * nu + cc];
                Xv[t * nx + i] = acc;
            }
            w.sync();
        }
    }

    // out[(t,a)] = C^ Xv_t
    MPCQP_HD void C_apply(const double* Xv, double* out) {
        for (int r = w.lane; r < nY; r += WAVE) {
            const int t = r / ny, a = r - t * ny;
            double acc = 0.0;
            for (int k = 0; k < nx; ++k) acc += Cm[a + ny * k] * Xv[t * nx + k];
            out[r] = acc;
        }
        w.sync();
    }

    // (G z)[row] without the slack column, from the stage variables (CXv = C^ Xv)
    MPCQP_HD double prim(int g, int k, const double* Xv, const double* Vv, const double* DUv, const double* CXv, double e) const {
        switch (g) {
            case MS_UMIN: return -Vv[k];
            case MS_UMAX: return Vv[k];
            case MS_DUMIN: return -DUv[k];
            case MS_DUMAX: return DUv[k];
            case MS_YMIN: return -CXv[k];
            case MS_YMAX: return CXv[k];
            case MS_XMIN: return -Xv[(Hp - 1) * nx + k];
            case MS_XMAX: return Xv[(Hp - 1) * nx + k];
            default: return -e;
        }
    }

    // gX, gV, gDU (+ returned slack entry) <- G' wv(row) summed onto `base` gradients (which may be null = 0):
    // wv evaluated on finite rows.  Output arrays are OVERWRITTEN.
    template <class Fn>
    MPCQP_HD double Gt_apply(Fn wv, double* oX, double* oV, double* oDU, bool with_cost) {
        // per-row weights into rgd (scratch), slack entry accumulated
        double eacc = 0.0;
        for_rows([&](int g, int k, int r) {
            double v = 0.0;
            if (fin(r)) { v = wv(r); eacc -= (g == MS_EPS ? 1.0 : rcs[r]) * v; }
            rgd[r] = v;
        });
        eacc = w.sum(eacc);
        w.sync();
        auto rowv = [&](int g, int k) { return on(g) ? rgd[c.rowoff[g] + k] : 0.0; };
        // outputs: C^'(tYmax - tYmin) (+ cost gradient 2 M (C^ x - ry))
        for (int i = w.lane; i < nXt; i += WAVE) {
            const int t = i / nx, k = i - t * nx;
            double acc = 0.0;
            for (int a = 0; a < ny; ++a) {
                const int r = t * ny + a;
                double ty = rowv(MS_YMAX, r) - rowv(MS_YMIN, r);
                if (with_cost) {
                    if (const double* Mb = Mblk_()) {
                        for (int a2 = 0; a2 < ny; ++a2)
                            ty += 2.0 * Mb[((size_t)t * ny + a2) * ny + a] * (bg[c.CX + t * ny + a2] - bg[c.ry + t * ny + a2]);
                    } else {
                        ty += 2.0 * m.Mdiag[(size_t)b * nY + r] * (bg[c.CX + r] - bg[c.ry + r]);
                    }
                }
                acc += Cm[a + ny * k] * ty;
            }
            if (t == Hp - 1) acc += rowv(MS_XMAX, k) - rowv(MS_XMIN, k);
            oX[i] = acc;
        }
        for (int i = w.lane; i < nVt; i += WAVE) {
            double acc = rowv(MS_UMAX, i) - rowv(MS_UMIN, i);
            if (with_cost) acc += 2.0 * m.Ldiag[(size_t)b * d.nU + i] * (bg[c.V + i] - bg[c.ru + i]);
            oV[i] = acc;
        }
        for (int i = w.lane; i < nDU; i += WAVE) {
            double acc = rowv(MS_DUMAX, i) - rowv(MS_DUMIN, i);
            if (with_cost) acc += 2.0 * m.Ndiag[(size_t)b * nDU + i] * bg[c.DU + i];
            oDU[i] = acc;
        }
        w.sync();
        return eacc;
    }

    // ---- Riccati factorisation of the current Phi (QY, QV, RD hold the stage diagonals) --------------------
    // store(i, j, sum_k a(i, k) b(k, j)) for i < M, j < N (lower: only j <= i).  On the device the product runs on the matrix
    // cores, v_mfma_f64_16x16x4 per 16 x 16 tile and four k: A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15],
    // D[row = (lane >> 4) + 4 reg][col = lane & 15] -- a lane fetches ONE a and ONE b per 1024 multiply-adds.  The scalar
    // form (one output per lane, two LDS reads per multiply-add) made the factorisation LDS-bound: eight wavefronts per CU
    // share one LDS pipe, 65 k cycles per stage on C3 shapes.  The emulator runs the scalar form with the same accessors.
    template <class FA, class FB, class FS>
    MPCQP_HD void mm(int M, int N, int K, bool lower, FA a, FB b, FS store) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef double v4d_ __attribute__((ext_vector_type(4)));
        const int li = w.lane & 15, lk = w.lane >> 4;
        for (int I = 0; I < (M + 15) / 16; ++I) {
            for (int J = 0; J < (N + 15) / 16; ++J) {
                if (lower && J > I) continue;
                const int i = 16 * I + li, j = 16 * J + li;
                v4d_ acc = {0.0, 0.0, 0.0, 0.0};
                // (requesting the operands of four k steps together before their MFMAs: measured, no gain, more spills)
                for (int k0 = 0; k0 < K; k0 += 4) {
                    const int k = k0 + lk;
                    const double av = (i < M && k < K) ? a(i, k) : 0.0;
                    const double bv = (j < N && k < K) ? b(k, j) : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
                for (int r = 0; r < 4; ++r) {
                    const int ii = 16 * I + lk + 4 * r;
                    if (ii < M && j < N && (!lower || j <= ii)) store(ii, j, acc[r]);
                }
            }
        }
#else
        for (int idx = w.lane; idx < M * N; idx += WAVE) {
            const int i = idx / N, j = idx - i * N;
            if (lower && j > i) continue;
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += a(i, k) * b(k, j);
            store(i, j, acc);
        }
#endif
    }
    // entries of Abar = [A^ B^u; 0 I] and Bbar = [B^u; I]
    MPCQP_HD double Abar(int i, int j) const { return i < nx ? (j < nx ? A[i + nx * j] : Bu[i + nx * (j - nx)]) : (i == j ? 1.0 : 0.0); }
    MPCQP_HD double Bbar(int i, int e) const { return i < nx ? Bu[i + nx * e] : (i - nx == e ? 1.0 : 0.0); }
    // T = Pn M for the packed symmetric Pn (cost-to-go of stage t+1) and M = Abar (Mfull == nullptr: the structure
    // Abar = [A^ B^u; 0 I] is used) or a full ns x ns matrix (the closed-loop matrix Abar + Bbar K)
    MPCQP_HD void PtimesM(const double* Pn, const double* Mfull) {
        double* T = sm + c.T;
        for (int idx = w.lane; idx < ns * ns; idx += WAVE) {
            const int i = idx / ns, j = idx - i * ns;
            double acc = 0.0;
            if (Mfull) {
                for (int k = 0; k < ns; ++k) acc += Pn[i * ns + k] * Mfull[k * ns + j];
            } else if (j < nx) {
                for (int k = 0; k < nx; ++k) acc += Pn[i * ns + k] * A[k + nx * j];
            } else {
                const int cc = j - nx;
                for (int k = 0; k < nx; ++k) acc += Pn[i * ns + k] * Bu[k + nx * cc];
                acc += Pn[i * ns + j];
            }
            T[idx] = acc;
        }
        w.sync_lds();
    }
    // row i of Abar' T (or Mfull' T): entry (i, j)
    MPCQP_HD double MtT(const double* Mfull, int i, int j) const {
        const double* T = sm + c.T;
        double acc = 0.0;
        if (Mfull) {
            for (int k = 0; k < ns; ++k) acc += Mfull[k * ns + i] * T[k * ns + j];
        } else if (i < nx) {
            for (int k = 0; k < nx; ++k) acc += A[k + nx * i] * T[k * ns + j];
        } else {
            const int cc = i - nx;
            for (int k = 0; k < nx; ++k) acc += Bu[k + nx * cc] * T[k * ns + j];
            acc += T[i * ns + j];
        }
        return acc;
    }
    MPCQP_HD static void unpack_low(int idx, int& i, int& j) {
        i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > idx) --i;
        while ((i + 1) * (i + 2) / 2 <= idx) ++i;
        j = idx - i * (i + 1) / 2;
    }

    // Q_t of stage t (0-based: the stage that holds x^0(k+t+1), u0(k+t)) added to the packed Pt
    MPCQP_HD void add_Q(double* Pt, int t) {
        for (int idx = w.lane; idx < ns * ns; idx += WAVE) {
            const int i = idx / ns, j = idx - i * ns;
            double acc = 0.0;
            if (i < nx && j < nx) {           // C^' (diag(QY_t) [+ 2 M_t, block weights]) C^  (+ terminal rows on the last stage)
                for (int a = 0; a < ny; ++a) acc += Cm[a + ny * i] * bg[c.QY + t * ny + a] * Cm[a + ny * j];
                if (const double* Mb = Mblk_()) {
                    for (int a = 0; a < ny; ++a) {
                        double ms = 0.0;
                        for (int a2 = 0; a2 < ny; ++a2) ms += Mb[((size_t)t * ny + a2) * ny + a] * Cm[a2 + ny * j];
                        acc += 2.0 * Cm[a + ny * i] * ms;
                    }
                }
                if (t == Hp - 1 && i == j) acc += xterm(i);
            } else if (i == j) {
                acc = bg[c.QV + t * nu + (i - nx)];
            }
            Pt[idx] += acc;
        }
    }
    // D~ of the terminal rows of state i (filled by assemble())
    MPCQP_HD double xterm(int i) const { return bg[c.XT + i]; }

    // in-place inverse of the symmetric positive definite nu x nu matrix M (Gauss-Jordan, lanes over the entries);
    // returns false on a non-positive pivot (wave-uniform)
    MPCQP_HD bool invert_spd(double* M) {
        bool ok = true;
        for (int p = 0; p < nu; ++p) {
            const double piv = M[p * nu + p];
            w.sync_lds();
            if (!(piv > 0.0)) ok = false;
            const double ip = 1.0 / (piv > 0.0 ? piv : 1.0);
            // row p scaled, others eliminated; the pivot column becomes the inverse's column
            double upd[1];
            (void)upd;
            for (int idx = w.lane; idx < nu * nu; idx += WAVE) {
                const int i = idx / nu, j = idx - i * nu;
                const double mip = M[i * nu + p], mpj = M[p * nu + j];
                double v;
                if (i == p && j == p) v = ip;
                else if (i == p) v = mpj * ip;
                else if (j == p) v = -mip * ip;
                else v = M[idx] - mip * mpj * ip;
                sm[c.T + idx] = v;
            }
            w.sync_lds();
            for (int idx = w.lane; idx < nu * nu; idx += WAVE) M[idx] = sm[c.T + idx];
            w.sync_lds();
        }
        return ok;
    }

    // Backward matrix sweep.  The cost-to-go is propagated in the SYMMETRIC (Joseph) form
    //     P_t = Q_t + (Abar + Bbar K_t)' P_{t+1} (Abar + Bbar K_t) + K_t' R_t K_t,
    // a sum of positive semidefinite terms: the textbook form Q + S - S_.u Lam^-1 S_u. subtracts 1e12-size numbers (rows
    // held at D~ = 1/delta on the input part of the state) from each other and loses the definiteness of P within a few
    // stages (met on the GPU: randomised family 1, a pivot of Lam <= 0 at mu = 3e-7).  One product more per free move.
    // All data of the stage in flight -- P_{t+1}, P_t, K_t, Lam, Lam^-1 -- live in LDS and are fenced with sync_lds(); what the
    // sweeps need later (P_t, K_t, Lam_t, Lam_t^-1) is streamed to the horizon-long arrays without waiting for the stores
    // (one full fence at the end): with the HBM placement a stage costs its LDS work, not a store round trip per fence.
    MPCQP_HD bool factor() {
        double* Pc = bg + c.P;          // [Hp][ns]: P_{t+1} c_t
        double* S = sm + c.S;
        double* Pa = sm + c.Pl;        // P_{t+1}
        double* Pb = sm + c.Pl2;       // P_t
        double* K = sm + c.Kl;
        double* Li = sm + c.Ll;
        double* Lm = sm + c.Lml;
        bool ok = true;
        for (int i = w.lane; i < ns * ns; i += WAVE) Pa[i] = 0.0;
        w.sync_lds();
        add_Q(Pa, Hp - 1);
        w.sync_lds();
        for (int t = Hp - 1; t >= 0; --t) {
            // stage t maps xi_t (stored at t-1; given for t = 0) to xi_{t+1} (stored at t)
            const int j = ctrl[t];
            // P_{t+1} c_t for the sweeps (c: the defects of the iterate, the same for every solve of this iteration)
            for (int i = w.lane; i < ns; i += WAVE) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc += Pa[i * ns + k] * bg[c.cX + t * nx + k];
                for (int cc = 0; cc < nu; ++cc) acc += Pa[i * ns + nx + cc] * bg[c.cV + t * nu + cc];
                Pc[t * ns + i] = acc;
            }
            {   // T = P_{t+1} Abar
                double* T = sm + c.T;
                const double* Pq = Pa;
                mm(ns, ns, ns, false, [&](int i, int k) { return Pq[i * ns + k]; }, [&](int k, int jj) { return Abar(k, jj); },
                   [&](int i, int jj, double v) { T[i * ns + jj] = v; });
                w.sync_lds();
            }
            if (j >= 0) {
                // S_u. = Bbar' T (rows nx.. of Abar' T) into S[0 .. nu*ns); Lam = R + S_uu
                {
                    const double* T = sm + c.T;
                    mm(nu, ns, ns, false, [&](int a, int k) { return Bbar(k, a); }, [&](int k, int col) { return T[k * ns + col]; },
                       [&](int a, int col, double v) { S[a * ns + col] = v; });
                }
                w.sync_lds();
                for (int idx = w.lane; idx < nu * nu; idx += WAVE) {
                    const int a = idx / nu, e = idx - a * nu;
                    // (symmetrised: the two triangles of Bbar'P Bbar differ by rounding)
                    const double v = 0.5 * (S[a * ns + nx + e] + S[e * ns + nx + a]) + (a == e ? bg[c.RD + j * nu + a] : 0.0);
                    Li[idx] = v;
                    Lm[idx] = v;
                }
                w.sync_lds();
                ok = invert_spd(Li) && ok;
                for (int idx = w.lane; idx < nu * ns; idx += WAVE) {          // K = -Lam^-1 S_u.
                    const int a = idx / ns, col = idx - a * ns;
                    double acc = 0.0;
                    for (int e = 0; e < nu; ++e) acc += Li[a * nu + e] * S[e * ns + col];
                    K[idx] = -acc;
                }
                w.sync_lds();
                // one refinement step of the gain: K -= Lam^-1 (S_u. + Lam K)  (the explicit inverse of a Lam with 1/delta-size
                // entries next to O(0.1) ones leaves eps cond(Lam) in K; a row held at D~ = 1/delta multiplies that by 1e12)
                for (int idx = w.lane; idx < nu * ns; idx += WAVE) {
                    const int a = idx / ns, col = idx - a * ns;
                    double acc = S[a * ns + col];
                    for (int e = 0; e < nu; ++e) acc += Lm[a * nu + e] * K[e * ns + col];
                    sm[c.T + idx] = acc;
                }
                w.sync_lds();
                for (int idx = w.lane; idx < nu * ns; idx += WAVE) {
                    const int a = idx / ns, col = idx - a * ns;
                    double acc = 0.0;
                    for (int e = 0; e < nu; ++e) acc += Li[a * nu + e] * sm[c.T + e * ns + col];
                    K[idx] -= acc;
                }
                w.sync_lds();
                // the sweeps' copies
                for (int idx = w.lane; idx < nu * ns; idx += WAVE) bg[c.K + (size_t)j * nu * ns + idx] = K[idx];
                for (int idx = w.lane; idx < nu * nu; idx += WAVE) {
                    bg[c.Li + (size_t)j * nu * nu + idx] = Li[idx];
                    bg[c.Lm + (size_t)j * nu * nu + idx] = Lm[idx];
                }
            }
            if (t == 0) break;                     // xi_0 is data: no cost-to-go needed
            if (j >= 0) {
                // closed-loop matrix Acl = Abar + Bbar K into S (full ns x ns), T = P_{t+1} Acl, P_t = Acl' T + K' R K
                for (int idx = w.lane; idx < ns * ns; idx += WAVE) {
                    const int i = idx / ns, col = idx - i * ns;
                    double acc;
                    if (i < nx) {
                        acc = col < nx ? A[i + nx * col] : Bu[i + nx * (col - nx)];
                        for (int e = 0; e < nu; ++e) acc += Bu[i + nx * e] * K[e * ns + col];
                    } else {
                        acc = (col == i ? 1.0 : 0.0) + K[(i - nx) * ns + col];
                    }
                    S[idx] = acc;
                }
                w.sync_lds();
                {
                    double* T = sm + c.T;
                    const double* Pq = Pa;
                    double* Pn = Pb;
                    mm(ns, ns, ns, false, [&](int i, int k) { return Pq[i * ns + k]; }, [&](int k, int jj) { return S[k * ns + jj]; },
                       [&](int i, int jj, double v) { T[i * ns + jj] = v; });
                    w.sync_lds();
                    // P_t = Acl' T + K' R K: the inner index runs over the ns rows of Acl / T, then over the nu rows of K
                    const double* Rd = bg + c.RD + j * nu;
                    for (int e = w.lane; e < nu; e += WAVE) sm[c.wv + e] = Rd[e];         // (R of the stage into LDS: read per k)
                    w.sync_lds();
                    const double* Rl = sm + c.wv;
                    mm(ns, ns, ns + nu, true,
                       [&](int i, int k) { return k < ns ? S[k * ns + i] : K[(k - ns) * ns + i] * Rl[k - ns]; },
                       [&](int k, int jj) { return k < ns ? T[k * ns + jj] : K[(k - ns) * ns + jj]; },
                       [&](int i, int jj, double v) { Pn[i * ns + jj] = v; Pn[jj * ns + i] = v; });
                }
            } else {
                {
                    const double* T = sm + c.T;
                    double* Pn = Pb;
                    mm(ns, ns, ns, true, [&](int i, int k) { return Abar(k, i); }, [&](int k, int jj) { return T[k * ns + jj]; },
                       [&](int i, int jj, double v) { Pn[i * ns + jj] = v; Pn[jj * ns + i] = v; });
                }
            }
            w.sync_lds();
            add_Q(Pb, t - 1);
            w.sync_lds();
            { double* t_ = Pa; Pa = Pb; Pb = t_; }
        }
        w.sync();                                  // the streamed copies are visible to the sweeps
        return ok;
    }

    // ---- one solve with the current factor -----------------------------------------------------------------
    // minimise 1/2 dz'Phi dz + g'dz  s.t.  dxi_{t+1} = Abar dxi_t + Bbar du_t + c_t   (c = 0 when !defect)
    // in: gX, gV (stage gradients), gDU; out: oX, oV, oDU (the step), nuX, nuV (multipliers nu+)
    MPCQP_HD void sweep(const double* gX, const double* gV, const double* gDU, bool defect,
                        double* oX, double* oV, double* oDU, double* nuX, double* nuV) {
        const double* Pc = bg + c.P;
        double* wv = sm + c.wv;
        double* av = sm + c.av;
        double* kk = sm + c.kkl;
        double* ul = sm + c.ul;
        // The carries of the three passes (p_t, dxi_t, nu_t: ns doubles, ping-pong) live in LDS and are fenced with
        // sync_lds(); the horizon-long arrays are read-only inside a pass or written without being read back in it.  The
        // read-only inputs of a stage (gain, Lam, Lam^-1, gradients, defects) are STAGED into LDS at the top of the stage
        // with all their loads in flight at once: a stage then pays one round trip to the horizon-long arrays (HBM scratch
        // placement: ~2 us) instead of one per dependent step (measured before: 28k cycles per stage of a sweep).
        double* stg = sm + c.stg;
        double* sK = stg;                      // nu x ns
        double* sLi = sK + nu * ns;            // nu x nu
        double* sLm = sLi + nu * nu;           // nu x nu
        double* sgu = sLm + nu * nu;           // nu
        double* sv0 = sgu + nu;                // ns
        double* sv1 = sv0 + ns;                // ns
        double* sv2 = sv1 + ns;                // ns + ny
        auto copy = [&](double* dst, const double* src, int n) { for (int i = w.lane; i < n; i += WAVE) dst[i] = src[i]; };
        double* pa = sm + c.pl;
        double* pb = pa + ns;
        // backward: p_t for t = Hp..1, k_j
        for (int i = w.lane; i < ns; i += WAVE) pa[i] = i < nx ? gX[(Hp - 1) * nx + i] : gV[(Hp - 1) * nu + i - nx];
        w.sync_lds();
        for (int t = Hp - 1; t >= 0; --t) {
            const int j = ctrl[t];
            if (j >= 0) {
                copy(sK, bg + c.K + (size_t)j * nu * ns, nu * ns);
                copy(sLi, bg + c.Li + (size_t)j * nu * nu, nu * nu);
                copy(sLm, bg + c.Lm + (size_t)j * nu * nu, nu * nu);
                copy(sgu, gDU + j * nu, nu);
            }
            if (defect) copy(sv0, Pc + t * ns, ns);
            if (t > 0) { copy(sv1, gX + (t - 1) * nx, nx); copy(sv1 + nx, gV + (t - 1) * nu, nu); }
            w.sync_lds();
            // w = P_{t+1} c_t + p_{t+1}
            for (int i = w.lane; i < ns; i += WAVE) wv[i] = pa[i] + (defect ? sv0[i] : 0.0);
            w.sync_lds();
            // a = Abar' w
            for (int i = w.lane; i < ns; i += WAVE) {
                double acc = 0.0;
                if (i < nx) {
                    for (int k = 0; k < nx; ++k) acc += A[k + nx * i] * wv[k];
                } else {
                    for (int k = 0; k < nx; ++k) acc += Bu[k + nx * (i - nx)] * wv[k];
                    acc += wv[i];
                }
                av[i] = acc;
            }
            w.sync_lds();
            if (j >= 0) {
                // k_j = -Lam^-1 (g_u + (Abar'w)_u), refined once: k -= Lam^-1 (h + Lam k)
                for (int a = w.lane; a < nu; a += WAVE) {
                    double acc = 0.0;
                    for (int e = 0; e < nu; ++e) acc += sLi[a * nu + e] * (sgu[e] + av[nx + e]);
                    kk[j * nu + a] = -acc;
                }
                w.sync_lds();
                for (int a = w.lane; a < nu; a += WAVE) {
                    double acc = sgu[a] + av[nx + a];
                    for (int e = 0; e < nu; ++e) acc += sLm[a * nu + e] * kk[j * nu + e];
                    ul[a] = acc;
                }
                w.sync_lds();
                for (int a = w.lane; a < nu; a += WAVE) {
                    double acc = 0.0;
                    for (int e = 0; e < nu; ++e) acc += sLi[a * nu + e] * ul[e];
                    kk[j * nu + a] -= acc;
                }
                w.sync_lds();
            }
            if (t == 0) break;
            // p_t = g_xi[t] + a + K'(g_u + a_u)
            for (int i = w.lane; i < ns; i += WAVE) {
                double acc = sv1[i] + av[i];
                if (j >= 0)
                    for (int e = 0; e < nu; ++e) acc += sK[e * ns + i] * (sgu[e] + av[nx + e]);
                pb[i] = acc;
            }
            w.sync_lds();
            { double* t_ = pa; pa = pb; pb = t_; }
        }
        // forward: dxi_t carried in LDS (xa = dxi_t, xb = dxi_{t+1}); dxi_0 = 0
        double* xa = sm + c.xl;
        double* xb = xa + ns;
        for (int i = w.lane; i < ns; i += WAVE) xa[i] = 0.0;
        w.sync_lds();
        for (int t = 0; t < Hp; ++t) {
            const int j = ctrl[t];
            if (j >= 0 && t > 0) copy(sK, bg + c.K + (size_t)j * nu * ns, nu * ns);
            if (defect) { copy(sv0, bg + c.cX + t * nx, nx); copy(sv0 + nx, bg + c.cV + t * nu, nu); }
            w.sync_lds();
            if (j >= 0) {
                for (int a = w.lane; a < nu; a += WAVE) {
                    double acc = kk[j * nu + a];
                    if (t > 0)
                        for (int k = 0; k < ns; ++k) acc += sK[a * ns + k] * xa[k];
                    ul[a] = acc;
                    oDU[j * nu + a] = acc;
                }
                w.sync_lds();
            }
            for (int cc = w.lane; cc < nu; cc += WAVE) {
                const double v = xa[nx + cc] + (j >= 0 ? ul[cc] : 0.0) + (defect ? sv0[nx + cc] : 0.0);
                xb[nx + cc] = v;
                oV[t * nu + cc] = v;
            }
            w.sync_lds();
            // dx_{t+1} = A^ dx_t + B^u (dv_t + du_t) + c_x  (dv_t + du_t = dv_{t+1} - c_v)
            for (int i = w.lane; i < nx; i += WAVE) {
                double acc = defect ? sv0[i] : 0.0;
                for (int k = 0; k < nx; ++k) acc += A[i + nx * k] * xa[k];
                for (int cc = 0; cc < nu; ++cc) acc += Bu[i + nx * cc] * (xb[nx + cc] - (defect ? sv0[nx + cc] : 0.0));
                xb[i] = acc;
                oX[t * nx + i] = acc;
            }
            w.sync_lds();
            { double* t_ = xa; xa = xb; xb = t_; }
        }
        w.sync();                                  // oX, oV are read back by the adjoint pass
        // multipliers of the model equations by the ADJOINT recursion of the Newton system's state rows,
        //     nu+_t = g_t + Phi_tt dxi_t + Abar' nu+_{t+1}            (nu+_{Hp+1} = 0),
        // instead of nu+_t = P_t dxi_t + p_t: P carries the 1e12-size barrier weights of rows held at D~ = 1/delta, and the
        // product with a 1e-12-size step leaves O(1) noise in nu.  Built this way the state rows of the Newton system hold
        // exactly and whatever error the recursion made shows up in its control rows, where the next Newton step removes it.
        double* na = sm + c.pl;        // nu_{t+1} (zero beyond the horizon)
        double* nb = na + ns;
        for (int i = w.lane; i < ns; i += WAVE) na[i] = 0.0;
        w.sync_lds();
        for (int t = Hp - 1; t >= 0; --t) {
            copy(sv0, oX + t * nx, nx); copy(sv0 + nx, oV + t * nu, nu);               // dxi_{t+1}
            copy(sv1, gX + t * nx, nx); copy(sv1 + nx, gV + t * nu, nu);               // g_{t+1}
            copy(sv2, bg + c.QV + t * nu, nu); copy(sv2 + nu, bg + c.QY + t * ny, ny);  // stage Hessian diagonals
            w.sync_lds();
            if (const double* Mb = Mblk_()) {        // (2 M_t + diag) (C^ dx): C^ dx first, then the block product
                for (int a = w.lane; a < ny; a += WAVE) {
                    double acc = 0.0;
                    for (int k = 0; k < nx; ++k) acc += Cm[a + ny * k] * sv0[k];
                    ul[a] = acc;
                }
                w.sync_lds();
                double nv[4];                            // rows a = lane + 64 q (ny <= 256)
                int nq = 0;
                for (int a = w.lane; a < ny; a += WAVE, ++nq) {
                    double acc = ul[a] * sv2[nu + a];
                    for (int a2 = 0; a2 < ny; ++a2) acc += 2.0 * Mb[((size_t)t * ny + a2) * ny + a] * ul[a2];
                    nv[nq] = acc;
                }
                w.sync_lds();
                nq = 0;
                for (int a = w.lane; a < ny; a += WAVE, ++nq) ul[a] = nv[nq];
            } else {
                for (int a = w.lane; a < ny; a += WAVE) {
                    double acc = 0.0;
                    for (int k = 0; k < nx; ++k) acc += Cm[a + ny * k] * sv0[k];
                    ul[a] = acc * sv2[nu + a];
                }
            }
            w.sync_lds();
            for (int i = w.lane; i < ns; i += WAVE) {
                double acc;
                if (i < nx) {
                    acc = sv1[i];
                    for (int a = 0; a < ny; ++a) acc += Cm[a + ny * i] * ul[a];
                    if (t == Hp - 1) acc += xterm(i) * sv0[i];
                    for (int k = 0; k < nx; ++k) acc += A[k + nx * i] * na[k];
                    nuX[t * nx + i] = acc;
                } else {
                    const int cc = i - nx;
                    acc = sv1[i] + sv2[cc] * sv0[i];
                    for (int k = 0; k < nx; ++k) acc += Bu[k + nx * cc] * na[k];
                    acc += na[i];
                    nuV[t * nu + cc] = acc;
                }
                nb[i] = acc;
            }
            w.sync_lds();
            { double* t_ = na; na = nb; nb = t_; }
        }
        w.sync();
    }

    MPCQP_HD double dot_z(const double* aX, const double* aV, const double* aDU, const double* bX, const double* bV, const double* bDU) {
        double acc = 0.0;
        for (int i = w.lane; i < nXt; i += WAVE) acc += aX[i] * bX[i];
        for (int i = w.lane; i < nVt; i += WAVE) acc += aV[i] * bV[i];
        for (int i = w.lane; i < nDU; i += WAVE) acc += aDU[i] * bDU[i];
        return w.sum(acc);
    }

    // ---- residuals of the iterate: r_p (rows), mu, defects c, dual residual with the current nu ----------
    MPCQP_HD void residuals(double& mu, double& rpn, double& rdn, double& ndd, double& cn, double& xs) {
        double* X = bg + c.X; double* V = bg + c.V; double* DU = bg + c.DU;
        C_apply(X, bg + c.CX);
        double musum = 0.0, rpmax = 0.0;
        for_rows([&](int g, int k, int r) {
            if (!fin(r)) return;
            const double gz = prim(g, k, X, V, DU, bg + c.CX, eps) - (g == MS_EPS ? 0.0 : rcs[r] * eps);
            const double v = gz + rs[r] - rh[r];
            rrp[r] = v;
            rpmax = fmax(rpmax, fabs(v));
            musum += rs[r] * rl[r];
        });
        mu = mact ? w.sum(musum) / wsum : 0.0;
        rpn = w.maxv(rpmax);
        defects(cn, xs);
        // gradient of the Lagrangian without the model multipliers: cost + G'lam  (into gX, gV, gDU)
        const double ge = Gt_apply([&](int r) { return rl[r]; }, bg + c.gX, bg + c.gV, bg + c.gDU, true);
        dual_residual(ge, bg + c.NX, bg + c.NV, rdn, ndd);
    }

    // defects c_t = Abar xi_t + Bbar du_t + g_t - xi_{t+1} of the iterate (into cX, cV)
    MPCQP_HD void defects(double& cn, double& xs) {
        double* X = bg + c.X; double* V = bg + c.V; double* DU = bg + c.DU;
        double cmax = 0.0, xmax = 0.0;
        for (int i = w.lane; i < nVt; i += WAVE) {
            const int t = i / nu, cc = i - t * nu, j = ctrl[t];
            const double v = vat(V, t - 1, cc, false) + (j >= 0 ? DU[j * nu + cc] : 0.0) - V[i];
            bg[c.cV + i] = v;
            cmax = fmax(cmax, fabs(v));
        }
        w.sync();
        for (int i = w.lane; i < nXt; i += WAVE) {
            const int t = i / nx, r = i - t * nx;
            double acc = bg[c.gv + i] - X[i];
            for (int k = 0; k < nx; ++k) acc += A[r + nx * k] * xat(X, t - 1, k, false);
            for (int cc = 0; cc < nu; ++cc) acc += Bu[r + nx * cc] * (V[t * nu + cc] + bg[c.cV + t * nu + cc]);
            bg[c.cX + i] = acc;
            cmax = fmax(cmax, fabs(acc));
            xmax = fmax(xmax, fabs(X[i]));
        }
        cn = w.maxv(cmax);
        xs = 1.0 + w.maxv(xmax);
        // Defects at the rounding floor of the states are noise, and a Newton system asked to remove them pays for it: the
        // sweeps multiply c by the cost-to-go, whose entries reach 1/delta = 1e12 for rows held at the barrier's cap, and
        // 2e-16 of noise becomes 2e-4 in the control rows of the system (randomised family 278: the dual residual settled at
        // 1e-2 once mu < 1e-9, the iterate 1.2e-5 from the optimum; with c = 0 the same systems are solved to 1e-13).
        // The iterate starts from a rollout and every step satisfies the linearised model to rounding, so the defects stay at
        // that floor; they are corrected only when they rise above it (never observed; kept for iterates that do not start
        // from a rollout).
        use_defect = cn > 1e-11 * xs;
        w.sync();
    }

    // dual residual of (gX, gV, gDU, ge) = cost gradient + G'(multipliers of the rows) with the model multipliers NX, NV:
    // r_xi_t = g_t - nu_t + Abar' nu_{t+1};  r_u_j = g_u + (Abar' nu_{t+1})_u at t = j_l;  r_e = 2 C eps + g_e
    MPCQP_HD void dual_residual(double ge, const double* NX, const double* NV, double& rdn, double& ndd) {
        double mx = 0.0, sc = 0.0;
        for (int i = w.lane; i < nXt + nVt; i += WAVE) {
            const bool isx = i < nXt;
            const int ii = isx ? i : i - nXt;
            const int t = isx ? ii / nx : ii / nu, k = isx ? ii - t * nx : ii - t * nu;
            double g0 = isx ? bg[c.gX + ii] : bg[c.gV + ii];
            double nu_t = isx ? NX[ii] : NV[ii];
            double an = 0.0;                     // (Abar' nu_{t+2})[component]
            if (t + 1 < Hp) {
                if (isx) {
                    for (int kk2 = 0; kk2 < nx; ++kk2) an += A[kk2 + nx * k] * NX[(t + 1) * nx + kk2];
                } else {
                    for (int kk2 = 0; kk2 < nx; ++kk2) an += Bu[kk2 + nx * k] * NX[(t + 1) * nx + kk2];
                    an += NV[(t + 1) * nu + k];
                }
            }
            const double r = g0 - nu_t + an;
            mx = fmax(mx, fabs(r));
            sc = fmax(sc, fmax(fabs(g0), fmax(fabs(nu_t), fabs(an))));
        }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_DEBUG_MS)
        { const double mxs = w.maxv(mx); if (w.lane == 0 && b == MPCQP_DEBUG_MS) printf("      rd state rows %.3e", mxs); }
#endif
        for (int i = w.lane; i < nDU; i += WAVE) {
            const int j = i / nu, cc = i - j * nu, t = jlt[j];
            double an = NV[t * nu + cc];
            for (int kk2 = 0; kk2 < nx; ++kk2) an += Bu[kk2 + nx * cc] * NX[t * nx + kk2];
            const double g0 = bg[c.gDU + i];
            const double r = g0 + an;
            mx = fmax(mx, fabs(r));
            sc = fmax(sc, fmax(fabs(g0), fabs(an)));
        }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_DEBUG_MS)
        { const double mxs = w.maxv(mx); if (w.lane == 0 && b == MPCQP_DEBUG_MS) printf(" +control rows %.3e\n", mxs); }
#endif
        if (d.neps) {
            const double re = 2.0 * m.Cwt[b] * eps + ge;
            mx = fmax(mx, fabs(re));
            sc = fmax(sc, fmax(fabs(2.0 * m.Cwt[b] * eps), fabs(ge)));
        }
        rdn = w.maxv(mx);
        ndd = 1.0 + w.maxv(sc);
        w.sync();
    }

    // row step of the dual-regularised system (mpcqp_bodies.h: Step::row_step)
    MPCQP_HD void row_step(int r, double rc, double& ds, double& dl) const {
        const double wi = rwi[r], a = rrp[r] + rgd[r];
        dl = wi * fma(rl[r], a, -rc);
        ds = -wi * fma(rs[r], a, delta * rc);
    }

    // One Newton solve with the current factor: rc(row) given; on return dX, dV, dDU, deps hold the step, nX, nV the
    // multipliers nu+, rgd[row] = (G dz)[row].
    template <class Fn>
    MPCQP_HD void newton(Fn rc, double schur, double phiee) {
        // g^ = cost gradient + G'(lam + D~ rp - wi rc)
        const double ge0 = Gt_apply([&](int r) { return rl[r] + rwi[r] * (rl[r] * rrp[r] - rc(r)); }, bg + c.gX, bg + c.gV, bg + c.gDU, true);
        solve(ge0, schur, phiee);
    }

    // the Newton system solved for the gradient in gX, gV, gDU, ge0 (defects of the iterate in cX, cV) with the current factor
    // and psi: dX, dV, dDU, deps <- the step, nX, nV <- the multipliers nu+, rgd[row] <- (G dz)[row]
    MPCQP_HD void solve(double ge0, double schur, double phiee) {
        sweep(bg + c.gX, bg + c.gV, bg + c.gDU, use_defect, bg + c.dX, bg + c.dV, bg + c.dDU, bg + c.nX, bg + c.nV);
        if (MPCQP_MS_REFINE) for (int i = w.lane; i < nDU; i += WAVE) bg[c.hDU + i] = bg[c.gDU + i];
        deps = 0.0;
        const double ge_keep = d.neps ? 2.0 * m.Cwt[b] * eps + ge0 : 0.0;
        if (d.neps) {
            const double ge = ge_keep;
            const double fy = dot_z(bg + c.fX, bg + c.fV, bg + c.fDU, bg + c.dX, bg + c.dV, bg + c.dDU);
            deps = -(ge + fy) / schur;
            for (int i = w.lane; i < nXt; i += WAVE) { bg[c.dX + i] += deps * bg[c.pX + i]; bg[c.nX + i] += deps * bg[c.qX + i]; }
            for (int i = w.lane; i < nVt; i += WAVE) { bg[c.dV + i] += deps * bg[c.pV + i]; bg[c.nV + i] += deps * bg[c.qV + i]; }
            for (int i = w.lane; i < nDU; i += WAVE) bg[c.dDU + i] += deps * bg[c.pDU + i];
            w.sync();
        }
        // Optional iterative refinement of the whole Newton solve (MPCQP_MS_REFINE, off: not needed once the gains are
        // refined, see factor()).  History: with K = -Lam^-1 S_u. from the explicit inverse alone, the control rows of the
        // Newton system were left with an O(1) residual as mu -> 0 (0.99 against terms of 6.6 at mu = 1e-9) -- eps cond(Lam)
        // of relative error in K, times the 1e-5-size state step, times the 1e12 of a row held at D~ = 1/delta -- the dual
        // residual grew instead of shrinking and the iterates wandered at mu < 1e-13.  One refinement step of K and k
        // against the stored Lam brings that residual to 1e-10 .. 1e-14 and the dual residual to 1e-14 (quadratic
        // convergence to the end).  The state rows hold by construction (adjoint nu+), so what is left lives in the control
        // rows and the slack row:
        //     r_u = R du + g_u + Bbar' nu+ + phi_u deps,     r_e = g_e + phi'dz + Phi_ee deps
        // and the correction solves the same system for it (one more sweep with the same factor).
        for (int pass = 0; pass < MPCQP_MS_REFINE; ++pass) {
            for (int i = w.lane; i < nXt; i += WAVE) bg[c.gX + i] = 0.0;
            for (int i = w.lane; i < nVt; i += WAVE) bg[c.gV + i] = 0.0;
            double mxr = 0.0;
            for (int i = w.lane; i < nDU; i += WAVE) {
                const int j = i / nu, cc = i - j * nu, t = jlt[j];
                double an = bg[c.nV + t * nu + cc];
                for (int k = 0; k < nx; ++k) an += Bu[k + nx * cc] * bg[c.nX + t * nx + k];
                const double r = bg[c.RD + i] * bg[c.dDU + i] + bg[c.hDU + i] + an + bg[c.fDU + i] * deps;
                bg[c.eDU + i] = r;
                mxr = fmax(mxr, fabs(r));
            }
            w.sync();
            for (int i = w.lane; i < nDU; i += WAVE) bg[c.gDU + i] = bg[c.eDU + i];
            double re = 0.0;
            if (d.neps) re = ge_keep + dot_z(bg + c.fX, bg + c.fV, bg + c.fDU, bg + c.dX, bg + c.dV, bg + c.dDU) + phiee * deps;
            w.sync();
#if !defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_DEBUG_MS)
            { mxr = w.maxv(mxr); if (w.lane == 0 && b == MPCQP_DEBUG_MS) printf("      newton control-row residual %.3e slack row %.3e deps %.3e\n", mxr, re, deps); }
#endif
            sweep(bg + c.gX, bg + c.gV, bg + c.gDU, false, bg + c.eX, bg + c.eV, bg + c.eDU, bg + c.mX, bg + c.mV);
            double de = 0.0;
            if (d.neps) {
                const double fy = dot_z(bg + c.fX, bg + c.fV, bg + c.fDU, bg + c.eX, bg + c.eV, bg + c.eDU);
                de = -(re + fy) / schur;
            }
            for (int i = w.lane; i < nXt; i += WAVE) { bg[c.dX + i] += bg[c.eX + i] + de * bg[c.pX + i]; bg[c.nX + i] += bg[c.mX + i] + de * bg[c.qX + i]; }
            for (int i = w.lane; i < nVt; i += WAVE) { bg[c.dV + i] += bg[c.eV + i] + de * bg[c.pV + i]; bg[c.nV + i] += bg[c.mV + i] + de * bg[c.qV + i]; }
            for (int i = w.lane; i < nDU; i += WAVE) bg[c.dDU + i] += bg[c.eDU + i] + de * bg[c.pDU + i];
            deps += de;
            w.sync();
        }
        G_dz();
    }

    // rgd[row] = (G dz)[row] of the direction in dX, dV, dDU, deps
    MPCQP_HD void G_dz() {
        C_apply(bg + c.dX, bg + c.CD);
        for_rows([&](int g, int k, int r) {
            if (!fin(r)) return;
            rgd[r] = prim(g, k, bg + c.dX, bg + c.dV, bg + c.dDU, bg + c.CD, deps) - (g == MS_EPS ? 0.0 : rcs[r] * deps);
        });
        w.sync();
    }

    // Stage data of Phi = H + G' diag(dt) G from the weights dt(row) of the finite rows: the stage diagonals QY, QV, RD,
    // the terminal diagonal XT, the border column phi (fX, fV, fDU); returns the rows' part of Phi_ee.  rgd is scratch.
    template <class Fn>
    MPCQP_HD double assemble(Fn dtf) {
        double ee = 0.0;
        for_rows([&](int g, int k, int r) {
            double dt = 0.0;
            if (fin(r)) {
                dt = dtf(r);
                const double cs = g == MS_EPS ? 1.0 : rcs[r];
                ee += cs * cs * dt;
            }
            rgd[r] = dt;
        });
        ee = w.sum(ee);
        w.sync();
        auto dt_ = [&](int g, int k) { return on(g) ? rgd[c.rowoff[g] + k] : 0.0; };
        auto cs_ = [&](int g, int k) { return on(g) ? rcs[c.rowoff[g] + k] : 0.0; };
        for (int r = w.lane; r < nY; r += WAVE) {
            bg[c.QY + r] = (m.Mblk ? 0.0 : 2.0 * m.Mdiag[(size_t)b * nY + r]) + dt_(MS_YMIN, r) + dt_(MS_YMAX, r);   // (block weights: add_Q, sweep)
            bg[c.CD + r] = cs_(MS_YMIN, r) * dt_(MS_YMIN, r) - cs_(MS_YMAX, r) * dt_(MS_YMAX, r);      // tB of the output rows
        }
        for (int r = w.lane; r < nVt; r += WAVE) {
            bg[c.QV + r] = 2.0 * m.Ldiag[(size_t)b * d.nU + r] + dt_(MS_UMIN, r) + dt_(MS_UMAX, r);
            bg[c.fV + r] = cs_(MS_UMIN, r) * dt_(MS_UMIN, r) - cs_(MS_UMAX, r) * dt_(MS_UMAX, r);
        }
        for (int r = w.lane; r < nDU; r += WAVE) {
            bg[c.RD + r] = 2.0 * m.Ndiag[(size_t)b * nDU + r] + dt_(MS_DUMIN, r) + dt_(MS_DUMAX, r);
            bg[c.fDU + r] = cs_(MS_DUMIN, r) * dt_(MS_DUMIN, r) - cs_(MS_DUMAX, r) * dt_(MS_DUMAX, r);
        }
        for (int k = w.lane; k < nx; k += WAVE) bg[c.XT + k] = dt_(MS_XMIN, k) + dt_(MS_XMAX, k);
        w.sync();
        for (int i = w.lane; i < nXt; i += WAVE) {
            const int t = i / nx, k = i - t * nx;
            double acc = 0.0;
            for (int a = 0; a < ny; ++a) acc += Cm[a + ny * k] * bg[c.CD + t * ny + a];
            if (t == Hp - 1) acc += cs_(MS_XMIN, k) * dt_(MS_XMIN, k) - cs_(MS_XMAX, k) * dt_(MS_XMAX, k);
            bg[c.fX + i] = acc;
        }
        w.sync();
        return ee;
    }

    // psi = -Phi^-1 phi (no defects) with the current factor and its multipliers q; returns the Schur complement of the slack
    // S = Phi_ee + phi'psi.  (The form 2 C + psi'H psi + sum dt (g_row'psi - cs_row)^2, a sum of non-negative terms whose
    //  error is second order in that of psi, was tried against the cancellation of the two 1e11-size terms: no difference
    //  in the polish, and the interior-point systems at mu < 1e-9 got worse -- with the first form the slack row of the
    //  system holds exactly for the psi that was computed, whatever its error.)
    MPCQP_HD double border(double phiee) {
        sweep(bg + c.fX, bg + c.fV, bg + c.fDU, false, bg + c.pX, bg + c.pV, bg + c.pDU, bg + c.qX, bg + c.qV);
        return phiee + dot_z(bg + c.fX, bg + c.fV, bg + c.fDU, bg + c.pX, bg + c.pV, bg + c.pDU);
    }

    // ---- active-set polish of an interior-point iterate (Step::polish of mpcqp_bodies.h in stage form) ----------------
    // Method of multipliers for the rows of the interior-point partition A = {i: lam_i > s_i}, started from l = lam on A:
    // with r = G z - h evaluated EXACTLY from the stage variables every round (and the defects c of the model with it),
    //     Phi = H + rho G_A'G_A (one Riccati factorisation),   g^ = cost gradient + G_A'(l + rho r_A),
    //     Newton step (dz, nu+) of the stage problem for g^, c;    z += dz,   l += rho (r_A + G_A dz).
    // The weights of the rows are 0 or rho, not the interior-point D~ spread over 24 decades, and the residuals never pass
    // through the ill-conditioned system: the rounds are self-correcting and end at the floor of the residual evaluation.
    // (Randomised family 278, third member: the interior-point Newton systems at mu < 1e-8 leave 3e-2 in their control rows
    // -- nine soft output rows and two move bounds held at D~ = 1e12 -- the dual residual settles at 1e-2 and the iterate
    // 1.2e-5 from the optimum although the iterate at mu = 1e-8 was right to 1e-8.  The polish is entered from mu <= 1e-6.)
    // Accepted when r_A, the defects and the dual residual with l are at their floors, l >= 0 on A and every other row is
    // feasible: the point then satisfies the KKT conditions of the QP whatever iterate the polish started from.  Otherwise
    // dU and eps are restored, X^0 rolled out from them, and the interior-point iteration goes on.
    // Row arrays during the polish: rwi = 1/0 (row in A), rpp = l, rrp = r; s and lam are left alone.
    MPCQP_HD bool polish(double xs, int& nfact) {
        const double rho = MPCQP_MS_POLISH_RHO;
        double* X = bg + c.X; double* V = bg + c.V; double* DU = bg + c.DU;
        for (int k = w.lane; k < nDU; k += WAVE) bg[c.bDU + k] = DU[k];
        const double eps_keep = eps;
        for_rows([&](int, int, int r) {
            const bool a = fin(r) && rl[r] > rs[r];
            rwi[r] = a ? 1.0 : 0.0;
            rpp[r] = a ? rl[r] : 0.0;
        });
        w.sync();
        bool ok = false;
        const double* nuX = bg + c.NX; const double* nuV = bg + c.NV;     // multipliers of the model that go with l
        for (int fact = 0; fact < MPCQP_MS_POLISH_FACTS && !ok; ++fact) {
            ++nfact;
            const double ee = assemble([&](int r) { return rho * rwi[r]; });
            if (!factor()) break;
            const double phiee = d.neps ? 2.0 * m.Cwt[b] + ee : 1.0;
            const double schur = d.neps ? border(phiee) : 1.0;
            bool retry = false, solved = false;
            double rpa_prev = 1e300, lmax = 0.0;
            int nsteps = 0;                   // Newton steps taken with this working set
            for (int round = 0; round < 8; ++round) {
                // exact residuals of the rows and of the model at the current point
                C_apply(X, bg + c.CX);
                double rpa = 0.0;
                for_rows([&](int g, int k, int r) {
                    if (!fin(r)) return;
                    const double v = prim(g, k, X, V, DU, bg + c.CX, eps) - (g == MS_EPS ? 0.0 : rcs[r] * eps) - rh[r];
                    rrp[r] = v;
                    rpa = fmax(rpa, rwi[r] * fabs(v));
                });
                rpa = w.maxv(rpa);
                double cn;
                defects(cn, xs);
                if (!(rpa == rpa) || !(cn == cn)) break;
                // r_A that stops shrinking above its floor: the rows of A cannot all be tight (with the right rows a round
                // contracts by 1e-3 or better; family 278's first attempt, one row too many, by 0.14)
                if (nsteps >= 2 && rpa > 1e-13 * nh && rpa >= 0.1 * rpa_prev) break;
                rpa_prev = rpa;
                const bool last = rpa <= 1e-13 * nh && cn <= 1e-13 * xs && !retry;
                const double ge = Gt_apply([&](int r) { return last ? rpp[r] : rwi[r] * fma(rho, rrp[r], rpp[r]); }, bg + c.gX, bg + c.gV, bg + c.gDU, true);
                if (last) {
                    double rdn, ndd;
                    dual_residual(ge, nuX, nuV, rdn, ndd);
#if !defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_DEBUG_MS)
                    if (w.lane == 0 && b == MPCQP_DEBUG_MS) printf("  [ms] polish set %d round %d LAST rpa %.3e cn %.3e rd %.3e (ndd %.3e)\n", fact, round, rpa, cn, rdn, ndd);
#endif
                    if (!(rdn == rdn)) break;
                    if (rdn <= MPCQP_MS_POLISH_RD * ndd) {
                        // the equality-constrained problem of this working set is solved: l >= 0 on A, every other row feasible?
                        double lmin = 0.0;
                        bool infeas = false;
                        lmax = 0.0;
                        for_rows([&](int, int, int r) {
                            if (!fin(r)) return;
                            if (rwi[r] != 0.0) { lmax = fmax(lmax, fabs(rpp[r])); lmin = fmin(lmin, rpp[r]); }
                            else infeas = infeas || rrp[r] > 1e-11 * nh;
                        });
                        lmax = w.maxv(lmax);
                        solved = true;
                        ok = !w.any(infeas) && w.minv(lmin) >= -1e-12 * (1.0 + lmax);
                        break;
                    }
                    retry = true;             // r_d not there yet: one more step (G_A'l^ is needed for it)
                    continue;
                }
                retry = false;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_DEBUG_MS)
                if (w.lane == 0 && b == MPCQP_DEBUG_MS) printf("  [ms] polish set %d round %d rpa %.3e cn %.3e\n", fact, round, rpa, cn);
#endif
                solve(ge, schur, phiee);
                ++nsteps;
                for (int k = w.lane; k < nDU; k += WAVE) DU[k] += bg[c.dDU + k];
                for (int i = w.lane; i < nXt; i += WAVE) X[i] += bg[c.dX + i];
                for (int i = w.lane; i < nVt; i += WAVE) V[i] += bg[c.dV + i];
                eps += deps;
                for_rows([&](int, int, int r) {
                    if (fin(r) && rwi[r] != 0.0) rpp[r] = fma(rho, rrp[r] + rgd[r], rpp[r]);
                });
                nuX = bg + c.nX; nuV = bg + c.nV;
                w.sync();
            }
            if (ok || !solved || fact + 1 >= MPCQP_MS_POLISH_FACTS) break;
            // Equality-constrained problem solved but not the QP: the next working set by the multiplier rule of the method,
            // l_i + rho r_i > 0 (for a row of A, l holds it already), with the acceptance thresholds as dead bands -- a row of
            // A whose multiplier went negative is dropped, a violated row outside A is added (Step::polish; interior-point
            // partitions of degenerate vertices, where weakly active rows have s_i ~ lam_i, need it).  The point is kept.
            bool chg = false;
            for_rows([&](int, int, int r) {
                if (!fin(r)) return;
                if (rwi[r] != 0.0) {
                    if (rpp[r] < -1e-12 * (1.0 + lmax)) { rwi[r] = 0.0; rpp[r] = 0.0; chg = true; }
                } else if (rrp[r] > 1e-11 * nh) {
                    rwi[r] = 1.0; chg = true;
                }
            });
            if (!w.any(chg)) break;
            w.sync();
        }
        if (!ok) {
            for (int k = w.lane; k < nDU; k += WAVE) DU[k] = bg[c.bDU + k];
            eps = eps_keep;
            w.sync();
            rollout(DU, X, V);
        }
        return ok;
    }

    static constexpr int ST_OPTIMAL = 0, ST_ITERATION_LIMIT = 1, ST_ERROR = 2;

    MPCQP_HD int run(int& iters_out, double& defect_out) {
        double* X = bg + c.X; double* V = bg + c.V; double* DU = bg + c.DU;
        // warm start: dU shifted (transcription.jl:1001-1004), X^0 rolled out from it, multipliers of the model 0
        const double* Zg = io.Z + (size_t)b * d.nZ;
        const bool cold = d.flags & 2u;
        for (int k = w.lane; k < nDU; k += WAVE) DU[k] = (!cold && k < nDU - nu) ? Zg[k + nu] : 0.0;
        eps = (!cold && d.neps) ? Zg[d.nZ - 1] : 0.0;
        for (int i = w.lane; i < nXt; i += WAVE) bg[c.NX + i] = 0.0;
        for (int i = w.lane; i < nVt; i += WAVE) bg[c.NV + i] = 0.0;
        w.sync();
        rollout(DU, X, V);
        const double eps_ws = eps;
        double mu = 0.0, rpn = 0.0, rdn = 0.0, ndd = 1.0, cn = 0.0, xs = 1.0;
        int status = ST_ITERATION_LIMIT, it = 0;
        // starting point of the rows: s = max(h - G z, 1), lam = 10 / s   (Step::run)
        C_apply(X, bg + c.CX);
        for_rows([&](int g, int k, int r) {
            if (!fin(r)) return;
            const double gz = prim(g, k, X, V, DU, bg + c.CX, eps) - (g == MS_EPS ? 0.0 : rcs[r] * eps);
            rs[r] = fmax(rh[r] - gz, 1.0);
            rl[r] = 10.0 / rs[r];
        });
        w.sync();
        double step_c = 1e300, zabs_c = 0.0, rd_prev = 1e300, rp_prev = 1e300, alpha_prev = 0.0, rd_best = 1e300;
        int rd_flat = 0, npolish = 0;
        double polmu_next = 1e-6;
        const int max_iter = mact ? d.max_iter : 1;
        while (true) {
            { const long long t0_ = clk(); residuals(mu, rpn, rdn, ndd, cn, xs); prof_[0] += (double)(clk() - t0_); }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_DEBUG_MS)
            if (w.lane == 0 && b == MPCQP_DEBUG_MS) printf("  [ms] it %2d mu %.3e rd %.3e (ndd %.3e) rp %.3e (nh %.2e) defect %.3e step %.3e eps %.6e delta %.1e\n", it, mu, rdn, ndd, rpn, nh, cn, step_c, eps, delta);
#endif
            if (!(mu == mu) || !(rdn == rdn) || !(rpn == rpn) || !(cn == cn)) { status = ST_ERROR; break; }
            // (a dual residual that an exact evaluation finds where the previous one left it although the step in between
            //  was nearly full sits on the float64 floor of the Newton systems -- rows held at D~ = 1/delta -- and counts as
            //  converged; the step criterion vouches for the inputs then.  Same rule as Step::run.)
            // ... or one that has not come below half of its best value for three iterations once the gap is closed (it then
            // hops around its floor: the nearly-full-step rule alone misses it -- member 251 of the C3 check ran 80 iterations
            // on a dual residual of 5e-4 relative, 1.4e-5 from the optimum)
            if (rdn < 0.5 * rd_best) { rd_best = rdn; rd_flat = 0; } else if (mu <= d.gap_tol) ++rd_flat;
            // (... and never one above 1e-5 relative: the Newton systems of a problem whose polish fails can leave the dual
            //  residual at O(1) once mu < 1e-10 -- randomised family 60, 1e-4 from the optimum -- and such a point is flagged
            //  with the iteration limit, not returned as optimal)
            const bool rd_stalled = ((rdn >= 0.5 * rd_prev && alpha_prev >= 0.9) || rd_flat >= 3) && rdn <= 1e-5 * ndd;
            rd_prev = rdn;
            // The primal residual of a dual-regularised iteration follows r_p <- (1 - alpha) r_p + alpha delta dlam: once the
            // gap is small and r_p no longer shrinks although the steps are long, it sits on delta dlam -- rows whose
            // multipliers still have a long way to go (soft rows, 2 Cwt eps ~ 1e5) -- and the iteration has become a method
            // of multipliers with penalty 1/delta: a ten times smaller delta makes it ten times faster (down to 1e-12; a
            // factorisation that breaks raises it again, below).
            if (MPCQP_MS_ADAPT_DELTA && mu <= 1e-6 && rpn > 10.0 * d.res_tol * nh && rpn >= 0.5 * rp_prev && alpha_prev >= 0.5 && delta > 1e-12)
                delta *= 0.1;
            rp_prev = rpn;
            const bool conv = mu <= d.gap_tol && (rdn <= d.res_tol * ndd || rd_stalled) && rpn <= 10.0 * d.res_tol * nh &&
                              cn <= d.res_tol * xs && step_c <= 1e-6 * fmax(1.0, zabs_c);
            if (conv && mact) { status = ST_OPTIMAL; break; }
            // no finite row at all: the first Newton step IS the optimum (what ExplicitMPC computes, explicitmpc.jl:216)
            if (!mact && it > 0) { status = ST_OPTIMAL; it = 0; break; }
            if (it >= max_iter) break;
            // Active-set polish once the gap is small: first at mu <= 1e-6, again after every further factor 100 (Step::run)
            if (MPCQP_MS_POLISH && mact && mu <= polmu_next && rpn <= 1e-6 * nh && cn <= 1e-6 * xs && npolish < MPCQP_MS_POLISH_BUDGET && !(d.flags & 16u)) {
                polmu_next = 1e-2 * mu;
                const long long t6_ = clk();
                const bool pok_ = polish(xs, npolish);
                prof_[6] += (double)(clk() - t6_);
                if (pok_) { status = ST_OPTIMAL; break; }
                residuals(mu, rpn, rdn, ndd, cn, xs);        // (the row arrays served the polish)
            }
            // D~ of the rows -> stage diagonals, border column phi, Phi_ee
            const long long t1_ = clk();
            const double ee = assemble([&](int r) {
                const double wi = 1.0 / fma(delta, rl[r], rs[r]);
                rwi[r] = wi;
                return rl[r] * wi;
            });
            prof_[1] += (double)(clk() - t1_);
            const long long t2_ = clk();
            const bool fok_ = factor();
            prof_[2] += (double)(clk() - t2_);
            if (!fok_) {
                // a pivot of some Lam_t <= 0: Phi left float64's range (rows held at D~ = 1/delta).  No step is taken with
                // that factor; the iteration goes on with a 100 times larger dual regularisation (it biases nothing: delta
                // multiplies the multiplier step, which vanishes at the optimum; same rule as Step::run)
                if (delta < 1e-8) { delta *= 100.0; continue; }
                status = ST_ERROR;
                break;
            }
            double schur = 1.0;
            const double phiee = d.neps ? 2.0 * m.Cwt[b] + ee : 1.0;
            const long long t3_ = clk();
            if (d.neps) schur = border(phiee);
            prof_[3] += (double)(clk() - t3_);
            const long long t4_ = clk();
            double smu = 0.0, tmax = 1.0;
            for (int pass = 0; pass < 2; ++pass) {
                const double cpp = pass ? 1.0 : 0.0;
                if (pass == 0) for_rows([&](int, int, int r) { rpp[r] = 0.0; });
                w.sync();
                newton([&](int r) { return fma(cpp, rpp[r], fma(rs[r], rl[r], -smu)); }, schur, phiee);
                double ppsum = 0.0;
                tmax = pass ? 1e-300 : 1.0;
                for_rows([&](int, int, int r) {
                    if (!fin(r)) return;
                    double ds, dl;
                    row_step(r, fma(cpp, rpp[r], fma(rs[r], rl[r], -smu)), ds, dl);
                    tmax = fmax(tmax, -ds / rs[r]);
                    tmax = fmax(tmax, -dl / rl[r]);
                    rpp[r] = pass ? ds : ds * dl;
                    rgd[r] = pass ? dl : rgd[r];
                    ppsum += ds * dl;
                });
                if (pass == 0) {
                    if (!mact) break;
                    const double aaff = 1.0 / w.maxv(tmax);
                    const double muaff = (1.0 - aaff) * mu + aaff * aaff * w.sum(ppsum) / wsum;
                    double sig = muaff / mu;
                    sig = sig * sig * sig;
                    smu = sig * mu;
                }
                w.sync();
            }
            prof_[4] += (double)(clk() - t4_);
            const long long t5_ = clk();
            double alpha = 1.0;
            if (mact) {
                const double amin = 1.0 / w.maxv(tmax);
                const double ahi = fmin(1.0, 0.9999 * amin);
                double pmin = 1e300, psum = 0.0;
                for_rows([&](int, int, int r) {
                    if (!fin(r)) return;
                    const double p = (rs[r] + ahi * rpp[r]) * (rl[r] + ahi * rgd[r]);
                    pmin = fmin(pmin, p);
                    psum += p;
                });
                pmin = w.minv(pmin);
                psum = w.sum(psum);
                alpha = (pmin * wsum >= 0.01 * psum) ? ahi : fmin(1.0, 0.99 * amin);
                for_rows([&](int, int, int r) {
                    if (!fin(r)) return;
                    rs[r] += alpha * rpp[r];
                    rl[r] += alpha * rgd[r];
                });
            }
            double stc = alpha >= 0.5 ? 0.0 : 1e300, zab = 0.0;      // (a blocked step says nothing about convergence)
            for (int k = w.lane; k < nDU; k += WAVE) {
                const double st = alpha * bg[c.dDU + k];
                stc = fmax(stc, fabs(st)); zab = fmax(zab, fabs(DU[k]));
                DU[k] += st;
            }
            for (int i = w.lane; i < nXt; i += WAVE) { X[i] += alpha * bg[c.dX + i]; bg[c.NX + i] += alpha * (bg[c.nX + i] - bg[c.NX + i]); }
            for (int i = w.lane; i < nVt; i += WAVE) { V[i] += alpha * bg[c.dV + i]; bg[c.NV + i] += alpha * (bg[c.nV + i] - bg[c.NV + i]); }
            eps += alpha * deps;
            alpha_prev = alpha;
            step_c = w.maxv(stc); zabs_c = w.maxv(zab);
            w.sync();
            prof_[5] += (double)(clk() - t5_);
            ++it;
        }
        if (status == ST_ITERATION_LIMIT && !(rpn <= 1e-6 * nh && cn <= 1e-6 * xs) && !(d.flags & 32u)) status = ST_ERROR;
        if (status == ST_ERROR) {          // mpc.Z~ .= Z~s   (execute.jl:499-500)
            for (int k = w.lane; k < nDU; k += WAVE) DU[k] = (!cold && k < nDU - nu) ? Zg[k + nu] : 0.0;
            eps = eps_ws;
            w.sync();
            rollout(DU, X, V);
        }
        iters_out = it + npolish;      // factorisations: interior-point iterations + polish attempts
        defect_out = cn;
        if (io.audit && w.lane == 0) {
            double* au = io.audit + (size_t)b * 4;
            au[0] = mu; au[1] = rdn / ndd; au[2] = rpn / nh; au[3] = 0.0;
        }
        return status;
    }
};

// `scratch`: the wavefront's horizon-long data in HBM (make_ms_carve(d, m).big doubles) when IN_HBM, else unused: they live
// in LDS behind the small block.  (A template parameter, not a run-time choice: a pointer that may be LDS or HBM is a
// generic pointer, every access a flat_load / flat_store -- which count on BOTH memory counters, so that not even the
// LDS-only fences of the stage loops could run ahead of the stores.)
template <bool IN_HBM, class W>
MPCQP_HD void ms_step_body(W& w, const Dims& d, const Model& m, const StepIO& io, const MsIO& ms, int b, double* sm, double* scratch) {
    double* big;
    if constexpr (IN_HBM) big = scratch;
    else big = sm + make_ms_carve(d, m).small;
    MsStep<W> st(w, d, m, io, b, sm, big);
    const long long tp0_ = MsStep<W>::clk();
    (void)tp0_;
    st.load();
    st.build_rows();
    int iters = 0;
    double defect = 0.0;
    const int status = st.run(iters, defect);
    const double* DU = st.bg + st.c.DU;
    for (int k = w.lane; k < d.nDU; k += WAVE) io.Z[(size_t)b * d.nZ + k] = DU[k];
    if (d.neps && w.lane == 0) io.Z[(size_t)b * d.nZ + d.nZ - 1] = st.eps;
    for (int k = w.lane; k < d.nu; k += WAVE) io.u0[(size_t)b * d.nu + k] = DU[k] + io.lastu0[(size_t)b * d.nu + k];
    if (io.Yhat0) {                // predict!: Y^0 = C^ X^0 + D^d D^0  (transcription.jl:1136-1145 with E = [0 diag(C^)])
        st.C_apply(st.bg + st.c.X, st.bg + st.c.CX);
        for (int r = w.lane; r < d.nY; r += WAVE) {
            const int t = r / d.ny, a = r - t * d.ny;
            double acc = st.bg[st.c.CX + r];
            for (int e = 0; e < d.nd; ++e) acc += m.Dd[(size_t)b * d.ny * d.nd + a + d.ny * e] * io.Dhat0[(size_t)b * d.nD + t * d.nd + e];
            io.Yhat0[(size_t)b * d.nY + r] = acc;
        }
    }
    if (io.kf_predict) {
        // updatestate! (predict_estimate_obsv!, kalman.jl:298-309) with the input just computed: x^0 <- A^ x^0 + B^u u0 + B^d d0 + (f^op - x^op)
        const int nx = d.nxh, nu = d.nu, nd = d.nd;
        const double* xh = sm + st.c.x0;
        const double* A = m.Ahat + (size_t)b * nx * nx;
        for (int i = w.lane; i < nx; i += WAVE) {
            double acc = m.dop ? m.dop[(size_t)b * nx + i] : 0.0;
            for (int k = 0; k < nx; ++k) acc += A[i + nx * k] * xh[k];
            for (int cc = 0; cc < nu; ++cc)
                acc += m.Bu[(size_t)b * nx * nu + i + nx * cc] * (DU[cc] + io.lastu0[(size_t)b * nu + cc]);
            for (int e = 0; e < nd; ++e) acc += m.Bd[(size_t)b * nx * nd + i + nx * e] * io.d0[(size_t)b * nd + e];
            io.xhat0_out[(size_t)b * nx + i] = acc;
        }
    }
    if (ms.Xhat)
        for (int i = w.lane; i < d.nxh * d.Hp; i += WAVE) ms.Xhat[(size_t)b * d.nxh * d.Hp + i] = st.bg[st.c.X + i];
#if defined(MPCQP_MS_PROFILE)
    if (ms.Xhat && w.lane == 0) { st.prof_[7] = (double)(MsStep<W>::clk() - tp0_); for (int i = 0; i < 8; ++i) ms.Xhat[(size_t)b * d.nxh * d.Hp + i] = st.prof_[i]; }
#endif
    if (w.lane == 0) {
        io.status[b] = status;
        if (io.iters) io.iters[b] = iters;
        if (ms.defect) ms.defect[b] = defect;
    }
}

}  // namespace mpcqp
