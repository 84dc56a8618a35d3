// mhe_types.h -- plain-old-data of the batched linear MovingHorizonEstimator kernels.
//
// One estimator (one QP) per 16-lane DPP row, four estimators per wavefront.  Lane r of a row owns
// component r of every stage vector (x̂0(j), ŵ(j), v̂(j)) and row r of every stage matrix, so an
// nx̂ x nx̂ block lives in nx̂ registers per lane and a product / inverse is a chain of
// `v_mov_b64_dpp row_newbcast:k` + `v_fma_f64` -- no LDS round trips on the critical path.
//
// The QP is solved in the STATE-SEQUENCE variables X = (x̂0(k-Nk+p), ..., x̂0(k+p)) instead of the
// reference's condensed Z̃ = [x̂0arr; Ŵ] (estimator/mhe/transcription.jl:2): ŵ(j) = x(j+1) - Â x(j) - g(j)
// is a bijection, the optimum is the same, and the Newton matrix becomes block tridiagonal
// (nx̂ x nx̂ blocks) instead of the dense 252 x 252 of the reference (He = 20, nx̂ = 12).
#pragma once
#include <stdint.h>

#include "mpcqp_types.h"

namespace mpcqp {
namespace mhe {

constexpr int RL = 16;    // lanes per estimator: one DPP row
constexpr int GPW = 4;    // estimators per wavefront
enum { CLS_X = 1u, CLS_W = 2u, CLS_V = 4u, CLS_S = 8u, CLS_L = 16u, CLS_C = 32u };   // bound classes; CLS_S: a slack variable ε relaxes some rows;
                                                                        // CLS_L: the bound arrays are window-long (one row per stage);
                                                                        // CLS_C: the softness arrays are window-long (C_x̂min ... C_v̂max)

struct Dims {
    int B, nx, nu, nym, nd, He;
    int direct;        // 1: current form (p = 0), 0: predictor form (p = 1)
    int NX;            // register columns: max(nx, nym) rounded up to a multiple of four (<= 16)
    int N;             // window length Nk of this period (1..He)
    int hy, hd;        // ring heads: window entry i of Y0m/U0/X0old lives in slot (hy + i) % He,
                       // entry i of D0 (He + 1 entries) in slot (hd + i) % (He + 1)
    uint32_t cls;      // CLS_X | CLS_W | CLS_V: bound classes that may hold finite rows
    int max_iter;
    double gap_tol, res_tol, dual_reg;
    int nwaves;        // wavefronts launched (each loops over groups of four estimators)
    int nslot;         // slots of scratch per wavefront (a slot = one double per active lane, 4 NX doubles)
    int cst_stride;    // doubles per estimator in the constant block
    uint32_t opt;      // experiment switches (MPCQP_MHE_OPT), 0 in production
};

// offsets (doubles) inside one estimator's constant block; a matrix is a "row-lane" array:
// element (row r, column c) at c * RL + r, so lane r reads its row with stride RL
struct CstMap {
    int A, At, Oc, OcT, T1, T2, T3, Bmid, Cm, Ct, CR, Q, R, Bu, Bd, Ddm, fx, stride;
};
MPCQP_HD inline CstMap cst_map(int NX, int nu, int nd) {
    CstMap m{};
    int o = 0;
    auto take = [&](int n) { int p = o; o += n * RL; return p; };
    m.A = take(NX); m.At = take(NX); m.Oc = take(NX); m.OcT = take(NX);
    m.T1 = take(NX); m.T2 = take(NX); m.T3 = take(NX); m.Bmid = take(NX);
    m.Cm = take(NX); m.Ct = take(NX); m.CR = take(NX); m.Q = take(NX); m.R = take(NX);
    m.Bu = take(nu); m.Bd = take(nd); m.Ddm = take(nd); m.fx = take(1);
    m.stride = o;
    return m;
}

// 64-double slots of one wavefront's scratch
struct SlotMap {
    int X, DX, DXA, RD, T, Q, G, E, SI, XR, WR, WW, WG, WGA, WD, VR, VV, VG, VGA, VD, PHI, TP, PSI, total;
};
MPCQP_HD inline SlotMap slot_map(int NX, int He, uint32_t cls) {
    SlotMap m{};
    const int Hs = He + 1;
    int o = 0;
    auto take = [&](int n) { int p = o; o += n; return p; };
    m.X = take(Hs); m.DX = take(Hs); m.DXA = take(Hs); m.RD = take(Hs); m.T = take(Hs); m.Q = take(Hs);
    m.G = take(He); m.E = take(He);
    m.SI = take(NX * Hs);
    m.XR = take((cls & CLS_X) ? 4 * Hs : 0);
    const int nw = (cls & CLS_W) ? He : 0, nv = (cls & CLS_V) ? Hs : 0;
    m.WR = take(4 * nw); m.WW = take(nw); m.WG = take(nw); m.WGA = take(nw); m.WD = take(nw);
    m.VR = take(4 * nv); m.VV = take(nv); m.VG = take(nv); m.VGA = take(nv); m.VD = take(nv);
    const int ns = (cls & CLS_S) ? Hs : 0;      // slack column of the Newton matrix: φ(s), its forward and backward solves
    m.PHI = take(ns); m.TP = take(ns); m.PSI = take(ns);
    m.total = o;
    return m;
}

// doubles of scratch per wavefront: nslot packed slots (one double per active lane)
MPCQP_HD inline size_t wave_scratch_doubles(int NX, int nslot) { return (size_t)nslot * GPW * NX + WAVE; }

struct Raw {               // inputs of mpcqp_mhe_set_model (ABI layout: column-major inside an estimator)
    const double *Ahat, *Bu, *Cm, *Bd, *Ddm, *fx;    // [B][nx*nx] [B][nx*nu] [B][nym*nx] [B][nx*nd] [B][nym*nd] [B][nx] (fx may be null)
    const double *Q, *R;                             // [B][nx*nx] [B][nym*nym]
};

struct Args {
    double* cst;                 // [B][cst_stride]
    double* P;                   // [B][NX*RL]  arrival covariance P̄ (row-lane)
    double* Pi2;                 // [B][NX*RL]  2 P̄⁻¹
    const double *xmin, *xmax, *wmin, *wmax, *vmin, *vmax;   // [B][RL] per channel, |v| >= BIG: absent (null: class absent);
                                                             // with CLS_L: x [B][He+1][RL] (arrival state, then the window blocks
                                                             // oldest first), w and v [B][He][RL]; a window of Nk < He uses the LAST Nk blocks
    const double *cxmin, *cxmax, *cwmin, *cwmax, *cvmin, *cvmax;   // [B][RL] softness c >= 0 of the rows (null: hard), CLS_S;
                                                             // with CLS_C: laid out like the window-long bounds (x [B][He+1][RL], w / v [B][He][RL])
    const double* Cwt;           // [B] weight of ε² (CLS_S)
    double* eps_out;             // [B] optimal slack ε (CLS_S)
    double *Y0m, *U0, *D0, *X0old;   // data windows (rings): [B][He][nym], [B][He][nu], [B][He+1][nd], [B][He][nx]
    const double *y0m_new, *u0_new, *d0_new;   // [B][nym], [B][nu], [B][nd]: data of this period (pushed as window entry N-1)
    double* xhat0;               // [B][nx]  in: x̂0 before this period (pushed into X0old), out: new estimate
    double* Zt;                  // [B][nx + He nx]  out: [x̂0arr; Ŵ] (reference order, zero beyond Nk), may be null
    double *Vhat, *Xhat;         // optional outs: [B][He nym], [B][He nx]
    int32_t *status, *iters;
    double* scratch;             // [nwaves][wave_scratch_doubles(NX, nslot)]
};

}  // namespace mhe
}  // namespace mpcqp
