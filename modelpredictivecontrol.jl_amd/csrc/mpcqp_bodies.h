// mpcqp_bodies.h -- bodies of the three kernels of the batched LinMPC step, one QP per
// wavefront.  Written against a tiny "wave" interface W (lane id, wave barrier, wave
// reductions) so that the very same source is compiled (a) by hipcc for gfx950, where W maps to
// s_barrier / DPP-shuffles, and (b) by g++ for tests/emu, where 64 host threads play the lanes.
// (b) is test infrastructure to debug index arithmetic without a GPU; the product library only
// ever contains (a).
//
//   K1 predmat_body  <-> init_predmat(::LinModel, ::SingleShooting)
//                        /root/reference/src/controller/transcription.jl:115-194
//   K2 hessian_body  <-> init_quadprog            src/controller/construct.jl:837-845
//   K3 step_body     <-> initpred!                src/controller/execute.jl:247-277
//                        linconstraint!           src/controller/transcription.jl:811-848
//                        set_warmstart_mpc!       src/controller/transcription.jl:997-1007
//                        optim_objective!         src/controller/execute.jl:466-505  (JuMP/OSQP
//                          replaced by a dual-regularised Mehrotra predictor-corrector IPM on
//                          the normal equations, Cholesky in LDS)
//                        getinput!                src/controller/execute.jl:536-546
//
// Nothing of size (rows of A) x nZ is ever formed: E is block-Toeplitz in the step-response
// blocks Σ_m = Ĉ S(m) B̂u (transcription.jl:134-139,156-165), Pu is a held cumulative sum
// (construct.jl:797-806), hard ΔU bounds are variable bounds (construct.jl:1217-1229).
#pragma once
#include <math.h>

#include <type_traits>

#include "mpcqp_types.h"

#if defined(MPCQP_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define MPCQP_TIC() const long long tic_ = clock64()
#define MPCQP_TOC(i) prof_[i] += (double)(clock64() - tic_)
#define MPCQP_TICK(name) const long long name = clock64()
#define MPCQP_TOCK(i, name) prof_[i] += (double)(clock64() - name)
#elif defined(MPCQP_ISA_MARKERS) && defined(__HIP_DEVICE_COMPILE__)
// Phase boundaries as comments in the device assembly (plus a scheduling barrier, so that no instruction crosses them):
// scripts/isa_phase_table.py attributes every instruction of the interior-point loop to its phase and class.
#define MPCQP_MARK_(txt) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MPCQP_MARK " txt ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MPCQP_TIC() MPCQP_MARK_("tic")
#define MPCQP_TOC(i) MPCQP_MARK_("toc " #i)
#define MPCQP_TICK(name) MPCQP_MARK_("tic " #name)
#define MPCQP_TOCK(i, name) MPCQP_MARK_("toc " #i)
#define MPCQP_MTIC() MPCQP_MARK_("tic")
#define MPCQP_MTOC(name) MPCQP_MARK_("toc " name)
#else
#define MPCQP_TIC() ((void)0)
#define MPCQP_TOC(i) ((void)0)
#define MPCQP_TICK(name) ((void)0)
#define MPCQP_TOCK(i, name) ((void)0)
#endif
#ifndef MPCQP_MTIC
#define MPCQP_MTIC() ((void)0)      // regions that only exist for the assembly markers (MPCQP_ISA_MARKERS)
#define MPCQP_MTOC(name) ((void)0)
#endif

#ifndef MPCQP_POLISH_FACTS
#define MPCQP_POLISH_FACTS 1      // factorisations (working sets) per polish attempt; measured on C3 (65536):
                                  // 1 -> 19.1 ms, 2 -> 19.5 ms, 4 -> 21.3 ms for the same 95 % of instances polished
#endif
#ifndef MPCQP_POLISH_BUDGET
#define MPCQP_POLISH_BUDGET 3     // no new polish attempt once this many polish factorisations are spent
#endif
// Timing experiments only (wrong results): MPCQP_FIXED_ITERS > 0 runs exactly that many interior-point iterations (no
// convergence test, no polish); MPCQP_ABLATE is a mask of phases that are skipped (1 E'DE, 2 factorisation, 4 triangular
// solves, 8 E v, 16 E'w, 32 row passes of the step rules, 64 structured part of G'DG).
#ifndef MPCQP_FIXED_ITERS
#define MPCQP_FIXED_ITERS 0
#endif
#ifndef MPCQP_ABLATE
#define MPCQP_ABLATE 0
#endif
#ifndef MPCQP_CHOL_REDUNDANT
#define MPCQP_CHOL_REDUNDANT 0    // 4 x 4 diagonal blocks of the factorisation factored redundantly by every lane (chol_static)
#endif
#ifndef MPCQP_CHOL_INVD
#define MPCQP_CHOL_INVD 0         // how 1/L_kk reaches lane k: 0 select, 1 LDS vector written by lane 0, 2 by every lane
#endif
#ifndef MPCQP_CHOL_RMW_BATCH
#define MPCQP_CHOL_RMW_BATCH 1    // read-modify-writes of the in-panel update / the U rows: all reads first, then all writes
#endif
#ifndef MPCQP_CHOL_DIAG
#define MPCQP_CHOL_DIAG 1         // chol_static: pivot guard by a floor, 1/L_kk from the diagonal slot after the last column
#endif
#ifndef MPCQP_FOLD_H
#define MPCQP_FOLD_H 1            // diagonal weights: the Newton matrix and H~ z are assembled without the packed H~ (Step::fold_H)
#endif
#ifndef MPCQP_CHOL_LDL
#define MPCQP_CHOL_LDL 1          // chol_static / solve_static: Phi = M D M' with unit-triangular M (stored negated), no square roots, no per-solve scaling
#endif
#ifndef MPCQP_SOLVE_DPP
#define MPCQP_SOLVE_DPP 1         // triangular solves of the specialised kernels blocked by DPP rows (Step::solve_static)
#endif
#ifndef MPCQP_POLISH_MU
#define MPCQP_POLISH_MU 1e-7      // complementarity gap at which the first polish attempt is made (then every factor 100); C3, round 5: 1e-4 15.6 ms, 1e-5 15.25, 1e-6 15.05, 1e-7 14.9, 1e-8 14.9, never 15.55 (profiles/r5e)
#endif
#ifndef MPCQP_POLISH_RP
#define MPCQP_POLISH_RP 1e-6      // ... and the relative primal residual it needs
#endif
#ifndef MPCQP_EAPPLY44_HCMAX
#define MPCQP_EAPPLY44_HCMAX 10   // longer control horizons take the general form of E v (the row-load form spills there)
#endif
#ifndef MPCQP_EAPPLY44_UNROLL
#define MPCQP_EAPPLY44_UNROLL 2   // block columns per group of loads in flight of the row-load form of E v (nu = ny = 4)
#endif
#ifndef MPCQP_EV_UNROLL
#if defined(MPCQP_STEP_WAVES) && MPCQP_STEP_WAVES == 1
// (one wavefront per SIMD has nobody to hide an LDS round trip behind and 512 registers: profiles/r6c -- E v of nZ~ = 151 took
//  21k cycles per product for 150 multiply-adds per lane, one round trip per group of four block columns)
#define MPCQP_EV_UNROLL 16
#else
#define MPCQP_EV_UNROLL 4         // block columns / steps per unrolled pass of the general E v and E'w forms
#endif
#endif
#define MPCQP_PRAGMA_(x) _Pragma(#x)
#define MPCQP_PRAGMA(x) MPCQP_PRAGMA_(x)
// Unrolled K steps (= operand loads in flight) of the matrix-core loops.  A kernel with one wavefront per SIMD
// (MPCQP_STEP_WAVES = 1: nZ~ > 64, two wavefronts per CU by LDS) has nobody to hide an LDS round trip behind and the
// whole register file to itself: measured on nZ~ = 106, B = 8192: panel update 2 / 8 / 16 steps 26.5 / 25.5 / 25.3 ms,
// then E'DE 2 / 4 / 8 steps 25.5 / 24.75 / 24.75 ms.  The two-waves-per-SIMD kernels sit at their register limit: 2.
#ifndef MPCQP_ETDE_UNROLL
#if defined(MPCQP_STEP_WAVES) && MPCQP_STEP_WAVES == 1
#define MPCQP_ETDE_UNROLL 4
#else
#define MPCQP_ETDE_UNROLL 2
#endif
#endif
// E'DE with the matrix-core operands in REGISTERS (round 6).  E is block-Toeplitz: the operand of tile column J at K step kk is
// the operand of tile column 0 at K step kk - DJ J (DJ = K steps per 16 columns), so ONE vector V[0 .. NK-1] per lane -- read
// from the Sigma table once per assembly -- serves every tile of every pass: the K loops become a straight stream of v_mul /
// v_mfma with the row factors d streaming in behind it (NK + NK + (eps row) LDS reads per assembly instead of one per tile
// column and K step, no address arithmetic, no selects in the loop).  Shapes: ny a multiple of 4, nu a divisor of 16, default
// move blocking, NK = Hp ny / 4 <= MPCQP_ETDE_VREG_MAX (2 VGPRs per entry).  0: operands from LDS at every K step (rounds 1-5).
#ifndef MPCQP_ETDE_VREG
#define MPCQP_ETDE_VREG 1
#endif
#ifndef MPCQP_ETDE_VREG_MAX
#define MPCQP_ETDE_VREG_MAX 32
#endif
#ifndef MPCQP_ETDE_VREG_CHUNK
#define MPCQP_ETDE_VREG_CHUNK 2   // K steps per batch of row-factor loads (one batch in flight ahead of the matrix-core stream)
#endif
// chunks of four columns per unrolled pass of the several-rows-per-lane substitutions (solve_big_static): the factor entries
// of a pass are requested together, so one LDS round trip is paid per pass (profiles/r6c: 86 cycles per column at 2)
#ifndef MPCQP_SOLVEBIG_DPP
#define MPCQP_SOLVEBIG_DPP 0       // several rows per lane: 1 = substitutions blocked by DPP rows (Step::solve_big_dpp), 0 = column at a time.
                                   // Measured (profiles/r6u): nZ~ = 106 / 8192 19.30 -> 19.29 ms, nZ~ = 151 / 4096 29.74 -> 29.74 ms, same
                                   // optima to 1e-10 -- no gain (the loads, the scaling by 1/L_ii and the ds_bpermute of a tile cost what the
                                   // v_readlane chain did), so the simpler form stays the default
#endif
#ifndef MPCQP_SOLVEBIG_UNROLL
#if defined(MPCQP_STEP_WAVES) && MPCQP_STEP_WAVES == 1
#define MPCQP_SOLVEBIG_UNROLL 8
#else
#define MPCQP_SOLVEBIG_UNROLL 2
#endif
#endif
// Issue priority of the wavefront by phase (s_setprio; two wavefronts share a SIMD's FP64 pipe, a 64-cycle v_mfma_f64 of one
// holds up the other's next dependent instruction): bit 0 factorisation high, bit 1 triangular solves high, bit 2 everything
// high except the matrix-core stream of E'DE.  0: no priority changes (rounds 1-5).
#ifndef MPCQP_PRIO
#define MPCQP_PRIO 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MPCQP_SETPRIO(bit, p) do { if ((MPCQP_PRIO) & (bit)) __builtin_amdgcn_s_setprio(p); } while (0)
#else
#define MPCQP_SETPRIO(bit, p) ((void)0)
#endif
// teams of three or more wavefronts: 0 keeps wavefront 0 (which carries the row state of the whole problem in registers) out of
// the matrix-core shares (E'DE passes, panel updates): the helpers split them among themselves.  Measured at nZ~ = 151, four
// wavefronts, 4096 controllers (profiles/r6h, after the spill of the substitution's relane site was repaired): 31.3 ms with
// wavefront 0 in, 31.8 ms without
#ifndef MPCQP_PANELROWS_RL
#define MPCQP_PANELROWS_RL 0    // rows below a panel's diagonal block (chol_big_panel_rows): 1 = right-looking column steps (measured slower: 29.7 -> 30.4 ms at nZ~ = 151, 19.4 -> 19.5 ms at 106)
#endif
#ifndef MPCQP_TEAM_DIAG
#define MPCQP_TEAM_DIAG 1       // teams of three or more: diagonal block of a panel in its DPP row (Step::chol_big_panel_diag)
#endif
#ifndef MPCQP_TEAM_MAIN_MFMA
#define MPCQP_TEAM_MAIN_MFMA 1
#endif
#ifndef MPCQP_HZ_UNROLL
#define MPCQP_HZ_UNROLL 4         // terms per unrolled pass of the two loops of H~ z (dual_residual)
#endif
#ifndef MPCQP_PANEL_UNROLL
#define MPCQP_PANEL_UNROLL 16     // (the panel update only exists beyond one row per lane)
#endif
#ifndef MPCQP_URMW_UNROLL
#define MPCQP_URMW_UNROLL 4       // read-modify-writes in flight along a row of Pu'dU Pu (several rows per lane)
#endif
#ifndef MPCQP_SPEC_DENSE
#define MPCQP_SPEC_DENSE 0        // 1: this specialisation carries the dense M_Hp / L_Hp products of the gradient
#endif
#ifndef MPCQP_SPEC_NW
#define MPCQP_SPEC_NW 0           // custom linear constraint rows per step of this specialisation (mpcqp_set_custom_constraints)
#endif
#ifndef MPCQP_ETAPPLY_NB
#define MPCQP_ETAPPLY_NB 3        // steps per (double-buffered) batch of E'w
#endif
#ifndef MPCQP_RELANE_MASK
#define MPCQP_RELANE_MASK 0x010u     // measured on C3: solve_into_dz only
#endif
// relane site i is active iff bit i of MPCQP_RELANE_MASK is set (see DevWave::relane)
#define MPCQP_RELANE(i) do { if ((MPCQP_RELANE_MASK >> (i)) & 1u) w.relane(); } while (0)

namespace mpcqp {

// 1/x for the per-row interior-point algebra.  Device: v_rcp_f64 + one Newton step.  Measured on
// gfx950 over 4M arguments spanning e^+-40: raw v_rcp_f64 4.6e-8, after one step 2.2e-15 relative
// (v_rsq_f64: 5.2e-8 / 4.2e-15) -- an IEEE f64 division is ~5x the instructions and the IPM does not
// need correctly rounded quotients (the converged iterate is verified with exact residuals).
// Host (emulator): plain division.
MPCQP_HD inline double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
#else
    return 1.0 / x;
#endif
}

// v_rcp_f64 alone (4.6e-8 relative): the ratio tests of the step length, where a fraction-to-the-boundary factor of
// 0.99 .. 0.9999 makes that precision irrelevant.
MPCQP_HD inline double rcp_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    return 1.0 / x;
#endif
}

// min / max without the operand canonicalisation (v_max_f64 x, x, x in front of every llvm.maxnum in IEEE mode):
// NaN handling is the instruction's own (a NaN operand loses), which is what fmin / fmax specify as well.
MPCQP_HD inline double fmx(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmax(a, b);
#endif
}
MPCQP_HD inline double fmn(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmin(a, b);
#endif
}
// max(a, |b|)
MPCQP_HD inline double fmx_abs(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmax(a, fabs(b));
#endif
}

// ------------------------------------------------------------------------------------------
// Compile-time dimensions: same member names as the runtime `Dims`, so the bodies below are
// written once against a dims policy DM.  Specialised kernels (mpcqp_dispatch.h) give the
// compiler constant trip counts everywhere and let the per-row IPM state live in registers.
// The number of measured disturbances nd and the move-blocking table stay run-time data.
// ------------------------------------------------------------------------------------------
template <int NU, int NY, int NXH, int HP, int HC, int NEPS, unsigned GMASK, int DNB = 1>
struct StaticDims {
    static constexpr bool is_static = true;
    static constexpr int nu = NU, ny = NY, nxh = NXH, Hp = HP, Hc = HC, neps = NEPS;
    static constexpr int nDU = NU * HC, nZ = NU * HC + NEPS, nU = NU * HP, nY = NY * HP;
    static constexpr int npk = pk_size(nZ);
    static constexpr int nw = MPCQP_SPEC_NW, nW = MPCQP_SPEC_NW * (HP + 1);   // custom rows: on-demand variants only (0 in the library's own kernels)
    // LDS stride of one Σ_m block: padded so the MFMA operand reads of E'DE (64 lanes = 4 block
    // columns x NY*NU entries) fall in distinct bank groups (see DESIGN.md "LDS layout")
    static constexpr int sp = (NY * NU) % 16 == 0 ? NY * NU + 8 : NY * NU;
    // row stride inside a block.  4 x 4 blocks: rows padded 4 -> 6 doubles (the block still takes
    // 24): with it the row reads of E v / E'w (quarter-waves of 4 steps x 4 rows, 16 B per lane) and,
    // with the K rows of a step taken in the order 0,2,1,3, the MFMA operand reads (half-waves of 4
    // block columns x 2 rows x 4 entries) all fall in distinct LDS banks
    static constexpr int rs = (NY == 4 && NU == 4) ? 6 : NU;
    static constexpr uint32_t gmask = GMASK;
    static constexpr int default_nb = DNB;          // 1: nb = [1,..,1,Hp-Hc+1]; 0: table in LDS
    // Zero blocks Σ_{-1} .. Σ_{-(Hc-1)} in front of the table (default move blocking, j_l = l): block
    // (t - j) of E then exists for every step t and block column j, and the Toeplitz products
    // (E v, E'w, the matrix-core operands of E'DE) address it without a select on t >= j.
    static constexpr int zpad = DNB == 1 ? HC - 1 : 0;
    int B, nd, nD, max_iter;
    double gap_tol, res_tol, dual_reg;
    uint32_t flags;
    MPCQP_HD static constexpr int eps_host() { return eps_host_group(GMASK, NEPS); }
    MPCQP_HD static constexpr int len(int p) {
        return p == P_BOX ? nZ : p == P_U ? nDU : p == P_DU ? nDU : p == P_Y ? nY : p == P_X ? nxh : p == P_W ? nW : 0;
    }
    MPCQP_HD static constexpr int cnt(int p) { return len(p) + ((p == P_Y && eps_host() >= 0) ? 1 : 0); }
    MPCQP_HD static constexpr int rowoff(int g) {
        int o = 0;
        for (int i = 0; i < g; ++i)
            if ((GMASK >> i) & 1u) o += cnt(i >> 1);
        return o;
    }
    MPCQP_HD static constexpr int nrows() { return rowoff(NGROUP); }
    MPCQP_HD explicit StaticDims(const Dims& d)
        : B(d.B), nd(d.nd), nD(d.nD), max_iter(d.max_iter), gap_tol(d.gap_tol), res_tol(d.res_tol),
          dual_reg(d.dual_reg), flags(d.flags) {}
    static bool matches_dims(const Dims& d) {
        return d.nu == NU && d.ny == NY && d.nxh == NXH && d.Hp == HP && d.Hc == HC &&
               d.neps == NEPS && d.default_nb == DNB && d.nw == MPCQP_SPEC_NW;
    }
    static bool matches(const Dims& d) { return matches_dims(d) && d.gmask == GMASK; }
};

// ------------------------------------------------------------------------------------------
// LDS carve-up of one problem (all doubles unless stated).  Same function on host (to size the
// dynamic LDS) and device.  Row arrays exist only for runtime dims (registers otherwise).
// ------------------------------------------------------------------------------------------
// custom linear constraint rows exist in this instantiation (runtime dims, or an on-demand variant built for nw > 0)
template <class DM>
constexpr bool has_w() {
    if constexpr (DM::is_static) return DM::nw > 0;
    else return true;
}

constexpr int NROWARR = 7;     // h, s, lam, rp, gd, pp, cs (cs only stored with runtime dims)

struct Carve {
    int S, Phi, zero, z, dz, q, zlo, zhi, gt, rd, dinv, zb, xh, F, tA[NPAIR], tB[NPAIR], ucum, exT, Wm;
    int rows[NROWARR];
    int jl, blk;               // int tables (offset in doubles, storage as int)
    int total;                 // doubles
};

// compile-time dims with one Newton-system row per lane (nZ~ <= 64): the register / DPP-row forms of the factorisation,
// the solves and the polish apply; larger compile-time problems share the several-rows-per-lane code of the runtime dims
template <class DM>
MPCQP_HD constexpr bool one_row_per_lane() {
    if constexpr (DM::is_static) return DM::nZ <= WAVE;
    else return false;
}

// Ŷ-row slots per lane whose output weights a one-row-per-lane kernel keeps in registers (Step::hwy_; 0: not cached)
template <class DM>
MPCQP_HD constexpr int hwq_() {
    if constexpr (DM::is_static) {
        constexpr int n = (DM::nY + WAVE - 1) / WAVE;
        return (DM::nZ <= WAVE && n <= 4) ? n : 0;
    } else {
        return 0;
    }
}

template <class DM>
MPCQP_HD inline int stride_S(const DM& d) {
    if constexpr (DM::is_static) return DM::sp;
    else return d.ny * d.nu;
}

template <class DM>
MPCQP_HD inline int zpad_S(const DM&) {
    if constexpr (DM::is_static) return DM::zpad;
    else return 0;
}

template <class DM>
MPCQP_HD inline int rowstride_S(const DM& d) {
    if constexpr (DM::is_static) return DM::rs;
    else return d.nu;
}

template <class DM>
MPCQP_HD inline Carve make_carve(const DM& d) {
    Carve c{};
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 1) & ~1; return r; };   // keep 16-B alignment
#ifdef MPCQP_LDS_FRONT_PAD
    (void)take(MPCQP_LDS_FRONT_PAD);      // (experiment of DESIGN 4 "out-of-line members": nothing within 4 KB of LDS offset 0)
#endif
    c.S = take((d.Hp + zpad_S(d)) * stride_S(d));   // zero blocks first (StaticDims::zpad)
    c.Phi = take(d.npk);
    c.zero = take(6);                             // four zeros: where masked lanes of a chunk read point; [4]: trash slot
                                                  // (target of the unconditional tile write-backs' invalid lanes)
    c.z = take(d.nZ); c.dz = take(d.nZ); c.q = take(d.nZ);
    c.gt = take(d.nZ); c.rd = take(d.nZ);
    c.dinv = take(d.nZ > WAVE ? d.nZ : 0);        // 1/L[k][k] of the several-rows-per-lane factorisation
    c.xh = take(d.nxh);                           // x̂0 of this period (corrected in place by the fused Kalman step)
    c.zb = take((DM::is_static && d.nZ <= WAVE) ? 0 : d.nZ);   // iterate kept while the polish runs (a register with compile-time dims, nZ~ <= 64)
    c.zlo = c.dz; c.zhi = c.gt;                   // only live while the rows are being set up
    c.F = -1;                                     // placed below (aliases the Ŷ-row scratch when it exists)
    MPCQP_UNROLL
    for (int p = 0; p < NPAIR; ++p) {
        const bool won = (d.gmask >> (2 * P_W)) & 3u;        // custom rows act through the Y and U primitives
        const bool on = ((d.gmask >> (2 * p)) & 3u) || (won && (p == P_Y || p == P_U));
        // pair Y's tA doubles as the E*v scratch, so it always exists
        c.tA[p] = take((on || p == P_Y) ? d.cnt(p) : 0);
        c.tB[p] = take((on && p != P_BOX) ? d.cnt(p) : 0);
    }
    // F is consumed by the end of Step::build (it goes into the right-hand sides; the optional Ŷ
    // output parks it in the caller's Yhat0 buffer), so it shares the tB scratch of the Ŷ rows
    c.F = (((d.gmask >> (2 * P_Y)) & 3u) && c.tB[P_Y] >= 0) ? c.tB[P_Y] : take(d.nY);
    c.ucum = take(d.nDU);
    c.Wm = take(((d.gmask >> (2 * P_W)) & 3u) ? d.nw * (d.ny + d.nu) : 0);   // Wy (nw x ny), Wu (nw x nu), column-major
    c.exT = take(((d.gmask >> (2 * P_X)) & 3u) ? d.Hc * d.nxh * d.nu : 0);
    const int M = DM::is_static ? 0 : d.nrows();
    MPCQP_UNROLL
    for (int a = 0; a < NROWARR; ++a) c.rows[a] = take(M);
    c.jl = take(d.default_nb ? 0 : (d.Hc + 2) / 2 + 1);      // tables only for move-blocking vectors
    c.blk = take(d.default_nb ? 0 : (d.Hp + 1) / 2 + 1);
    c.total = o;
    return c;
}

// which wavefront of a team takes item `idx` of a matrix-core job (helpers first; see MPCQP_TEAM_MAIN_MFMA), and whether this
// wavefront has any of `count` items
template <class W>
MPCQP_HD constexpr bool team_mine_mfma(int idx) {
    if (W::NTEAM == 1) return true;
    if (W::NTEAM >= 3 && !MPCQP_TEAM_MAIN_MFMA) return W::WV != 0 && idx % (W::NTEAM - 1) == W::WV - 1;
    return (idx + 1) % W::NTEAM == W::WV;
}
template <class W>
MPCQP_HD constexpr bool team_any_mfma(int count) {
    for (int i = 0; i < count; ++i)
        if (team_mine_mfma<W>(i)) return true;
    return false;
}

// ------------------------------------------------------------------------------------------
// condensed problem resident in LDS + structured products with E, Pu, ex̂
// ------------------------------------------------------------------------------------------
template <class W, class DM>
struct Qp {
    W& w;
    const DM& d;
    const Model& m;
    const int b;        // problem index
    double* sm;         // LDS base
    Carve c;
    int *jlt, *blkt;
    double *S, *Phi;
    int sp;             // LDS stride of one Σ_m block (>= ny*nu)
    int rs;             // LDS stride of one row of a block (>= nu)

    MPCQP_HD Qp(W& w_, const DM& d_, const Model& m_, int b_, double* sm_)
        : w(w_), d(d_), m(m_), b(b_), sm(sm_), c(make_carve(d_)), sp(stride_S(d_)), rs(rowstride_S(d_)) {
        jlt = reinterpret_cast<int*>(sm + c.jl);
        blkt = reinterpret_cast<int*>(sm + c.blk);
        S = sm + c.S + zpad_S(d_) * sp;       // block 0; zero blocks at negative indices
        Phi = sm + c.Phi;
    }

    // four consecutive doubles from a 16-byte aligned LDS address (two ds_read_b128)
    MPCQP_HD static void load4q(const double* p, double* x) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef double v2q_ __attribute__((ext_vector_type(2)));
        const v2q_ a = reinterpret_cast<const v2q_*>(p)[0], b2 = reinterpret_cast<const v2q_*>(p)[1];
        x[0] = a.x; x[1] = a.y; x[2] = b2.x; x[3] = b2.y;
#else
        x[0] = p[0]; x[1] = p[1]; x[2] = p[2]; x[3] = p[3];
#endif
    }

    MPCQP_HD bool pair_on(int p) const { return (d.gmask >> (2 * p)) & 3u; }
    MPCQP_HD bool group_on(int g) const { return (d.gmask >> g) & 1u; }
    // move blocking tables: j_l and the block that holds step t
    MPCQP_HD int jl(int j) const { return d.default_nb ? j : jlt[j]; }
    MPCQP_HD int blk(int t) const { return d.default_nb ? (t < d.Hc - 1 ? t : d.Hc - 1) : blkt[t]; }

    // dst(i, g[i]) for i < n, eight loads of a lane in flight: `for (i = lane; i < n; i += WAVE) dst(i, g[i])` compiles to one
    // conditional block per trip -- load, wait, store, one memory latency each (the set-up of a C3 step was 50 us of its 460,
    // round 5).  The loads here are unconditional (index clamped to 0: n >= 1) and all requested before the first is used.
    template <class Fn>
    MPCQP_HD void stage(const double* g, int n, Fn dst) {
        constexpr int NB = 8;
        for (int i0 = 0; i0 < n; i0 += NB * WAVE) {
            double v[NB];
            MPCQP_UNROLL
            for (int q_ = 0; q_ < NB; ++q_) { const int i = i0 + w.lane + WAVE * q_; v[q_] = g[i < n ? i : 0]; }
            MPCQP_UNROLL
            for (int q_ = 0; q_ < NB; ++q_) { const int i = i0 + w.lane + WAVE * q_; if (i < n) dst(i, v[q_]); }
        }
    }

    MPCQP_HD void load_tables() {
        const int nb_ = d.ny * d.nu, ns = d.Hp * nb_;
        stage(m.Stab + (size_t)b * ns, ns, [&](int i, double v) {
            const int blk_ = i / nb_, e = i - blk_ * nb_, a = e / d.nu;
            S[blk_ * sp + a * rs + (e - a * d.nu)] = v;
        });
        for (int i = w.lane; i < zpad_S(d) * sp; i += WAVE) sm[c.S + i] = 0.0;
        if (!d.default_nb) {
            for (int i = w.lane; i <= d.Hc; i += WAVE) jlt[i] = m.jl[i];
            for (int i = w.lane; i < d.Hp; i += WAVE) blkt[i] = m.blk[i];
        }
        if (pair_on(P_X) && m.exT) {      // (K2 of a specialisation with terminal rows runs before they are built)
            const int ne = d.Hc * d.nxh * d.nu;
            stage(m.exT + (size_t)b * ne, ne, [&](int i, double v) { sm[c.exT + i] = v; });
        }
        if constexpr (has_w<DM>()) {
            if (pair_on(P_W)) {
                const int n1 = d.nw * d.ny, n2 = d.nw * d.nu;
                for (int i = w.lane; i < n1; i += WAVE) sm[c.Wm + i] = m.Wy[(size_t)b * n1 + i];
                for (int i = w.lane; i < n2; i += WAVE) sm[c.Wm + n1 + i] = m.Wu[(size_t)b * n2 + i];
            }
        }
        if (w.lane < 4) sm[c.zero + w.lane] = 0.0;
        w.sync();
    }

    // custom linear constraints (relaxW, construct.jl:1086-1160): row (t, i), t = 0..Hp, of
    //   E_w = W̄y [0; E] + W̄u [Pu; pu]
    // acts on the output primitive of step t-1 (none for t = 0) and on the input of step
    // min(t, Hp-1) (pu = last block row of Pu)
    MPCQP_HD double Wy_(int i, int a) const { return sm[c.Wm + i + d.nw * a]; }
    MPCQP_HD double Wu_(int i, int cc) const { return sm[c.Wm + d.nw * d.ny + i + d.nw * cc]; }
    MPCQP_HD int blkW(int t) const { return blk(t < d.Hp ? t : d.Hp - 1); }
    // E_w[(t,i), (j,c)]
    MPCQP_HD double Ew_at(int t, int i, int j, int cc) const {
        double g = 0.0;
        if (t >= 1 && t - 1 >= jl(j)) {
            const double* Sb = S + (t - 1 - jl(j)) * sp + cc;
            for (int a = 0; a < d.ny; ++a) g += Wy_(i, a) * Sb[a * rs];
        }
        if (j <= blkW(t)) g += Wu_(i, cc);
        return g;
    }
    // Y / U primitives += (their share of) E_w' wW:  tY[(t',a)] += sum_i Wy[i,a] wW[(t'+1, i)],
    // tU[(j,c)] += sum_{t: block(t) = j} sum_i Wu[i,c] wW[(t, i)]; `fresh`: the scratch vector holds
    // nothing yet (its own pair is off)
    MPCQP_HD void W_fold(const double* wW, double* tY, bool freshY, double* tU, bool freshU) {
        const int nw = d.nw, ny = d.ny, nu = d.nu;
        for (int r = w.lane; r < d.nY; r += WAVE) {
            const int t = r / ny, a = r - t * ny;
            double acc = freshY ? 0.0 : tY[r];
            for (int i = 0; i < nw; ++i) acc += Wy_(i, a) * wW[(t + 1) * nw + i];
            tY[r] = acc;
        }
        for (int k = w.lane; k < d.nDU; k += WAVE) {
            const int j = k / nu, cc = k - j * nu;
            const int t1 = (j + 1 < d.Hc) ? jl(j + 1) : d.Hp + 1;      // step Hp repeats the last block
            double acc = freshU ? 0.0 : tU[k];
            for (int t = jl(j); t < t1; ++t)
                for (int i = 0; i < nw; ++i) acc += Wu_(i, cc) * wW[t * nw + i];
            tU[k] = acc;
        }
        w.sync();
    }

    // Held cumulative sums over the block columns (Pu v and Pu'w, construct.jl:797-806) without a loop
    // of lane-dependent length: lane k = (j, c) holds x[k] (0 on lanes >= nDU), log2(Hc) steps of
    // "fetch the value s blocks away and add".  One value per lane: nDU <= 64.  All lanes call.
    MPCQP_HD double block_prefix(double x) {            // sum_{jj <= j} x[(jj, c)]
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DM::is_static) {
            if constexpr (DM::nu == 4 && DM::nDU <= WAVE) {
                // four blocks per 16-lane row: inclusive scan inside the row by two row shifts (zero fill), then the
                // totals of the rows before (their last block) in ONE round of independent fetches
                x += W::template dpp<0x114>(x);          // row_shr:4
                x += W::template dpp<0x118>(x);          // row_shr:8
                const int row = w.lane >> 4, cc = w.lane & 3;
                double carry = 0.0;
                MPCQP_UNROLL
                for (int rr = 1; rr <= (DM::Hc - 1) / 4; ++rr) {
                    const double y = w.fetch(x, ((row - rr) << 4) + 12 + cc);
                    carry += row >= rr ? y : 0.0;
                }
                return x + carry;
            }
        }
#endif
        const int nu = d.nu, j = w.lane / nu;
        for (int s_ = 1; s_ < d.Hc; s_ <<= 1) {
            const double y = w.fetch(x, w.lane - s_ * nu);
            x += (j >= s_ && w.lane < d.nDU) ? y : 0.0;
        }
        return x;
    }
    MPCQP_HD double block_suffix(double x) {            // sum_{jj >= j} x[(jj, c)]
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DM::is_static) {
            if constexpr (DM::nu == 4 && DM::nDU <= WAVE) {
                x += W::template dpp<0x104>(x);          // row_shl:4
                x += W::template dpp<0x108>(x);          // row_shl:8
                const int row = w.lane >> 4, cc = w.lane & 3;
                double carry = 0.0;
                MPCQP_UNROLL
                for (int rr = 1; rr <= (DM::Hc - 1) / 4; ++rr) {
                    const double y = w.fetch(x, ((row + rr) << 4) + cc);      // lanes >= nDU hold zeros
                    carry += row + rr < 4 ? y : 0.0;
                }
                return x + carry;
            }
        }
#endif
        const int nu = d.nu, j = w.lane / nu;
        for (int s_ = 1; s_ < d.Hc; s_ <<= 1) {
            const double y = w.fetch(x, w.lane + s_ * nu);
            x += (j + s_ < d.Hc && w.lane < d.nDU) ? y : 0.0;
        }
        return x;
    }

    // out[r] = sum_k E[r,k] v[k]   (r < nY; v has >= nDU entries)
    // (a team of wavefronts splits the row slots of the zero-padded form: W::NTEAM, mpcqp_devwave.h)
    MPCQP_HD void E_apply(const double* v, double* out) {
        if constexpr (W::NTEAM > 1) w.post(TJ_EV, (int)(v - sm), (int)(out - sm));
        E_apply_share(v, out);
        if constexpr (W::NTEAM > 1) w.join();
    }
    MPCQP_HD void E_apply_share(const double* v, double* out) {
        MPCQP_RELANE(0);
        const int ny = d.ny, nu = d.nu;
        if constexpr (DM::is_static) {
            if (W::NTEAM == 1 && DM::nu == 4 && DM::nY <= 2 * WAVE && DM::Hc <= MPCQP_EAPPLY44_HCMAX && d.default_nb) {
                // every lane owns rows r0 = lane and r1 = lane + 64: the (wave-uniform) v[j,:]
                // loads are shared by both rows; a block column j > t reads a zero block (zpad)
                const int r0 = w.lane, r1 = w.lane + WAVE;
                const bool ok0 = r0 < DM::nY, ok1 = r1 < DM::nY;
                const int t0 = (ok0 ? r0 : 0) / DM::ny, a0 = (ok0 ? r0 : 0) % DM::ny;
                const int t1 = (ok1 ? r1 : 0) / DM::ny, a1 = (ok1 ? r1 : 0) % DM::ny;
                const double* Sa = S + t0 * sp + a0 * rs;        // block (t - j) at S_[-j * sp]
                const double* Sb = S + t1 * sp + a1 * rs;
                double x0 = 0.0, x1 = 0.0, y0 = 0.0, y1 = 0.0;
                MPCQP_PRAGMA(unroll MPCQP_EAPPLY44_UNROLL)
                for (int j = 0; j < DM::Hc; ++j) {
                    double vv[4], sa[4], sb[4];
                    load4q(v + j * 4, vv);
                    load4q(Sa - j * DM::sp, sa);
                    load4q(Sb - j * DM::sp, sb);
                    x0 = fma(sa[0], vv[0], x0); x1 = fma(sa[1], vv[1], x1);
                    y0 = fma(sb[0], vv[0], y0); y1 = fma(sb[1], vv[1], y1);
                    x0 = fma(sa[2], vv[2], x0); x1 = fma(sa[3], vv[3], x1);
                    y0 = fma(sb[2], vv[2], y0); y1 = fma(sb[3], vv[3], y1);
                    if (j % MPCQP_EAPPLY44_UNROLL == MPCQP_EAPPLY44_UNROLL - 1) MPCQP_SCHED_FENCE();     // bounds the loads in flight (registers)
                }
                if (ok0) out[r0] = x0 + x1;
                if (ok1) out[r1] = y0 + y1;
                return;
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DM::is_static) {
            if (DM::zpad > 0 && d.default_nb) {
                // any nu, ny with the zero-padded table: every block column for every row (no lane-dependent trip
                // count), the wave-uniform v[j,:] loads shared by the lane's rows r = lane, lane + 64, ..
                // (team: row slot q belongs to wavefront q mod NTEAM -- `mine` is a compile-time test after unrolling)
                constexpr int NR = (DM::nY + WAVE - 1) / WAVE;
                auto mine = [](int q_) { return (q_ + 1) % W::NTEAM == W::WV; };
                const double* Sa[NR];
                double acc[NR][2];
                MPCQP_UNROLL
                for (int q_ = 0; q_ < NR; ++q_) {
                    const int r = w.lane + WAVE * q_;
                    const int rr = r < DM::nY ? r : 0;
                    Sa[q_] = S + (rr / DM::ny) * DM::sp + (rr % DM::ny) * DM::rs;
                    acc[q_][0] = acc[q_][1] = 0.0;
                }
                if constexpr (W::NTEAM == 1 || (W::WV == 0 ? NR >= W::NTEAM : NR >= W::WV)) {
                MPCQP_PRAGMA(unroll MPCQP_EV_UNROLL)
                for (int j = 0; j < DM::Hc; ++j) {
                    MPCQP_UNROLL
                    for (int cc = 0; cc < DM::nu; ++cc) {
                        const double vv = v[j * DM::nu + cc];
                        MPCQP_UNROLL
                        for (int q_ = 0; q_ < NR; ++q_)
                            if (mine(q_)) acc[q_][cc & 1] = fma(Sa[q_][cc - j * DM::sp], vv, acc[q_][cc & 1]);
                    }
                }
                }
                MPCQP_UNROLL
                for (int q_ = 0; q_ < NR; ++q_) {
                    const int r = w.lane + WAVE * q_;
                    if (mine(q_) && r < DM::nY) out[r] = acc[q_][0] + acc[q_][1];
                }
                return;
            }
        }
#endif
        if constexpr (W::WV != 0) return;         // (forms without a team split: wavefront 0 alone)
        for (int r = w.lane; r < d.nY; r += WAVE) {
            const int t = r / ny, a = r - t * ny;
            double acc0 = 0.0;
            for (int j = 0; j < d.Hc && jl(j) <= t; ++j) {
                const double* Sb = S + (t - jl(j)) * sp + a * rs;
                const double* vj = v + j * nu;
                for (int cc = 0; cc < nu; ++cc) acc0 += Sb[cc] * vj[cc];
            }
            out[r] = acc0;
        }
    }

    // out[k] += scale * sum_r E[r,k] wv[r]   (k < nDU).  The t loop is wave-uniform (lanes of
    // later block columns just start contributing later), so wv[t,a] is a broadcast LDS read and
    // there is no divergent branch in the loop.
    MPCQP_HD void Et_apply_add(const double* wv, double* out, double scale = 1.0, int t_hi = -1) {
        if constexpr (W::NTEAM > 1) w.post(TJ_ETW, (int)(wv - sm), (int)(out - sm), t_hi, 0, scale);
        Et_apply_share(wv, out, scale, t_hi);
        if constexpr (W::NTEAM > 1) w.join();
    }
    MPCQP_HD void Et_apply_share(const double* wv, double* out, double scale, int t_hi) {
        MPCQP_RELANE(1);
        const int ny = d.ny, nu = d.nu;
        if (t_hi < 0) t_hi = d.Hp;          // only the steps t < t_hi contribute
        if constexpr (DM::is_static) {
            if (W::NTEAM == 1 && DM::ny == 4 && DM::nu == 4 && DM::nDU <= WAVE && d.default_nb) {
                // lane (j, a) reads whole rows S_{t-j}[a][0..3] (two 16-byte loads instead of four
                // strided 8-byte ones) and keeps one partial sum per channel c; the four a-lanes
                // of a block column are then added with two quad permutes and lane a keeps c = a.
                const int k = w.lane < DM::nDU ? w.lane : 0;
                const int j = k >> 2, a = k & 3;
                const double* Sb = S - j * sp + a * rs;     // block (t - j): a zero block before the column starts (zpad)
                const double* wa = wv + a;
                double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
                // batches of NB steps, double buffered: the loads of batch bt + 1 are in flight while batch bt is consumed
                constexpr int NB = MPCQP_ETAPPLY_NB, NBAT = (DM::Hp + NB - 1) / NB;
                double sr[2][NB][4], wt[2][NB];
                auto ld = [&](int bt, int buf) {
                    MPCQP_UNROLL
                    for (int u = 0; u < NB; ++u) {
                        const int t = bt * NB + u;
                        if (t >= DM::Hp) break;
                        load4q(Sb + t * DM::sp, sr[buf][u]);
                        wt[buf][u] = wa[t * 4];
                    }
                };
                ld(0, 0);
                MPCQP_UNROLL
                for (int bt = 0; bt < NBAT; ++bt) {
                    if (bt * NB >= t_hi) break;
                    const int buf = bt & 1;
                    if (bt + 1 < NBAT) ld(bt + 1, buf ^ 1);
                    MPCQP_SCHED_FENCE();
                    MPCQP_UNROLL
                    for (int u = 0; u < NB; ++u) {
                        const int t = bt * NB + u;
                        if (t >= DM::Hp) break;
                        const double wv_ = t < t_hi ? wt[buf][u] : 0.0;
                        p0 = fma(sr[buf][u][0], wv_, p0); p1 = fma(sr[buf][u][1], wv_, p1);
                        p2 = fma(sr[buf][u][2], wv_, p2); p3 = fma(sr[buf][u][3], wv_, p3);
                    }
                    MPCQP_SCHED_FENCE();
                }
                p0 = w.quad_sum(p0); p1 = w.quad_sum(p1); p2 = w.quad_sum(p2); p3 = w.quad_sum(p3);
                const double mine = a == 0 ? p0 : a == 1 ? p1 : a == 2 ? p2 : p3;
                if (w.lane < DM::nDU) out[k] += scale * mine;
                return;
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DM::is_static) {
            if (DM::zpad > 0 && d.default_nb) {
                // any nu, ny with the zero-padded table: no select on t >= j, the wave-uniform w[t,:] loads shared by the
                // lane's columns k = lane, lane + 64, ..
                // (team: column slot q belongs to wavefront q mod NTEAM)
                constexpr int NQ = (DM::nDU + WAVE - 1) / WAVE;
                auto mine = [](int q_) { return (q_ + 1) % W::NTEAM == W::WV; };
                const double* Sk[NQ];
                double acc[NQ][2];
                MPCQP_UNROLL
                for (int q_ = 0; q_ < NQ; ++q_) {
                    const int k = w.lane + WAVE * q_;
                    const int kk = k < DM::nDU ? k : 0;
                    Sk[q_] = S - (kk / DM::nu) * DM::sp + (kk % DM::nu);
                    acc[q_][0] = acc[q_][1] = 0.0;
                }
                if constexpr (W::NTEAM == 1 || (W::WV == 0 ? NQ >= W::NTEAM : NQ >= W::WV)) {
                MPCQP_PRAGMA(unroll MPCQP_EV_UNROLL)
                for (int t = 0; t < DM::Hp; ++t) {
                    if (t >= t_hi) break;
                    MPCQP_UNROLL
                    for (int a = 0; a < DM::ny; ++a) {
                        const double wt = wv[t * DM::ny + a];
                        MPCQP_UNROLL
                        for (int q_ = 0; q_ < NQ; ++q_)
                            if (mine(q_)) acc[q_][a & 1] = fma(Sk[q_][t * DM::sp + a * DM::rs], wt, acc[q_][a & 1]);
                    }
                }
                }
                MPCQP_UNROLL
                for (int q_ = 0; q_ < NQ; ++q_) {
                    const int k = w.lane + WAVE * q_;
                    if (mine(q_) && k < DM::nDU) out[k] += scale * (acc[q_][0] + acc[q_][1]);
                }
                return;
            }
        }
#endif
        if constexpr (W::WV != 0) return;         // (forms without a team split: wavefront 0 alone)
        for (int k = w.lane; k < d.nDU; k += WAVE) {
            const int j = k / nu, cc = k - j * nu, t0 = jl(j);
            const double* Sk = S + cc;
            double acc0 = 0.0, acc1 = 0.0;
            MPCQP_UNROLL4
            for (int t = 0; t < t_hi; ++t) {
                const bool ok = t >= t0;
                const double* Sb = Sk + (ok ? t - t0 : 0) * sp;
                const double* wt = wv + t * ny;
                double p0 = 0.0, p1 = 0.0;
                int a = 0;
                for (; a + 1 < ny; a += 2) { p0 += Sb[a * rs] * wt[a]; p1 += Sb[(a + 1) * rs] * wt[a + 1]; }
                if (a < ny) p0 += Sb[a * rs] * wt[a];
                acc0 += ok ? p0 : 0.0;
                acc1 += ok ? p1 : 0.0;
            }
            out[k] += scale * (acc0 + acc1);
        }
    }

    // idx (packed lower triangle of the ΔU block) -> (i, i'), i >= i'
    MPCQP_HD static void unpack_idx(int idx, int& i, int& ip) {
        i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > idx) --i;
        while ((i + 1) * (i + 2) / 2 <= idx) ++i;
        ip = idx - i * (i + 1) / 2;
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // E' diag(dd) E on the matrix cores: the one genuine contraction of the path
    // (nDU x nY x nDU).  v_mfma_f64_16x16x4_f64: A[i = lane&15][k = lane>>4], B[k][j = lane&15],
    // D[row = (lane>>4) + 4 reg][col = lane&15].  K runs over the rows r = (t, a) of E four at a
    // time; the 16-wide tiles run over the ΔU index; E is never formed -- operands come straight
    // from the block-Toeplitz table, and tiles whose block columns start after step t are skipped.
    typedef double v4d __attribute__((ext_vector_type(4)));
    // returns the first step t whose rows the ϵ row (tb) has been accumulated for (-1: no ϵ row).
    // Hg != nullptr: P is OVERWRITTEN with Hg (packed H̃ in global memory) + scale * E'DE on the whole stored
    // triangle (ϵ row and its diagonal included) instead of being updated in place -- no staging of H̃ in LDS and
    // no read-modify-write of the tiles; the H̃ entries are fetched in the accumulator layout before the K loop.
    // passes of EtDE_add_mfma over the tile rows: {0, 1} together, then one row per pass
    template <int I0, int NT, class F>
    static __device__ __forceinline__ void etde_passes(F& f) {
        if constexpr (I0 < NT) {
            f(std::integral_constant<int, I0>{});
            etde_passes<(I0 == 0 ? 2 : I0 + 1), NT>(f);
        }
    }
    // ow: P is OVERWRITTEN with (Hg ? H̃ : 0) + scale * E'DE on the whole stored triangle (Hg == nullptr: the weights of H̃
    // ride in dd, Step::fold_H)
    __device__ __forceinline__ int EtDE_add_mfma(const double* dd, double* P, double scale, const double* tb,
                                                 const double* Hg = nullptr, bool ow = false) {
        // (a team of wavefronts splits the passes over the tile rows: W::NTEAM, mpcqp_devwave.h)
        if constexpr (W::NTEAM > 1) w.post(TJ_ETDE, (int)(dd - sm), (int)(P - sm), tb ? (int)(tb - sm) : 0, (ow ? 1 : 0) | (Hg ? 2 : 0) | (tb ? 4 : 0), scale);
        const int e = EtDE_share(dd, P, scale, tb ? tb : dd, Hg, ow, tb != nullptr);
        if constexpr (W::NTEAM > 1) w.join();
        return e;
    }
    __device__ __forceinline__ int EtDE_share(const double* dd, double* P, double scale, const double* tb, const double* Hg, bool ow, bool has_tb) {
        MPCQP_SETPRIO(4, 0);
        const int e = EtDE_share_(dd, P, scale, tb, Hg, ow, has_tb);
        MPCQP_SETPRIO(4, 1);
        return e;
    }
    // (tb: a pointer into LDS also when there is no ϵ row to ride -- has_tb says so: a select with nullptr makes the pointer generic)
    __device__ __forceinline__ int EtDE_share_(const double* dd, double* P, double scale, const double* tb, const double* Hg, bool ow, bool has_tb) {
        MPCQP_RELANE(2);
        ow = ow || Hg != nullptr;
        constexpr int NU = DM::nu, NY = DM::ny, NDU = DM::nDU, NYR = DM::nY, SP = DM::sp, RS = DM::rs;
        constexpr int NT = (NDU + 15) / 16, NK = (NYR + 3) / 4;
        const int li = w.lane & 15, lk = w.lane >> 4;
        // the four K rows of an aligned step in the order 0,2,1,3 (see StaticDims::rs); any order works,
        // A and B operands use the same
        const int lkp = (NY % 4 == 0) ? (((lk & 1) << 1) | (lk >> 1)) : lk;
        int offI[NT], offL[NT], jI[NT];
        MPCQP_UNROLL
        for (int I = 0; I < NT; ++I) {
            const int i = 16 * I + li;
            const int j = i / NU, cc = i - j * NU;
            const int tj = jl(i < NDU ? j : 0);        // first step of block column j
            jI[I] = i < NDU ? tj : (1 << 20);          // padding columns never become valid
            offI[I] = -tj * SP + cc;
            offL[I] = offI[I] + lkp * RS;              // + the lane's row inside an aligned K step
        }
        // Tile rows are processed in passes whose accumulators fit the register budget of two
        // waves per SIMD: rows {0,1} together (3 tiles, 3 independent MFMA chains per K step),
        // then every further row on its own.  A pass starts at the first K step that reaches its
        // first block column (E is block lower triangular).
        constexpr int IE = NDU / 16, LE = NDU % 16;     // tile row / lane column of the ϵ row
        // first step whose rows the ϵ row is accumulated for by the pass of its tile row (kfirst() below; known up front: in a
        // team that pass may run on another wavefront)
        int eps_t0 = -1;
        if (DM::neps && has_tb && IE < NT) {
            int v = (jl((16 * IE) / NU) * NY) / 4;
            v = v < NK ? v : NK;
            if ((4 * v) % NY != 0) v = 0;
            eps_t0 = (4 * v) / NY;
        }
        // operands in registers (MPCQP_ETDE_VREG): V[k] = operand of tile column 0 at K step k = (t, a0): Sigma(t - li / nu)[a0 + row, li % nu]
        // (zero blocks in front of the table for t < li / nu)
        constexpr int QK = NY % 4 == 0 ? NY / 4 : 1;                  // K steps per step of the horizon
        constexpr int DJ = (16 % NU == 0) ? QK * (16 / NU) : 1;       // K steps by which a tile column's operands lag those of the one before
        constexpr bool VREG = MPCQP_ETDE_VREG && NY % 4 == 0 && 16 % NU == 0 && DM::zpad > 0 && DM::zpad >= 16 / NU - 1 &&
                              NK <= MPCQP_ETDE_VREG_MAX;
        double V[VREG ? NK : 1];
        if constexpr (VREG) {
            const int cbv = li / NU, civ = li - cbv * NU;
            const double* Vb = S - cbv * SP + lkp * RS + civ;
            MPCQP_UNROLL
            for (int k = 0; k < NK; ++k) V[k] = Vb[(k / QK) * SP + 4 * (k % QK) * RS];
        }
        // (one instantiation per pass: with the pass index a run-time value -- a loop the compiler declines to unroll at
        // seven tile rows -- the accumulator arrays are indexed dynamically and end up in scratch memory)
        auto pass = [&](auto I0c) {
            constexpr int I0 = decltype(I0c)::value;
            if constexpr (W::NTEAM > 1) {                 // pass number (I0 == 0 ? 0 : I0 - 1) mod NTEAM owns the pass
                if constexpr (!team_mine_mfma<W>(I0 == 0 ? 0 : I0 - 1)) return;
            }
            constexpr int MAXT = NT + 1;
            constexpr int I1 = (I0 == 0 && NT > 1) ? 1 : I0;          // last tile row of the pass
            v4d acc[2][MAXT];
            MPCQP_UNROLL
            for (int x = 0; x < 2; ++x) {
                MPCQP_UNROLL
                for (int J = 0; J < MAXT; ++J) acc[x][J] = v4d{0.0, 0.0, 0.0, 0.0};
            }
            // ϵ row (index NDU, when present): row NDU of Phi is sum_r tb[r] E[r,:], i.e. the same
            // contraction with A operand tb instead of E*dd -- rides in its tile row for free
            const bool erow = DM::neps && has_tb && IE >= I0 && IE <= I1;
            // entry (reg) of tile (I, J) held by this lane: row i = 16 I + 4 reg + lk, column ip = 16 J + li; with
            // G = 4 I + reg the packed index pk(i, ip) = 8 G (G + 1) + 16 J + 4 (G + 1) lk + li is linear in the lane's
            // (lk, li) with compile-time coefficients
            auto entry_ok = [&](int I, int J, int reg) {
                const bool eI = erow && I == IE;
                const int il = 4 * reg + lk;                           // row inside the tile
                const bool rowok = 16 * I + 15 < NDU || 16 * I + il < NDU;
                const bool colok = 16 * J + 15 < NDU || 16 * J + li < NDU;
                const bool low = J < I || li <= il;
                return colok && ((rowok && low) || (eI && il == LE));
            };
            auto entry_idx = [&](int I, int J, int reg) {
                const int G = 4 * I + reg;
                return 8 * G * (G + 1) + 16 * J + 4 * (G + 1) * lk + li;
            };
            // H̃ in the accumulator layout, requested now and consumed by the write-back
            // (register-operand form: requested after the K loop instead -- 24 more doubles live next to V, the accumulators and
            //  the row state do not fit the two-waves budget; only block / dense weight handles come this way)
            double hreg[2][MAXT][4];
            auto hload = [&]() {
                MPCQP_UNROLL
                for (int I = I0; I <= I1; ++I) {
                    MPCQP_UNROLL
                    for (int J = 0; J <= I; ++J) {
                        MPCQP_UNROLL
                        for (int reg = 0; reg < 4; ++reg)
                            hreg[I - I0][J][reg] = Hg[entry_ok(I, J, reg) ? (unsigned)entry_idx(I, J, reg) : 0u];
                    }
                }
            };
            // (problems of many tile rows: H̃ is requested in the LAST QUARTER of the K loop instead of in front of it -- up to
            //  NT + 1 tiles of four doubles live next to as many accumulators and the row state of several rows per lane
            //  spilled; the loop is long enough there to cover the loads from its last steps on)
            constexpr bool LATEH = !VREG && NT > 4;
            if (Hg && !VREG && !LATEH) hload();
            // First K step at which tile row I sees a block column that has started (t >= j_l of
            // its first column); the ϵ row needs every step.  The K loop is split at these points
            // so that its bodies are branch-free (accumulators stay in place across iterations).
            auto kfirst = [&](int I) {
                int v = (jl((16 * I) / NU) * NY) / 4;                 // first kk with (4kk+3)/NY >= j_l
                v = v < NK ? v : NK;
                // the ϵ row rides from its tile row's own start when that is a whole number of
                // steps (the few steps before are left to the caller, Et_apply_add), else from 0
                if (erow && I == IE) { if ((4 * v) % NY != 0) v = 0; }
                return v;
            };
            const int kB = (I1 > I0) ? kfirst(I1) : NK;               // second row of the pass joins here
            const int kA = kfirst(I0) < kB ? kfirst(I0) : kB;          // (an earlier start only adds zeros)
            // One K step: operands straight from the Σ table.  A lane whose block column has not
            // started at step t reads the zero slot (no select on the value).  With ny a multiple
            // of 4 the step t of the four K rows is wave-uniform and every row exists.
            // The operands of step kk + 1 are requested before the matrix-core instructions of step kk are issued
            // (software pipeline: a lone wave otherwise sits out one LDS round trip per K step).
            const int zoff = (int)((sm + c.zero) - S);       // the zero slot as an index from block 0 of the table
            struct Ops { double dv, tbv, e[NT]; };
            auto kload = [&](int kk, Ops& o) {
                o.tbv = 0.0;
                if constexpr (NY % 4 == 0) {
                    const int t = (4 * kk) / NY, a0 = 4 * kk - t * NY;         // uniform
                    const int base = t * SP + a0 * RS;
                    o.dv = dd[4 * kk + lkp];
                    if (erow) o.tbv = tb[4 * kk + lkp];
                    MPCQP_UNROLL
                    for (int J = 0; J <= I1; ++J) o.e[J] = S[(DM::zpad || t >= jI[J]) ? base + offL[J] : zoff];
                } else {
                    const int r = 4 * kk + lk;
                    const bool rok = r < NYR;
                    const int rr = rok ? r : 0;
                    const int t = rr / NY, a = rr - t * NY;
                    o.dv = rok ? dd[rr] : 0.0;
                    if (erow) o.tbv = rok ? tb[rr] : 0.0;
                    const int base = t * SP + a * RS;
                    MPCQP_UNROLL
                    for (int J = 0; J <= I1; ++J) o.e[J] = S[(rok && (DM::zpad || t >= jI[J])) ? base + offI[J] : zoff];
                }
            };
            auto kcomp = [&](const Ops& o, bool row0, bool row1) {
                MPCQP_UNROLL
                for (int I = I0; I <= I1; ++I) {
                    if (I == I0 ? !row0 : !row1) continue;
                    const bool eI = erow && I == IE;
                    double ad = o.e[I] * o.dv;
                    if (eI && li == LE) ad = o.tbv;
                    MPCQP_UNROLL
                    for (int J = 0; J <= I; ++J)
                        acc[I - I0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, o.e[J], acc[I - I0][J], 0, 0, 0);
                }
            };
            if constexpr (VREG) {
                // compile-time K ranges (default move blocking; with ny a multiple of 4 a tile row's first K step is a whole
                // number of steps, so the ϵ row rides from there: kfirst() above gives the same values)
                constexpr int kA_ = (DJ * I0 < NK) ? DJ * I0 : NK;
                constexpr int kB_ = (I1 > I0) ? ((DJ * I1 < NK) ? DJ * I1 : NK) : NK;
                constexpr int CH = MPCQP_ETDE_VREG_CHUNK;
                constexpr bool EP = DM::neps != 0 && IE >= I0 && IE <= I1;      // the ϵ row can ride in this pass
                double dvb[2][CH], tbb[2][EP ? CH : 1];
                const double* ddl = dd + lkp;
                const double* tbl = tb + lkp;
                auto ldc = [&](int c0, int buf) {
                    MPCQP_UNROLL
                    for (int u = 0; u < CH; ++u) {
                        const int kk = c0 + u;
                        if (kk >= NK) break;
                        dvb[buf][u] = ddl[4 * kk];
                        if constexpr (EP) { if (erow) tbb[buf][u] = tbl[4 * kk]; }
                    }
                };
                ldc(kA_, 0);
                MPCQP_UNROLL
                for (int c0 = kA_; c0 < NK; c0 += CH) {
                    const int buf = ((c0 - kA_) / CH) & 1;
                    if (c0 + CH < NK) ldc(c0 + CH, buf ^ 1);
                    MPCQP_SCHED_FENCE();
                    MPCQP_UNROLL
                    for (int u = 0; u < CH; ++u) {
                        const int kk = c0 + u;
                        if (kk >= NK) break;
                        MPCQP_UNROLL
                        for (int I = I0; I <= I1; ++I) {
                            if (I > I0 && kk < kB_) continue;
                            const bool eI = erow && I == IE;
                            double ad = V[kk - DJ * I] * dvb[buf][u];
                            if constexpr (EP) { if (eI && li == LE) ad = tbb[buf][u]; }
                            MPCQP_UNROLL
                            for (int J = 0; J <= I; ++J)
                                acc[I - I0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, V[kk - DJ * J], acc[I - I0][J], 0, 0, 0);
                        }
                    }
                    MPCQP_SCHED_FENCE();
                }
            } else {
            // (the request one step past the end reads whatever follows in LDS and is never used)
            constexpr bool PIPE = (NY % 4 == 0) && DM::zpad > 0;
            Ops cur, nxt;
            if (PIPE) kload(kA, cur);
            // (one-row passes run kA .. NK in the first loop; with LATEH it stops three quarters of the way for the H̃ requests)
            const int kB1 = (LATEH && I1 == I0) ? kA + (3 * (kB - kA)) / 4 : kB;
            MPCQP_PRAGMA(unroll MPCQP_ETDE_UNROLL)
            for (int kk = kA; kk < kB1; ++kk) {
                if (PIPE) { kload(kk + 1, nxt); MPCQP_SCHED_FENCE(); kcomp(cur, true, false); MPCQP_SCHED_FENCE(); cur = nxt; }
                else { kload(kk, cur); kcomp(cur, true, false); }
            }
            if (LATEH && I1 == I0) {
                if (Hg) hload();
                MPCQP_SCHED_FENCE();
                MPCQP_PRAGMA(unroll MPCQP_ETDE_UNROLL)
                for (int kk = kB1; kk < kB; ++kk) {
                    if (PIPE) { kload(kk + 1, nxt); MPCQP_SCHED_FENCE(); kcomp(cur, true, false); MPCQP_SCHED_FENCE(); cur = nxt; }
                    else { kload(kk, cur); kcomp(cur, true, false); }
                }
            }
            if (I1 > I0) {
                if (LATEH && Hg) hload();
                MPCQP_PRAGMA(unroll MPCQP_ETDE_UNROLL)
                for (int kk = kB; kk < NK; ++kk) {
                    if (PIPE) { kload(kk + 1, nxt); MPCQP_SCHED_FENCE(); kcomp(cur, true, true); MPCQP_SCHED_FENCE(); cur = nxt; }
                    else { kload(kk, cur); kcomp(cur, true, true); }
                }
            }
            }
            if (Hg && VREG) hload();
            if (ow) {
                // plain stores of H̃ + scale acc (or scale acc alone) on the stored triangle
                MPCQP_UNROLL
                for (int I = I0; I <= I1; ++I) {
                    MPCQP_UNROLL
                    for (int J = 0; J <= I; ++J) {
                        MPCQP_UNROLL
                        for (int reg = 0; reg < 4; ++reg)
                            if (entry_ok(I, J, reg))
                                P[entry_idx(I, J, reg)] = Hg ? fma(scale, acc[I - I0][J][reg], hreg[I - I0][J][reg]) : scale * acc[I - I0][J][reg];
                    }
                }
                return;
            }
            // write-back: every lane does an unconditional read-modify-write; entries outside the
            // stored triangle go to the trash slot (no exec-masked region per entry, so the reads of
            // a tile are issued back to back)
            double* const trash = sm + c.zero + 4;
            MPCQP_UNROLL
            for (int I = I0; I <= I1; ++I) {
                MPCQP_UNROLL
                for (int J = 0; J <= I; ++J) {
                    double* pp_[4];
                    double old_[4];
                    MPCQP_UNROLL
                    for (int reg = 0; reg < 4; ++reg) {
                        pp_[reg] = entry_ok(I, J, reg) ? P + entry_idx(I, J, reg) : trash;
                        old_[reg] = *pp_[reg];
                    }
                    MPCQP_UNROLL
                    for (int reg = 0; reg < 4; ++reg) *pp_[reg] = fma(scale, acc[I - I0][J][reg], old_[reg]);
                }
            }
        };
        etde_passes<0, NT>(pass);
        if constexpr (DM::neps != 0 && W::WV == 0) {
            if (ow && !Hg) {             // (same as below with H̃'s ϵ row: zero off the diagonal; the caller adds 2 C to the diagonal)
                if constexpr (IE >= NT) {
                    for (int k = w.lane; k < NDU; k += WAVE) P[pk(NDU, k)] = 0.0;
                }
                if (w.lane == 0) P[pk(NDU, NDU)] = 0.0;
            } else if (Hg) {
                // nDU a multiple of 16: the ϵ row (index nDU) starts a tile row of its own that no pass covers -- in
                // the overwrite mode nothing else initialises it (the caller adds E'tb to it: eps_t0 stays -1), and the
                // row would keep the previous factor's entries: a wrong Newton matrix, twice the iterations and
                // failed solves on every shape with nu Hc = 16, 32, 48, .. (found at nZ~ = 81, round 3)
                if constexpr (IE >= NT) {
                    for (int k = w.lane; k < NDU; k += WAVE) P[pk(NDU, k)] = Hg[pk(NDU, k)];
                }
                if (w.lane == 0) P[pk(NDU, NDU)] = Hg[pk(NDU, NDU)];     // Ñ's slack weight (construct.jl:842)
            }
        }
        return eps_t0;
    }
#endif

    // P[pk(i,i')] += scale * sum_r E[r,i] dd[r] E[r,i']   (i >= i' < nDU).  When `tb` is given
    // and the matrix-core path runs, the ϵ row P[pk(nDU, i')] += sum_r tb[r] E[r,i'] is added for the
    // rows of the steps t >= the returned value; the caller adds the rest (Et_apply_add; -1: all of it).
    MPCQP_HD_ETDE int EtDE_add(const double* dd, double* P, double scale = 1.0, const double* tb = nullptr,
                          const double* Hg = nullptr, bool ow = false) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DM::is_static) return EtDE_add_mfma(dd, P, scale, tb, Hg, ow);
#endif
        // Strips of four: the packed layout (pk) stores row i as (i/4 + 1) aligned chunks of four
        // columns, chunk s of the whole triangle at P[4s..4s+3].  A lane takes a strip (i, ip0..ip0+3):
        // the product E[r,i] dd[r] is formed once per row r of E and feeds four accumulators.
        const int ny = d.ny, nu = d.nu, nDU = d.nDU;
        const int nstrip = pk_size(nDU) / 4;
        for (int st = w.lane; st < nstrip; st += WAVE) {
            int g = (int)((sqrt(1.0 + 2.0 * st) - 1.0) * 0.5);      // 2g(g+1) <= st < 2(g+1)(g+2)
            while (2 * g * (g + 1) > st) --g;
            while (2 * (g + 1) * (g + 2) <= st) ++g;
            const int rem = st - 2 * g * (g + 1);
            const int i = 4 * g + rem / (g + 1), ip0 = 4 * (rem % (g + 1));
            if (i >= nDU) continue;
            const int j = i / nu, cc = i - j * nu, t0 = jl(j);
            const double* S2[4];
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u) {
                const int ip = ip0 + u <= i ? ip0 + u : i;          // pad columns: any valid one, not stored
                const int j2 = ip / nu, c2 = ip - j2 * nu;
                S2[u] = S + (t0 - jl(j2)) * sp + c2;                // j >= j2  =>  jl[j] >= jl[j2]
            }
            const double* S1 = S + cc;
            const double* dt = dd + t0 * ny;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int t = t0; t < d.Hp; ++t) {
                for (int a = 0; a < ny; ++a) {
                    const double pr = S1[a * rs] * dt[a];
                    MPCQP_UNROLL
                    for (int u = 0; u < 4; ++u) acc[u] = fma(pr, S2[u][a * rs], acc[u]);
                }
                S1 += sp; dt += ny;
                MPCQP_UNROLL
                for (int u = 0; u < 4; ++u) S2[u] += sp;
            }
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u)
                if (ip0 + u <= i) P[4 * st + u] += scale * acc[u];
        }
        return -1;
    }

    // P[pk(i,i')] += scale * sum_t E_t[:,i]' M_t E_t[:,i']  for a block-diagonal weight
    // M_Hp = blkdiag(M_1..M_Hp), Mb = [Hp][ny][ny] (symmetric blocks).  Set-up path only (K2).
    MPCQP_HD void EtMblkE_add(const double* Mb, double* P, double scale) {
        const int ny = d.ny, nu = d.nu, nDU = d.nDU;
        const int ntri = nDU * (nDU + 1) / 2;
        for (int idx = w.lane; idx < ntri; idx += WAVE) {
            int i, ip;
            unpack_idx(idx, i, ip);
            const int j = i / nu, cc = i - j * nu, j2 = ip / nu, c2 = ip - j2 * nu;
            const int t0 = jl(j), off2 = jl(j) - jl(j2);
            double acc = 0.0;
            for (int t = t0; t < d.Hp; ++t) {
                const double* S1 = S + (t - t0) * sp + cc;
                const double* S2 = S + (t - t0 + off2) * sp + c2;
                const double* Mt = Mb + (size_t)t * ny * ny;
                for (int a = 0; a < ny; ++a) {
                    double ms = 0.0;
                    for (int a2 = 0; a2 < ny; ++a2) ms += Mt[a + ny * a2] * S2[a2 * rs];
                    acc += S1[a * rs] * ms;
                }
            }
            P[pk(i, ip)] += scale * acc;
        }
    }

    // E[(t,a),(j,cc)] = Σ(t - j_j)[a,cc] for t >= j_j, else 0   (transcription.jl:134-139)
    MPCQP_HD double Eat(int t, int a, int j, int cc) const {
        const int t0 = jl(j);
        return t >= t0 ? S[(t - t0) * sp + a * rs + cc] : 0.0;
    }

    // P[pk(i,i')] += scale * E[:,i]' M E[:,i'] for a DENSE symmetric M_Hp (nY x nY, column-major), one column
    // i' at a time: v = M E[:,i'] into `tmp` (nY doubles of LDS), then the dot products with E[:,i], i >= i'.
    // Set-up path only (K2): nΔU (nY² + nY nΔU) multiply-adds.
    MPCQP_HD void EtMfullE_add(const double* Mf, double* tmp, double* P, double scale) {
        const int ny = d.ny, nu = d.nu, nDU = d.nDU, nY = d.nY;
        for (int ip = 0; ip < nDU; ++ip) {
            const int j2 = ip / nu, c2 = ip - j2 * nu;
            for (int r = w.lane; r < nY; r += WAVE) {
                double acc = 0.0;
                for (int t = jl(j2); t < d.Hp; ++t)
                    for (int a = 0; a < ny; ++a) acc += Mf[r + (size_t)nY * (t * ny + a)] * Eat(t, a, j2, c2);
                tmp[r] = acc;
            }
            w.sync();
            for (int i = ip + w.lane; i < nDU; i += WAVE) {
                const int j = i / nu, cc = i - j * nu;
                double acc = 0.0;
                for (int t = jl(j); t < d.Hp; ++t)
                    for (int a = 0; a < ny; ++a) acc += Eat(t, a, j, cc) * tmp[t * ny + a];
                P[pk(i, ip)] += scale * acc;
            }
            w.sync();
        }
    }

    // ex̂[i,(j,c)]
    MPCQP_HD double Xat(int i, int k) const {
        const int j = k / d.nu, cc = k - j * d.nu;
        return sm[c.exT + (j * d.nxh + i) * d.nu + cc];
    }
};

// ------------------------------------------------------------------------------------------
// K1: prediction tables of one problem.  LDS: see predmat_lds_doubles().
// ------------------------------------------------------------------------------------------
MPCQP_HD inline int predmat_lds_doubles(const Dims& d) {
    int nx = d.nxh;
    return nx * nx * 3 + nx * d.nu * 2 + d.ny * nx * 3 + nx * 2 + nx * (d.nd > 0 ? d.nd : 1) * 2 + 8;
}

template <class W>
MPCQP_HD void predmat_body(W& w, const Dims& d, const Model& m, int b, double* sm, bool terminal) {
    const int nx = d.nxh, nu = d.nu, ny = d.ny, nd = d.nd, Hp = d.Hp;
    double* A = sm;                         // A[i + nx*k]
    double* T0 = A + nx * nx;               // matrix power ping
    double* T1 = T0 + nx * nx;              // pong
    double* W0 = T1 + nx * nx;              // S(m) B̂u ping  W[i + nx*c]
    double* W1 = W0 + nx * nu;
    double* Cm = W1 + nx * nu;              // Ĉ  C[a + ny*k]
    double* P0 = Cm + ny * nx;              // Ĉ Â^t ping  P[a + ny*k]
    double* P1 = P0 + ny * nx;
    double* v0 = P1 + ny * nx;              // S(t) dop ping
    double* v1 = v0 + nx;
    double* X0 = v1 + nx;                   // Â^m B̂d ping  X[i + nx*e]
    double* X1 = X0 + nx * (nd > 0 ? nd : 1);
    const double* gA = m.Ahat + (size_t)b * nx * nx;
    const double* gB = m.Bu + (size_t)b * nx * nu;
    const double* gC = m.C + (size_t)b * ny * nx;
    for (int i = w.lane; i < nx * nx; i += WAVE) A[i] = gA[i];
    for (int i = w.lane; i < nx * nu; i += WAVE) W0[i] = gB[i];
    for (int i = w.lane; i < ny * nx; i += WAVE) Cm[i] = gC[i];
    for (int i = w.lane; i < nx; i += WAVE) v0[i] = m.dop ? m.dop[(size_t)b * nx + i] : 0.0;
    if (nd > 0)
        for (int i = w.lane; i < nx * nd; i += WAVE) X0[i] = m.Bd[(size_t)b * nx * nd + i];
    w.sync();
    // P0 = Ĉ Â
    for (int i = w.lane; i < ny * nx; i += WAVE) {
        int a = i % ny, k = i / ny;
        double acc = 0.0;
        for (int l = 0; l < nx; ++l) acc += Cm[a + ny * l] * A[l + nx * k];
        P0[i] = acc;
    }
    if (terminal)
        for (int i = w.lane; i < nx * nx; i += WAVE) T0[i] = A[i];
    w.sync();
    double* Stab = m.Stab + (size_t)b * Hp * ny * nu;
    double* Ktab = m.Ktab + (size_t)b * nx * d.nY;
    double* Bvec = m.Bvec + (size_t)b * d.nY;
    double* Gd = nd > 0 ? m.Gdtab + (size_t)b * Hp * ny * nd : nullptr;
    for (int t = 0; t < Hp; ++t) {
        // --- emit tables of step t -------------------------------------------------------
        for (int i = w.lane; i < ny * nu; i += WAVE) {          // Σ_t = Ĉ W_t
            int a = i / nu, cc = i - a * nu;
            double acc = 0.0;
            for (int l = 0; l < nx; ++l) acc += Cm[a + ny * l] * W0[l + nx * cc];
            Stab[(t * ny + a) * nu + cc] = acc;
        }
        for (int i = w.lane; i < ny * nx; i += WAVE) {          // K block t = Ĉ Â^{t+1}
            int a = i % ny, k = i / ny;
            Ktab[(size_t)k * d.nY + t * ny + a] = P0[i];
        }
        for (int a = w.lane; a < ny; a += WAVE) {               // B block t = Ĉ S(t) dop
            double acc = 0.0;
            for (int l = 0; l < nx; ++l) acc += Cm[a + ny * l] * v0[l];
            Bvec[t * ny + a] = acc;
        }
        if (nd > 0)
            for (int i = w.lane; i < ny * nd; i += WAVE) {      // Ĉ Â^t B̂d
                int a = i / nd, e = i - a * nd;
                double acc = 0.0;
                for (int l = 0; l < nx; ++l) acc += Cm[a + ny * l] * X0[l + nx * e];
                Gd[(t * ny + a) * nd + e] = acc;
            }
        if (terminal) {
            // ex̂ block j = S(Hp - j_j - 1) B̂u = W_t when t == Hp - j_j - 1
            for (int j = 0; j < d.Hc; ++j)
                if (t == Hp - m.jl[j] - 1) {
                    double* ex = m.exT + ((size_t)b * d.Hc + j) * nx * nu;
                    for (int i = w.lane; i < nx * nu; i += WAVE) {
                        int r = i / nu, cc = i - r * nu;
                        ex[i] = W0[r + nx * cc];
                    }
                }
            if (nd > 0) {
                double* Xd = m.Xdtab + ((size_t)b * Hp + t) * nx * nd;
                for (int i = w.lane; i < nx * nd; i += WAVE) {
                    int r = i / nd, e = i - r * nd;
                    Xd[i] = X0[r + nx * e];
                }
            }
            if (t == Hp - 1) {
                for (int i = w.lane; i < nx; i += WAVE) m.bxv[(size_t)b * nx + i] = v0[i];
                // T0 holds Â^{t+1} = Â^Hp
                for (int i = w.lane; i < nx * nx; i += WAVE) m.kxT[(size_t)b * nx * nx + i] = T0[i];
            }
        }
        if (t == Hp - 1) break;
        // --- advance recursions ----------------------------------------------------------
        for (int i = w.lane; i < nx * nu; i += WAVE) {          // W_{t+1} = Â W_t + B̂u
            int r = i % nx, cc = i / nx;
            double acc = gB[i];
            for (int l = 0; l < nx; ++l) acc += A[r + nx * l] * W0[l + nx * cc];
            W1[i] = acc;
        }
        for (int i = w.lane; i < ny * nx; i += WAVE) {          // P_{t+1} = P_t Â
            int a = i % ny, k = i / ny;
            double acc = 0.0;
            for (int l = 0; l < nx; ++l) acc += P0[a + ny * l] * A[l + nx * k];
            P1[i] = acc;
        }
        for (int r = w.lane; r < nx; r += WAVE) {               // v_{t+1} = Â v_t + dop
            double acc = m.dop ? m.dop[(size_t)b * nx + r] : 0.0;
            for (int l = 0; l < nx; ++l) acc += A[r + nx * l] * v0[l];
            v1[r] = acc;
        }
        if (nd > 0)
            for (int i = w.lane; i < nx * nd; i += WAVE) {      // X_{t+1} = Â X_t
                int r = i % nx, e = i / nx;
                double acc = 0.0;
                for (int l = 0; l < nx; ++l) acc += A[r + nx * l] * X0[l + nx * e];
                X1[i] = acc;
            }
        if (terminal)
            for (int i = w.lane; i < nx * nx; i += WAVE) {      // T_{t+1} = T_t Â
                int r = i % nx, k = i / nx;
                double acc = 0.0;
                for (int l = 0; l < nx; ++l) acc += T0[r + nx * l] * A[l + nx * k];
                T1[i] = acc;
            }
        w.sync_lds();          // (the recursion lives in LDS; the tables streamed to HBM above are never read back here)
        { double* t_ = W0; W0 = W1; W1 = t_; }
        { double* t_ = P0; P0 = P1; P1 = t_; }
        { double* t_ = v0; v0 = v1; v1 = t_; }
        { double* t_ = X0; X0 = X1; X1 = t_; }
        { double* t_ = T0; T0 = T1; T1 = t_; }
    }
}

// ------------------------------------------------------------------------------------------
// K1 on the matrix cores (nx̂ <= 16, ny <= 16, nu + 1 + nd <= 16): the recursions of init_predmat
// (transcription.jl:115-194) are chains of small dense products with one common factor,
//   [W | v | X](t+1) = Â [W | v | X](t) + [B̂u | dop | 0]      W = S(t) B̂u, v = S(t) dop, X = Â^t B̂d
//   P'(t+1) = Â' P'(t)  (P = Ĉ Â^(t+1)),   T'(t+1) = Â' T'(t)  (T = Â^(t+1), terminal rows only)
//   [Σ_t | B_t | Gd_t] = Ĉ [W | v | X](t)
// i.e. 16 x 16 x 16 tiles = four v_mfma_f64_16x16x4 each, the constant factor as operand A.  The contraction
// index is ordered so that K step kk covers the rows lk + 4 kk (lk = lane >> 4): the operand B of K step kk is then
// register kk of the previous product's accumulator -- the state never leaves the accumulator registers: no LDS, no
// barrier, no lane traffic; a wavefront's only memory traffic are the model (once) and the tables it emits.
// Replaces the LDS loops of predmat_body below for these shapes (2.1 ms of dependent LDS round trips per 65536
// controllers at C3, occupancy-bound: the time was inversely proportional to the resident wavefronts).
// ------------------------------------------------------------------------------------------
MPCQP_HD inline bool predmat_mfma_ok(const Dims& d) {
    return d.nxh <= 16 && d.ny <= 16 && d.nu + 1 + d.nd <= 16;
}
#if defined(__HIPCC__)
__device__ __forceinline__ void predmat_mfma(int lane, const Dims& d, const Model& m, int b, bool terminal) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef double v4d_ __attribute__((ext_vector_type(4)));
    const int nx = d.nxh, nu = d.nu, ny = d.ny, nd = d.nd, Hp = d.Hp;
    const int li = lane & 15, lk = lane >> 4;
    const double* gA = m.Ahat + (size_t)b * nx * nx;          // A[i + nx*k]
    const double* gB = m.Bu + (size_t)b * nx * nu;            // B[i + nx*c]
    const double* gC = m.C + (size_t)b * ny * nx;             // C[a + ny*k]
    const double* gD = m.dop ? m.dop + (size_t)b * nx : nullptr;
    const double* gBd = nd > 0 ? m.Bd + (size_t)b * nx * nd : nullptr;
    double aA[4], aAt[4], aC[4];
    v4d_ Wf, Cw, Pt, Tt;
    MPCQP_UNROLL
    for (int kk = 0; kk < 4; ++kk) {
        const int k = lk + 4 * kk;                            // contraction index / state row of this (lane, register)
        const bool in = li < nx && k < nx;
        aA[kk] = in ? gA[li + nx * k] : 0.0;                  // Â[li][k]
        aAt[kk] = in ? gA[k + nx * li] : 0.0;                 // Â'[li][k]
        aC[kk] = (li < ny && k < nx) ? gC[li + ny * k] : 0.0; // Ĉ[li][k]
        double c0 = 0.0, w0 = 0.0;
        if (k < nx) {
            if (li < nu) c0 = w0 = gB[k + nx * li];
            else if (li == nu) c0 = w0 = gD ? gD[k] : 0.0;
            else if (li <= nu + nd) w0 = gBd[k + nx * (li - nu - 1)];
        }
        Cw[kk] = c0; Wf[kk] = w0;
        Tt[kk] = aA[kk];                                      // T'(0)[k][li] = Â[li][k]
    }
    {   // P'(0) = Â' Ĉ'   (operand B: Ĉ'[k][li] = Ĉ[li][k] = aC)
        v4d_ acc = {0.0, 0.0, 0.0, 0.0};
        MPCQP_UNROLL
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aAt[kk], aC[kk], acc, 0, 0, 0);
        Pt = acc;
    }
    double* Stab = m.Stab + (size_t)b * Hp * ny * nu;
    double* Ktab = m.Ktab + (size_t)b * nx * d.nY;
    double* Bvec = m.Bvec + (size_t)b * d.nY;
    double* Gd = nd > 0 ? m.Gdtab + (size_t)b * Hp * ny * nd : nullptr;
    for (int t = 0; t < Hp; ++t) {
        // [Σ_t | B_t | Gd_t] = Ĉ [W | v | X](t): register r holds row a = lk + 4 r, column li
        v4d_ sg = {0.0, 0.0, 0.0, 0.0};
        MPCQP_UNROLL
        for (int kk = 0; kk < 4; ++kk) sg = __builtin_amdgcn_mfma_f64_16x16x4f64(aC[kk], Wf[kk], sg, 0, 0, 0);
        // the recursions (independent chains: issued together)
        v4d_ wn = Cw, pn = {0.0, 0.0, 0.0, 0.0}, tn = {0.0, 0.0, 0.0, 0.0};
        if (t + 1 < Hp) {
            MPCQP_UNROLL
            for (int kk = 0; kk < 4; ++kk) {
                wn = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[kk], Wf[kk], wn, 0, 0, 0);
                pn = __builtin_amdgcn_mfma_f64_16x16x4f64(aAt[kk], Pt[kk], pn, 0, 0, 0);
            }
            if (terminal) {
                MPCQP_UNROLL
                for (int kk = 0; kk < 4; ++kk) tn = __builtin_amdgcn_mfma_f64_16x16x4f64(aAt[kk], Tt[kk], tn, 0, 0, 0);
            }
        }
        // ---- tables of step t
        MPCQP_UNROLL
        for (int r = 0; r < 4; ++r) {
            const int a = lk + 4 * r;           // output row of sg; state row k of Wf / Pt / Tt
            if (a < ny) {
                if (li < nu) Stab[(t * ny + a) * nu + li] = sg[r];
                else if (li == nu) Bvec[t * ny + a] = sg[r];
                else if (li <= nu + nd) Gd[(t * ny + a) * nd + (li - nu - 1)] = sg[r];
            }
            // K block t = P(t): P[li][k = a].  (32-byte pieces at a stride of nY doubles; collecting several steps in LDS for
            // longer pieces was measured and is slower, 1.13 vs 0.95 ms at C3: the stores are a third of the time by their
            // volume -- 1.0 GB of K per 65536 controllers -- not by their shape.)
            if (li < ny && a < nx) Ktab[(size_t)a * d.nY + t * ny + li] = Pt[r];
        }
        if (terminal) {
            for (int j = 0; j < d.Hc; ++j)
                if (t == Hp - m.jl[j] - 1) {                  // ex̂ block j = S(Hp - j_j - 1) B̂u = W(t)
                    double* ex = m.exT + ((size_t)b * d.Hc + j) * nx * nu;
                    MPCQP_UNROLL
                    for (int r = 0; r < 4; ++r)
                        if (li < nu && lk + 4 * r < nx) ex[(lk + 4 * r) * nu + li] = Wf[r];
                }
            if (nd > 0) {
                double* Xd = m.Xdtab + ((size_t)b * Hp + t) * nx * nd;
                MPCQP_UNROLL
                for (int r = 0; r < 4; ++r)
                    if (li > nu && li <= nu + nd && lk + 4 * r < nx) Xd[(lk + 4 * r) * nd + (li - nu - 1)] = Wf[r];
            }
            if (t == Hp - 1) {
                MPCQP_UNROLL
                for (int r = 0; r < 4; ++r) {
                    const int k = lk + 4 * r;
                    if (li == nu && k < nx) m.bxv[(size_t)b * nx + k] = Wf[r];
                    if (li < nx && k < nx) m.kxT[(size_t)b * nx * nx + li + nx * k] = Tt[r];      // T[li][k] = Â^Hp
                }
            }
        }
        Wf = wn; Pt = pn; Tt = tn;
    }
#endif
}
#endif

// ------------------------------------------------------------------------------------------
// K2: H̃ = 2(Ẽ'MẼ + P̃Δu'ÑP̃Δu + P̃u'LP̃u), diagonal weights, packed lower triangle.
// ------------------------------------------------------------------------------------------
template <class W, class DM>
MPCQP_HD void hessian_body(W& w, const DM& d, const Model& m, int b, double* sm) {
    Qp<W, DM> qp(w, d, m, b, sm);
    qp.load_tables();
    double* P = qp.Phi;
    double* tY = sm + qp.c.tA[P_Y];
    for (int i = w.lane; i < d.npk; i += WAVE) P[i] = 0.0;
    if (!m.Mblk && !m.Mfull)
        for (int i = w.lane; i < d.nY; i += WAVE) tY[i] = m.Mdiag[(size_t)b * d.nY + i];
    w.sync();
    if (m.Mfull) qp.EtMfullE_add(m.Mfull + (size_t)b * d.nY * d.nY, tY, P, 2.0);
    else if (m.Mblk) qp.EtMblkE_add(m.Mblk + (size_t)b * d.Hp * d.ny * d.ny, P, 2.0);
    else qp.EtDE_add(tY, P, 2.0);                               // 2 E'ME
    w.sync();
    const int nu = d.nu;
    // 2 Pu'L Pu: entry ((j,c),(j',c)) = sum_{t >= j_max(j,j')} L[t,c]   (construct.jl:797-806)
    // (the suffix sum of a (block, channel) is formed once, by its own lane, and added along the
    // lane's row of the lower triangle: entry (i, i') with j >= j' takes the sum from j_j on)
    const double* L = m.Ldiag + (size_t)b * d.nU;
    for (int i = w.lane; i < d.nDU; i += WAVE) {
        const int j = i / nu, cc = i - j * nu;
        if (m.Ldense) {        // dense L_Hp: entry ((j,c),(j',c')) = sum_{t >= j_j} sum_{t' >= j_j'} L[(t,c),(t',c')]
            const double* Lf = m.Ldense + (size_t)b * d.nU * d.nU;
            for (int ip = 0; ip <= i; ++ip) {
                const int j2 = ip / nu, c2 = ip - j2 * nu;
                double acc = 0.0;
                for (int t = qp.jl(j); t < d.Hp; ++t)
                    for (int t2 = qp.jl(j2); t2 < d.Hp; ++t2) acc += Lf[(t * nu + cc) + (size_t)d.nU * (t2 * nu + c2)];
                P[pk(i, ip)] += 2.0 * acc;
            }
        } else {
            double suf = 0.0;
            for (int t = qp.jl(j); t < d.Hp; ++t) suf += L[t * nu + cc];
            for (int j2 = 0; j2 <= j; ++j2) P[pk(i, j2 * nu + cc)] += 2.0 * suf;
        }
        if (m.Ndense) {        // dense N_Hc
            const double* Nf = m.Ndense + (size_t)b * d.nDU * d.nDU;
            for (int ip = 0; ip <= i; ++ip) P[pk(i, ip)] += 2.0 * Nf[i + (size_t)d.nDU * ip];
        } else {
            P[pk(i, i)] += 2.0 * m.Ndiag[(size_t)b * d.nDU + i];    // 2 N
        }
    }
    if (d.neps && w.lane == 0) P[pk(d.nZ - 1, d.nZ - 1)] = 2.0 * m.Cwt[b];    // Ñ = blkdiag(N, C)
    w.sync();
    double* H = m.Hpk + (size_t)b * d.npk;
    for (int i = w.lane; i < d.npk; i += WAVE) H[i] = P[i];
}

// ------------------------------------------------------------------------------------------
// Per-row interior-point state.  A row (group g, local index k) is owned by lane k % 64, and so
// is primitive k of its pair: every access to row state is lane-local.  Runtime dims keep the
// arrays in LDS; compile-time dims keep them in registers (slots resolved by full unrolling).
// ------------------------------------------------------------------------------------------
struct Row {
    double &h, &s, &lam, &rp, &gd, &pp;
    double cs;       // softness of the row (stored with runtime dims, re-derived otherwise)
    double wt = 1.0; // multiplicity of the row in the barrier (re-derived on every pass, see rowweight())
    double* wic = nullptr;   // compile-time dims: register that keeps 1 / (s + δ lam) of the current iterate (row_wi_cached)
};

template <class DM, bool STATIC = DM::is_static>
struct RowStore;

template <class DM>
struct RowStore<DM, false> {
    double* a[NROWARR];
    const DM& d;
    int lane;
    MPCQP_HD RowStore(const DM& d_, double* sm, const Carve& c, int lane_) : d(d_), lane(lane_) {
        for (int i = 0; i < NROWARR; ++i) a[i] = sm + c.rows[i];
    }
    MPCQP_HD int qmax(int g) const { return (d.cnt(g >> 1) + WAVE - 1) / WAVE; }
    MPCQP_HD Row at(int g, int q) {
        const int r = d.rowoff(g) + lane + WAVE * q;
        return Row{a[0][r], a[1][r], a[2][r], a[3][r], a[4][r], a[5][r], a[6][r]};
    }
    MPCQP_HD void set_cs(int g, int q, double v) { a[6][d.rowoff(g) + lane + WAVE * q] = v; }
    static constexpr bool stores_cs = true;
};

template <class DM>
struct RowStore<DM, true> {
    MPCQP_HD static constexpr int qmax(int g) { return (DM::cnt(g >> 1) + WAVE - 1) / WAVE; }
    MPCQP_HD static constexpr int slotoff(int g) {
        int o = 0;
        for (int i = 0; i < g; ++i)
            if ((DM::gmask >> i) & 1u) o += qmax(i);
        return o;
    }
    static constexpr int NSLOT = slotoff(NGROUP) > 0 ? slotoff(NGROUP) : 1;
    double a[NROWARR - 1][NSLOT];
    double wi_[NSLOT];
    MPCQP_HD RowStore(const DM&, double*, const Carve&, int) {}
    MPCQP_HD Row at(int g, int q) {
        const int r = slotoff(g) + q;
        return Row{a[0][r], a[1][r], a[2][r], a[3][r], a[4][r], a[5][r], 0.0, 1.0, &wi_[r]};
    }
    MPCQP_HD void set_cs(int, int, double) {}
    static constexpr bool stores_cs = false;
};

// ------------------------------------------------------------------------------------------
// K3: one control period of one controller.
// ------------------------------------------------------------------------------------------
template <class W, class DM>
struct Step {
    Qp<W, DM>& qp;
    W& w;
    const DM& d;
    const Model& m;
    const int b;
    double* sm;
    const Carve& c;
    RowStore<DM> rows;
    double *z, *dz, *q, *zlo, *zhi, *gt, *rd, *F, *Phi;
    int mact;           // number of finite rows
    double wsum;        // their total multiplicity (rowweight), the m of mu = s'lam / m
    double nh;          // 1 + max |h|
    double delta;
    double prof_[16] = {0};   // phase cycle counters (profiling builds)
    double myinvd = 0.0;      // 1/L[lane][lane] of the current factor (LDL' form: 1/d[lane])
    double myd_ = 0.0;        // LDL' form: the lane's pivot d[lane] while the factorisation runs
    bool chol_broke = false;  // the last factorisation met a pivot below its threshold (wave-uniform)
    // H̃ = 2(E'M E + N + Pu'L Pu) (+) 2C (construct.jl:837-845) with DIAGONAL weights never has to be read by a step: its
    // terms have the structure of G'D G -- 2 M_r joins the factor of the Ŷ row r in E'(D + 2M)E, 2 sum_{t in block} L_t the
    // factor of the block's U row, 2 N_k (2 C for the slack) the diagonal -- and H̃ z is two structured products with the
    // Σ table already in LDS.  fold_H: this handle takes that route (no packed H̃ from global memory in the iteration: 24
    // loads per lane and factorisation with a drained vmcnt in front of the matrix-core loop, and the strided re-read
    // of the 861 entries for every H̃ z of the polish).  Block / dense weight matrices keep the packed H̃.
    bool fold_H = false;
    bool h_Lnz = false;       // some L weight of this controller is non-zero
    double h2n_ = 0.0, h2l_ = 0.0;      // one row per lane: lane k's 2 N_k (2 C on the slack's lane) and 2 sum_{t in block(k)} L_t
    // 2 M_r of this lane's Ŷ rows r = lane + 64 q (one row per lane: read once in init_fold_H; the assembly asked global memory
    // for them in every factorisation, one conditional block and one drained vmcnt per row)
    static constexpr int HWQ = hwq_<DM>();
    double hwy_[HWQ > 0 ? HWQ : 1] = {0.0};
    MPCQP_HD static constexpr bool hwy_cached() { return hwq_<DM>() > 0; }
    MPCQP_HD double hwy(int k) const {            // k = lane + 64 q
        double v = hwy_[0];
        MPCQP_UNROLL
        for (int q_ = 1; q_ < HWQ; ++q_) v = (k >= WAVE * q_) ? hwy_[q_] : v;
        return v;
    }

    MPCQP_HD Step(Qp<W, DM>& qp_)
        : qp(qp_), w(qp_.w), d(qp_.d), m(qp_.m), b(qp_.b), sm(qp_.sm), c(qp_.c),
          rows(qp_.d, qp_.sm, qp_.c, qp_.w.lane) {
        z = sm + c.z; dz = sm + c.dz; q = sm + c.q; zlo = sm + c.zlo; zhi = sm + c.zhi;
        gt = sm + c.gt; rd = sm + c.rd; F = sm + c.F;
        Phi = qp.Phi;
        delta = d.dual_reg;
    }

    // reference default softness: 0 for u and Δu, 1 for y and x̂end (construct.jl:909-913)
    MPCQP_HD double soft_init(int g, int k) const {
        if (!d.neps) return 0.0;
        const double* p = nullptr;
        double def = 0.0;
        switch (g) {
            case 2 * P_U: p = m.C_umin; break;
            case 2 * P_U + 1: p = m.C_umax; break;
            case 2 * P_DU: p = m.C_dumin; break;
            case 2 * P_DU + 1: p = m.C_dumax; break;
            case 2 * P_Y: p = m.C_ymin; def = 1.0; break;
            case 2 * P_Y + 1: p = m.C_ymax; def = 1.0; break;
            case 2 * P_X: p = m.c_x0min; def = 1.0; break;
            case 2 * P_X + 1: p = m.c_x0max; def = 1.0; break;
            case 2 * P_W: p = m.C_wmin; def = 1.0; break;
            case 2 * P_W + 1: p = m.C_wmax; def = 1.0; break;
            default: return 0.0;
        }
        if ((g >> 1) == P_Y && k >= d.nY) return 1.0;      // the hosted row -eps <= 0 (eps_host_group)
        if (!p) return def;
        if ((g >> 1) == P_U) {       // one row per (block, channel): softness of the block's first step
            const int j = k / d.nu, cc = k - j * d.nu;
            return p[(size_t)b * d.nU + qp.jl(j) * d.nu + cc];
        }
        return p[(size_t)b * d.len(g >> 1) + k];
    }

    // Multiplicity of a row in the barrier.  The U rows of one move-blocking interval are merged
    // into one (see build()); the reference's QP holds nb_j identical copies of it, and k copies of
    // a row act on an interior-point iteration like one row whose complementarity target is k mu
    // (same Phi, same step lengths).  Carrying that weight keeps the merged problem on the
    // reference problem's central path: measured on 8192 C3 instances, 13.5 instead of 14.4
    // iterations (the last interval of C3 holds 21 steps).  The optimum does not depend on it.
    MPCQP_HD double rowweight(int p, int k) const {
        if (p != P_U) return 1.0;
        const int j = k / d.nu;
        return (double)((j + 1 < d.Hc ? qp.jl(j + 1) : d.Hp) - qp.jl(j));
    }

    // fn(group, local index, Row&) for every row owned by this lane
    template <class Fn>
    MPCQP_HD void for_rows(Fn fn) {
        MPCQP_RELANE(7);
        MPCQP_UNROLL
        for (int g = 0; g < NGROUP; ++g) {
            if (!qp.group_on(g)) continue;
            const int n = d.cnt(g >> 1);
            MPCQP_UNROLL
            for (int qq = 0; qq < rows.qmax(g); ++qq) {
                const int k = w.lane + WAVE * qq;
                if (k < n) {
                    Row r = rows.at(g, qq);
                    if (!RowStore<DM>::stores_cs) r.cs = soft_init(g, k);
                    r.wt = rowweight(g >> 1, k);
                    fn(g, k, r);
                }
                MPCQP_SCHED_FENCE();
            }
        }
    }

    // fn(pair, k, rmin*, rmax*) for every primitive owned by this lane (null = group off)
    template <class Fn>
    MPCQP_HD void for_pairs(Fn fn) {
        MPCQP_UNROLL
        for (int p = 0; p < NPAIR; ++p) {
            if (!qp.pair_on(p)) continue;
            const int n = d.cnt(p);
            const bool gmin = qp.group_on(2 * p), gmax = qp.group_on(2 * p + 1);
            MPCQP_UNROLL
            for (int qq = 0; qq < rows.qmax(2 * p); ++qq) {
                const int k = w.lane + WAVE * qq;
                if (k < n) {
                    const bool live = !RowStore<DM>::stores_cs;
                    if (gmin && gmax) {
                        Row r0 = rows.at(2 * p, qq), r1 = rows.at(2 * p + 1, qq);
                        if (live) { r0.cs = soft_init(2 * p, k); r1.cs = soft_init(2 * p + 1, k); }
                        r0.wt = r1.wt = rowweight(p, k);
                        fn(p, k, &r0, &r1);
                    } else if (gmin) {
                        Row r0 = rows.at(2 * p, qq);
                        if (live) r0.cs = soft_init(2 * p, k);
                        r0.wt = rowweight(p, k);
                        fn(p, k, &r0, (Row*)nullptr);
                    } else {
                        Row r1 = rows.at(2 * p + 1, qq);
                        if (live) r1.cs = soft_init(2 * p + 1, k);
                        r1.wt = rowweight(p, k);
                        fn(p, k, (Row*)nullptr, &r1);
                    }
                }
                MPCQP_SCHED_FENCE();
            }
        }
    }

    MPCQP_HD static bool fin(const Row& r) { return r.h < BIG; }

    // F_w of linconstraint_custom! (execute.jl:337-364), row k = (t, i), t = 0..Hp:
    //   Wy [ŷ(k); F + Yop] + Wu [Tu u(k-1) + Uop; u(k-1)] + Wd [d(k); D̂] + Wr [ry(k); R̂y]
    // in deviation variables plus the operating-point part w_op = Wy yop + Wu uop + Wd dop + Wr yop.
    // ŷ0(k) = Ĉ x̂0 + D̂d d0 (evaloutput); r̂e(k) = ry(k) (Model::ry_now, mpcqp_set_current_setpoint; without it
    // the first block of R̂y, which is the same thing for the reference's default R̂y = repeat(ry)).
    MPCQP_HD double Fw_at(int k, const StepIO& io, const double* x0, const double* lu) const {
        const int nw = d.nw, ny = d.ny, nu = d.nu, nd = d.nd, nx = d.nxh;
        const int t = k / nw, i = k - t * nw;
        const bool rconst = d.flags & 1u;
        double acc = m.w_op ? m.w_op[(size_t)b * nw + i] : 0.0;
        for (int a = 0; a < ny; ++a) {
            double ye;
            if (t == 0) {
                const double* Cm = m.C + (size_t)b * ny * nx;          // (ny,nx̂) column-major
                ye = 0.0;
                for (int kk = 0; kk < nx; ++kk) ye += Cm[a + ny * kk] * x0[kk];
                if (nd > 0) {
                    const double* Dd = m.Dd + (size_t)b * ny * nd;
                    for (int e = 0; e < nd; ++e) ye += Dd[a + ny * e] * io.d0[(size_t)b * nd + e];
                }
            } else {
                ye = F[(t - 1) * ny + a];
            }
            acc += qp.Wy_(i, a) * ye;
            if (m.Wr) {
                // r̂e(k) is the CURRENT set point ry(k) (execute.jl:351, R̂e_term[1:ny] .= mpc.ry), the rest R̂y
                const int tr = t == 0 ? 0 : t - 1;
                const double re = rconst ? io.Ry[(size_t)b * ny + a]
                                  : (t == 0 && m.ry_now) ? m.ry_now[(size_t)b * ny + a]
                                                         : io.Ry[(size_t)b * d.nY + tr * ny + a];
                acc += m.Wr[(size_t)b * nw * ny + i + nw * a] * re;
            }
        }
        for (int cc = 0; cc < nu; ++cc) acc += qp.Wu_(i, cc) * lu[cc];
        if (m.Wd && nd > 0) {
            for (int e = 0; e < nd; ++e) {
                const double de = t == 0 ? io.d0[(size_t)b * nd + e] : io.Dhat0[(size_t)b * d.nD + (t - 1) * nd + e];
                acc += m.Wd[(size_t)b * nw * nd + i + nw * e] * de;
            }
        }
        return acc;
    }

    // ---- free response, gradient, right-hand sides (initpred!, linconstraint!) -------------
    MPCQP_HD void build(const StepIO& io) {
        const int nx = d.nxh, nu = d.nu, ny = d.ny, nd = d.nd, nY = d.nY;
        const double* x0 = sm + c.xh;                        // staged by step_body (after the optional correction)
        const double* lu = io.lastu0 + (size_t)b * nu;
        const double* K = m.Ktab + (size_t)b * nx * nY;
        const double* Bv = m.Bvec + (size_t)b * nY;
        // F = B + K x̂0 + V lastu0 (+ G d0 + J D̂0)           execute.jl:249-255
        for (int r = w.lane; r < nY; r += WAVE) {
            const int t = r / ny, a = r - t * ny;
            double acc = Bv[r];
            {
                // (the row's entries of K in batches of eight loads: see QP::stage)
                int k = 0;
                for (; k + 8 <= nx; k += 8) {
                    double kv[8];
                    MPCQP_UNROLL
                    for (int u = 0; u < 8; ++u) kv[u] = K[(size_t)(k + u) * nY + r];
                    MPCQP_UNROLL
                    for (int u = 0; u < 8; ++u) acc += kv[u] * x0[k + u];
                }
                for (; k < nx; ++k) acc += K[(size_t)k * nY + r] * x0[k];
            }
            const double* Sb = qp.S + t * qp.sp + a * qp.rs;    // V block t = Σ_t
            for (int cc = 0; cc < nu; ++cc) acc += Sb[cc] * lu[cc];
            if (nd > 0) {
                const double* Gd = m.Gdtab + (size_t)b * d.Hp * ny * nd;
                const double* dd0 = io.d0 + (size_t)b * nd;
                const double* Dh = io.Dhat0 + (size_t)b * d.nD;
                const double* Dd = m.Dd + (size_t)b * ny * nd;   // (ny,nd) col-major
                for (int e = 0; e < nd; ++e) {
                    acc += Gd[(t * ny + a) * nd + e] * dd0[e];          // G block t = Ĉ Â^t B̂d
                    acc += Dd[a + ny * e] * Dh[t * nd + e];             // J diagonal block = D̂d
                    for (int j = 0; j < t; ++j)                         // J[t, j] = G block t-j-1
                        acc += Gd[((t - j - 1) * ny + a) * nd + e] * Dh[j * nd + e];
                }
            }
            F[r] = acc;
        }
        w.sync();
        // q̃ = 2[(M Ẽ)'(F - R̂y) + (L P̃u)'(Tu lastu0 - R̂u)]   execute.jl:262-275 (deviation form)
        double* tY = sm + c.tA[P_Y];
        const double* Md = m.Mdiag + (size_t)b * nY;
        const bool rconst = d.flags & 1u;
        auto cy = [&](int r) {
            return F[r] - (rconst ? io.Ry[(size_t)b * ny + (r % ny)] : io.Ry[(size_t)b * nY + r]);
        };
        if ((!DM::is_static || MPCQP_SPEC_DENSE) && m.Mfull) {      // dense M_Hp: M (F - R̂y), column-major symmetric (coalesced over r)
            const double* Mf = m.Mfull + (size_t)b * nY * nY;
            for (int r = w.lane; r < nY; r += WAVE) {
                double acc = 0.0;
                for (int r2 = 0; r2 < nY; ++r2) acc += Mf[r + (size_t)nY * r2] * cy(r2);
                tY[r] = acc;
            }
        } else if (m.Mblk) {          // block-diagonal M_Hp: (M Cy)[t,a] = sum_a' M_t[a,a'] Cy[t,a']
            const double* Mb = m.Mblk + (size_t)b * d.Hp * ny * ny;
            for (int r = w.lane; r < nY; r += WAVE) {
                const int t = r / ny, a = r - t * ny;
                double acc = 0.0;
                for (int a2 = 0; a2 < ny; ++a2) acc += Mb[((size_t)t * ny + a2) * ny + a] * cy(t * ny + a2);
                tY[r] = acc;
            }
        } else {
            for (int r = w.lane; r < nY; r += WAVE) tY[r] = Md[r] * cy(r);
        }
        for (int k = w.lane; k < d.nZ; k += WAVE) q[k] = 0.0;
        w.sync();
        qp.Et_apply_add(tY, q, 2.0);
        const double* Ld = m.Ldiag + (size_t)b * d.nU;
        for (int k = w.lane; k < d.nDU; k += WAVE) {
            const int j = k / nu, cc = k - j * nu;
            double acc = 0.0;
            if ((!DM::is_static || MPCQP_SPEC_DENSE) && m.Ldense) {    // dense L_Hp: (Pu' L (Tu lastu0 - R̂u))[k]
                const double* Lf = m.Ldense + (size_t)b * d.nU * d.nU;
                for (int t = qp.jl(j); t < d.Hp; ++t)
                    for (int r2 = 0; r2 < d.nU; ++r2) {
                        const double ru = io.Ru ? io.Ru[(size_t)b * d.nU + r2] : 0.0;
                        acc += Lf[(t * nu + cc) + (size_t)d.nU * r2] * (lu[r2 % nu] - ru);
                    }
            } else {
                // every step of the horizon is loaded (eight at a time), the steps before the block's first count zero: the trip
                // count is the same for every lane and the loads do not wait for each other
                const int t0 = qp.jl(j);
                const double* Lp = Ld + cc;
                const double* Rp = io.Ru ? io.Ru + (size_t)b * d.nU + cc : nullptr;
                const double luc = lu[cc];
                for (int tb = 0; tb < d.Hp; tb += 8) {
                    double lv[8], rv[8];
                    MPCQP_UNROLL
                    for (int u = 0; u < 8; ++u) {
                        const int t = tb + u < d.Hp ? tb + u : d.Hp - 1;
                        lv[u] = Lp[t * nu];
                        rv[u] = Rp ? Rp[t * nu] : 0.0;
                    }
                    MPCQP_UNROLL
                    for (int u = 0; u < 8; ++u) {
                        const int t = tb + u;
                        acc += (t >= t0 && t < d.Hp) ? lv[u] * (luc - rv[u]) : 0.0;
                    }
                }
            }
            q[k] += 2.0 * acc;       // same lane wrote q[k] in Et_apply_add
        }
        // variable bounds (init_boxconstraint_mpc, construct.jl:1209-1234)
        for (int k = w.lane; k < d.nZ; k += WAVE) {
            double lo = -INFINITY, hi = INFINITY;
            if (k < d.nDU) {
                if (m.DUmin && (!d.neps || !m.C_dumin || m.C_dumin[(size_t)b * d.nDU + k] == 0.0))
                    lo = m.DUmin[(size_t)b * d.nDU + k];
                if (m.DUmax && (!d.neps || !m.C_dumax || m.C_dumax[(size_t)b * d.nDU + k] == 0.0))
                    hi = m.DUmax[(size_t)b * d.nDU + k];
            } else {
                lo = 0.0;            // ϵ >= 0
            }
            zlo[k] = lo; zhi[k] = hi;
        }
        w.sync();
        // terminal free response fx̂ = bx̂ + kx̂ x̂0 + vx̂ lastu0 (+ gx̂ d0 + jx̂ D̂0)  transcription.jl:815-821
        double* fx = sm + c.tB[P_X];       // parked here until the rows are initialised
        if (qp.pair_on(P_X)) {
            const double* kx = m.kxT + (size_t)b * nx * nx;
            const double* ex0 = sm + c.exT;          // block j=0 is vx̂ = S(Hp-1) B̂u (j_0 = 0)
            for (int i = w.lane; i < nx; i += WAVE) {
                double acc = m.bxv[(size_t)b * nx + i];
                for (int k = 0; k < nx; ++k) acc += kx[i + nx * k] * x0[k];
                for (int cc = 0; cc < nu; ++cc) acc += ex0[i * nu + cc] * lu[cc];
                if (nd > 0) {
                    const double* Xd = m.Xdtab + (size_t)b * d.Hp * nx * nd;
                    const double* dd0 = io.d0 + (size_t)b * nd;
                    const double* Dh = io.Dhat0 + (size_t)b * d.nD;
                    for (int e = 0; e < nd; ++e) {
                        acc += Xd[((d.Hp - 1) * nx + i) * nd + e] * dd0[e];       // gx̂ = Â^{Hp-1} B̂d
                        for (int j = 1; j < d.Hp; ++j)                            // jx̂ block j
                            acc += Xd[((d.Hp - j - 1) * nx + i) * nd + e] * Dh[(j - 1) * nd + e];
                    }
                }
                fx[i] = acc;
            }
            w.sync();
        }
        // b vector, finite rows only (linconstraint!, transcription.jl:824-842 and i_b :692-700)
        int cntl = 0;
        double hmax = 0.0, wacc = 0.0;
        for_rows([&](int g, int k, Row& r) {
            double bound = INFINITY;
            const size_t o = (size_t)b * d.len(g >> 1) + k;
            const bool hosted = (g >> 1) == P_Y && k >= d.nY;      // the row -eps <= 0 riding in a Ŷ group: bound 0 in its host, absent in the other
            switch (g) {
                case 0: bound = (k == d.nZ - 1 && d.neps && d.eps_host() >= 0) ? INFINITY : -zlo[k]; break;
                case 1: bound = zhi[k]; break;
                case 2 * P_U:
                case 2 * P_U + 1: {
                    // every step of a move-blocking interval has the same row of Pu
                    // (construct.jl:797-806): only its tightest bound can be active, so the
                    // interval contributes ONE row (same feasible set as the reference's nb_j rows)
                    const double* bd = (g & 1) ? m.U0max : m.U0min;
                    if (bd) {
                        const int j = k / nu, cc = k - j * nu;
                        const int t0 = qp.jl(j), t1 = (j + 1 < d.Hc) ? qp.jl(j + 1) : d.Hp;
                        const double* bp = bd + (size_t)b * d.nU + cc;
                        double v = (g & 1) ? INFINITY : -INFINITY;
                        // eight steps at a time, the index clamped to the interval's last step (min / max do not mind a repeat)
                        for (int tb = t0; tb < t1; tb += 8) {
                            double xv[8];
                            MPCQP_UNROLL
                            for (int u = 0; u < 8; ++u) { const int t = tb + u < t1 ? tb + u : t1 - 1; xv[u] = bp[t * nu]; }
                            MPCQP_UNROLL
                            for (int u = 0; u < 8; ++u) v = (g & 1) ? fmin(v, xv[u]) : fmax(v, xv[u]);
                        }
                        bound = (g & 1) ? v - lu[cc] : -v + lu[cc];
                    }
                    break;
                }
                case 2 * P_DU:
                    if (m.DUmin && m.C_dumin && m.C_dumin[o] != 0.0) bound = -m.DUmin[o];
                    break;
                case 2 * P_DU + 1:
                    if (m.DUmax && m.C_dumax && m.C_dumax[o] != 0.0) bound = m.DUmax[o];
                    break;
                case 2 * P_Y: if (hosted) bound = (g == d.eps_host()) ? 0.0 : INFINITY; else if (m.Y0min) bound = -m.Y0min[o] + F[k]; break;
                case 2 * P_Y + 1: if (hosted) bound = (g == d.eps_host()) ? 0.0 : INFINITY; else if (m.Y0max) bound = m.Y0max[o] - F[k]; break;
                case 2 * P_X: if (m.x0min) bound = -m.x0min[o] + fx[k]; break;
                case 2 * P_X + 1: if (m.x0max) bound = m.x0max[o] - fx[k]; break;
                case 2 * P_W: if (m.Wmin) bound = -m.Wmin[o] + Fw_at(k, io, x0, lu); break;
                case 2 * P_W + 1: if (m.Wmax) bound = m.Wmax[o] - Fw_at(k, io, x0, lu); break;
            }
            const bool ok = fabs(bound) < BIG && bound == bound;
            r.h = ok ? bound : 2.0 * BIG;
            r.s = 1.0;
            r.lam = ok ? 1.0 : 0.0;
            r.rp = 0.0; r.gd = 0.0; r.pp = 0.0;
            rows.set_cs(g, k / WAVE, soft_init(g, k));
            if (ok) { ++cntl; wacc += rowweight(g >> 1, k); hmax = fmax(hmax, fabs(bound)); }
        });
        mact = w.isum(cntl);
        wsum = w.sum(wacc);
        nh = 1.0 + w.maxv(hmax);
        w.sync();
    }

    // Running sums over the block columns of src[(j, c)] for nDU > 64 (several variables per lane): lane c < nu runs
    // down (prefix) or up (suffix) its channel with the sum in a register -- Hc dependent additions on independent LDS
    // loads, instead of a loop of up to Hc terms in EVERY lane (nZ~ = 106: 10k and 7k cycles per iteration in G'w and
    // G v).  dst may be src.  The caller fences before (src complete) -- the fence after is here.
    MPCQP_HD void block_scan(const double* src, double* dst, bool suffix) {
        const int nu = d.nu, Hc = d.Hc;
        if constexpr (DM::is_static) {
            // compile-time Hc: every load first (dst may alias src, which would otherwise order each load behind the
            // store before it: 160 cycles per step measured), then the additions, then the stores
            if (w.lane < nu) {
                double x[DM::Hc];
                MPCQP_UNROLL
                for (int i = 0; i < DM::Hc; ++i) x[i] = src[(suffix ? DM::Hc - 1 - i : i) * DM::nu + w.lane];
                MPCQP_UNROLL
                for (int i = 1; i < DM::Hc; ++i) x[i] += x[i - 1];
                MPCQP_UNROLL
                for (int i = 0; i < DM::Hc; ++i) dst[(suffix ? DM::Hc - 1 - i : i) * DM::nu + w.lane] = x[i];
            }
            w.sync();
            return;
        }
        if (w.lane < nu) {
            double acc = 0.0;
            MPCQP_UNROLL4
            for (int i = 0; i < Hc; ++i) {
                const int k = (suffix ? Hc - 1 - i : i) * nu + w.lane;
                acc += src[k];
                dst[k] = acc;
            }
        }
        w.sync();
    }

    // ---- primitives of G v: ucum (held cumulative sum), tY = E v, tX = ex̂ v ------------------
    MPCQP_HD void primitives(const double* v) {
        const int nu = d.nu;
        MPCQP_TICK(tic11_);
        if (qp.pair_on(P_U) || qp.pair_on(P_W)) {
            double* ucum = sm + c.ucum;
            if (d.nDU <= WAVE) {
                const double acc = qp.block_prefix(w.lane < d.nDU ? v[w.lane] : 0.0);
                if (w.lane < d.nDU) ucum[w.lane] = acc;
            } else {
                block_scan(v, ucum, false);
            }
        }
        MPCQP_TOCK(11, tic11_);
        MPCQP_TICK(tic12_);
        if (!(MPCQP_ABLATE & 8))
        if (qp.pair_on(P_Y) || qp.pair_on(P_W)) qp.E_apply(v, sm + c.tA[P_Y]);
        MPCQP_TOCK(12, tic12_);
        if (qp.pair_on(P_X)) {
            double* tX = sm + c.tA[P_X];
            for (int i = w.lane; i < d.nxh; i += WAVE) {
                double acc = 0.0;
                for (int k = 0; k < d.nDU; ++k) acc += qp.Xat(i, k) * v[k];
                tX[i] = acc;
            }
        }
        w.sync();
    }

    MPCQP_HD double prim(int p, int k, const double* v) const {
        switch (p) {
            case P_BOX: return v[k];
            case P_U: return sm[c.ucum + k];
            case P_DU: return v[k];
            case P_Y: return (d.eps_host() >= 0 && k >= d.nY) ? 0.0 : sm[c.tA[P_Y] + k];      // (hosted row -eps <= 0: its row of E is zero)
            case P_W: {
                if constexpr (!has_w<DM>()) return 0.0;
                else {
                    const int t = k / d.nw, i = k - t * d.nw;
                    double acc = 0.0;
                    if (t >= 1)
                        for (int a = 0; a < d.ny; ++a) acc += qp.Wy_(i, a) * sm[c.tA[P_Y] + (t - 1) * d.ny + a];
                    const int jb = qp.blkW(t);
                    for (int cc = 0; cc < d.nu; ++cc) acc += qp.Wu_(i, cc) * sm[c.ucum + jb * d.nu + cc];
                    return acc;
                }
            }
            default: return sm[c.tA[P_X] + k];
        }
    }

    // ---- fn(Row&, (G v)[row]) for every finite row -------------------------------------------
    template <class Fn>
    MPCQP_HD void apply_G(const double* v, Fn fn) {
        MPCQP_RELANE(5);
        MPCQP_TIC();
        primitives(v);
        const double e = d.neps ? v[d.nZ - 1] : 0.0;
        for_rows([&](int g, int k, Row& r) {
            if (!fin(r)) return;
            const double pv = prim(g >> 1, k, v);
            fn(r, ((g & 1) ? pv : -pv) - r.cs * e);
        });
        w.sync();    // tA[P_Y]/ucum are reused by the next product
        MPCQP_TOC(0);
    }

    // ---- gt = G' wv, wv(Row&) evaluated on finite rows ---------------------------------------
    template <class Fn>
    MPCQP_HD void apply_Gt(Fn wv) {
        MPCQP_RELANE(6);
        MPCQP_TIC();
        MPCQP_TICK(tic_gt_);
        const int nu = d.nu;
        // per pair: tA[k] = w_max - w_min ; eps accumulates -(c_min w_min + c_max w_max)
        double eacc = 0.0;
        for_pairs([&](int p, int k, Row* r0, Row* r1) {
            double wmin = 0.0, wmax = 0.0;
            if (r0 && fin(*r0)) { wmin = wv(*r0); eacc -= r0->cs * wmin; }
            if (r1 && fin(*r1)) { wmax = wv(*r1); eacc -= r1->cs * wmax; }
            sm[c.tA[p] + k] = wmax - wmin;
        });
        eacc = w.sum(eacc);
        w.sync();
        bool useY = qp.pair_on(P_Y), useU = qp.pair_on(P_U);
        if constexpr (has_w<DM>()) {
            if (qp.pair_on(P_W)) {          // custom rows reach z through the Y and U primitives
                qp.W_fold(sm + c.tA[P_W], sm + c.tA[P_Y], !useY, sm + c.tA[P_U], !useU);
                useY = useU = true;
            }
        }
        MPCQP_TOCK(8, tic_gt_);
        MPCQP_TICK(tic9_);
        double sufU = 0.0;
        if (useU && d.nDU <= WAVE) sufU = qp.block_suffix(w.lane < d.nDU ? sm[c.tA[P_U] + w.lane] : 0.0);
        else if (useU) block_scan(sm + c.tA[P_U], sm + c.tA[P_U], true);       // (tA[P_U] is consumed here: in place)
        for (int k = w.lane; k < d.nZ; k += WAVE) {
            double acc = 0.0;
            if (qp.pair_on(P_BOX)) acc += sm[c.tA[P_BOX] + k];
            if (k < d.nDU) {
                if (qp.pair_on(P_DU)) acc += sm[c.tA[P_DU] + k];
                if (useU && d.nDU <= WAVE) acc += sufU;
                else if (useU) acc += sm[c.tA[P_U] + k];
                if (qp.pair_on(P_X)) {
                    const double* tX = sm + c.tA[P_X];
                    for (int i = 0; i < d.nxh; ++i) acc += qp.Xat(i, k) * tX[i];
                }
            } else {
                acc += eacc;
            }
            gt[k] = acc;
        }
        MPCQP_TOCK(9, tic9_);
        MPCQP_TICK(tic10_);
        if (useY && !(MPCQP_ABLATE & 16)) qp.Et_apply_add(sm + c.tA[P_Y], gt);   // same lane owns gt[k]
        w.sync();
        MPCQP_TOCK(10, tic10_);
        MPCQP_TOC(1);
    }

    // 2 N_k (k < nDU), 2 C (slack); 2 sum over the steps of block(k) of L[t, c(k)]
    MPCQP_HD double H2N(int k) const {
        if (one_row_per_lane<DM>()) return h2n_;            // (callers pass their own lane's k)
        return k < d.nDU ? 2.0 * m.Ndiag[(size_t)b * d.nDU + k] : (d.neps ? 2.0 * m.Cwt[b] : 0.0);
    }
    MPCQP_HD double H2N_load(int k) const {
        return k < d.nDU ? 2.0 * m.Ndiag[(size_t)b * d.nDU + k] : ((d.neps && k == d.nZ - 1) ? 2.0 * m.Cwt[b] : 0.0);
    }
    MPCQP_HD double H2L_load(int k) const {
        if (k >= d.nDU) return 0.0;
        const int j = k / d.nu, cc = k - j * d.nu;
        const int t0 = qp.jl(j), t1 = (j + 1 < d.Hc) ? qp.jl(j + 1) : d.Hp;
        const double* Lp = m.Ldiag + (size_t)b * d.nU + cc;
        double acc = 0.0;
        for (int tb = t0; tb < t1; tb += 8) {           // eight loads in flight (QP::stage)
            double lv[8];
            MPCQP_UNROLL
            for (int u = 0; u < 8; ++u) { const int t = tb + u < t1 ? tb + u : t1 - 1; lv[u] = Lp[t * d.nu]; }
            MPCQP_UNROLL
            for (int u = 0; u < 8; ++u) acc += (tb + u < t1) ? lv[u] : 0.0;
        }
        return 2.0 * acc;
    }
    MPCQP_HD double H2L(int k) const { return one_row_per_lane<DM>() ? h2l_ : H2L_load(k); }
    MPCQP_HD void init_fold_H() {
        double lmx = 0.0;
        if (one_row_per_lane<DM>()) {
            h2l_ = H2L_load(w.lane);
            h2n_ = H2N_load(w.lane);
            lmx = fabs(h2l_);
        } else {
            for (int k = w.lane; k < d.nDU; k += WAVE) lmx = fmax(lmx, fabs(H2L_load(k)));
        }
        h_Lnz = w.maxv(lmx) > 0.0;
        if (hwy_cached()) {
            const double* Md = m.Mdiag + (size_t)b * d.nY;
            MPCQP_UNROLL
            for (int q_ = 0; q_ < HWQ; ++q_) { const int r = w.lane + WAVE * q_; hwy_[q_] = 2.0 * Md[r < d.nY ? r : 0]; }
            MPCQP_UNROLL
            for (int q_ = 0; q_ < HWQ; ++q_) { const int r = w.lane + WAVE * q_; hwy_[q_] = r < d.nY ? hwy_[q_] : 0.0; }
        }
        fold_H = MPCQP_FOLD_H && !m.Mblk && !m.Mfull && !m.Ndense && !m.Ldense && qp.pair_on(P_Y) && d.nDU <= WAVE &&
                 (qp.pair_on(P_U) || !h_Lnz);
        if (fold_H) {           // pads of the packed layout hold zero from here on (load_H used to bring them)
            for (int i = w.lane; i < d.npk; i += WAVE) Phi[i] = 0.0;
            w.sync();
        }
    }

    // rd <- H̃ z by structured products (fold_H): 2 E'(M (E z)) + 2 N z + 2 Pu'(L (Pu z)), 2 C z_eps.  `Ez_ready`: tA[P_Y]
    // holds E z already (apply_G(z) just ran).  tA[P_Y] is used as scratch.
    MPCQP_HD void Hz_structured(bool Ez_ready) {
        MPCQP_TIC();
        double* tY = sm + c.tA[P_Y];
        if (!Ez_ready) { qp.E_apply(z, tY); w.sync(); }
        const double* Md = m.Mdiag + (size_t)b * d.nY;
        if (hwy_cached()) { for (int r = w.lane; r < d.nY; r += WAVE) tY[r] *= 0.5 * hwy(r); }
        else { for (int r = w.lane; r < d.nY; r += WAVE) tY[r] *= Md[r]; }
        for (int k = w.lane; k < d.nZ; k += WAVE) rd[k] = 0.0;
        w.sync();
        qp.Et_apply_add(tY, rd, 2.0);                       // (lane k owns rd[k])
        double suf = 0.0;
        if (h_Lnz && d.nDU <= WAVE) {
            const double pre = qp.block_prefix(w.lane < d.nDU ? z[w.lane] : 0.0);
            suf = qp.block_suffix(w.lane < d.nDU ? H2L(w.lane) * pre : 0.0);
        }
        for (int k = w.lane; k < d.nZ; k += WAVE) {
            const double hn = one_row_per_lane<DM>() ? h2n_ : H2N_load(k);
            rd[k] = (k < d.nDU ? rd[k] + suf : 0.0) + hn * z[k];
        }
        w.sync();
        MPCQP_TOC(2);
    }

    // ---- Phi <- H̃ (global -> LDS) ------------------------------------------------------------
    MPCQP_HD void load_H() {
        MPCQP_RELANE(11);
        if constexpr (W::NTEAM > 1) w.post(TJ_LOADH);
        load_H_share();
        if constexpr (W::NTEAM > 1) w.join();
        w.sync();
    }
    // (eight loads of a lane in flight -- Qp::stage; a plain copy loop is one memory latency per trip: 180 trips at nZ~ = 151 --
    //  and a team's wavefronts take interleaved chunks of 64)
    MPCQP_HD void load_H_share() {
        const double* H = m.Hpk + (size_t)b * d.npk;
        constexpr int NB = 8;
        const int n = d.npk, stride = WAVE * W::NTEAM;
        for (int i0 = WAVE * W::WV; i0 < n; i0 += NB * stride) {
            double v[NB];
            MPCQP_UNROLL
            for (int q_ = 0; q_ < NB; ++q_) { const int i = i0 + w.lane + stride * q_; v[q_] = H[i < n ? i : 0]; }
            MPCQP_UNROLL
            for (int q_ = 0; q_ < NB; ++q_) { const int i = i0 + w.lane + stride * q_; if (i < n) Phi[i] = v[q_]; }
        }
    }

    // true: add_GtDG() builds Phi = H̃ + G'DG itself, H̃ read from global memory by the matrix-core pass of the Ŷ rows
    // (EtDE_add_mfma with Hg); false: Phi must hold H̃ when add_GtDG() is called (load_H)
    MPCQP_HD bool phi_direct() const {
#if defined(__HIP_DEVICE_COMPILE__)
        return fold_H || (DM::is_static && qp.pair_on(P_Y));
#else
        return fold_H;
#endif
    }
    // the matrix-core pass of E'DE writes the whole stored triangle (instead of adding to it)
    MPCQP_HD bool etde_overwrites() const {
#if defined(__HIP_DEVICE_COMPILE__)
        return DM::is_static && qp.pair_on(P_Y);
#else
        return false;
#endif
    }

    // Parts of G'DG whose inputs (scratch vectors of the row pass) and outputs (Phi) live in LDS: the lanes of every wavefront of
    // a team take their share (W::NTEAM, W::WV; one wavefront: the loops of rounds 1-5).
    // U rows of problems with several variables per lane: the suffix sums (tA[P_U], scanned in place) along the rows of Pu'dU Pu
    MPCQP_HD void GtDG_urows_share() {
        const int nu = d.nu, nDU = d.nDU;
        const double* tU = sm + c.tA[P_U];
        for (int k = w.lane + WAVE * ((W::WV + W::NTEAM - 1) % W::NTEAM); k < nDU; k += WAVE * W::NTEAM) {
            const int j = k / nu, cc = k - j * nu;
            const double suf = tU[k];
            MPCQP_PRAGMA(unroll MPCQP_URMW_UNROLL)
            for (int j2 = 0; j2 <= j; ++j2) Phi[pk(k, j2 * nu + cc)] += suf;
        }
    }
    // terminal rows: ex̂' dX ex̂
    MPCQP_HD void GtDG_xrows_share() {
        const int nDU = d.nDU, ntri = nDU * (nDU + 1) / 2;
        const double* tX = sm + c.tA[P_X];
        for (int idx = w.lane + WAVE * W::WV; idx < ntri; idx += WAVE * W::NTEAM) {
            int i, ip;
            Qp<W, DM>::unpack_idx(idx, i, ip);
            double acc = 0.0;
            for (int r = 0; r < d.nxh; ++r) acc += qp.Xat(r, i) * tX[r] * qp.Xat(r, ip);
            Phi[pk(i, ip)] += acc;
        }
    }
    // custom rows: E_w' dW E_w, rows formed on the fly (set-up-grade path)
    MPCQP_HD void GtDG_wrows_share() {
        if constexpr (has_w<DM>()) {
            const int nu = d.nu, nDU = d.nDU, ntri = nDU * (nDU + 1) / 2, nw = d.nw;
            const double* dW = sm + c.tA[P_W];
            for (int idx = w.lane + WAVE * W::WV; idx < ntri; idx += WAVE * W::NTEAM) {
                int i, ip;
                Qp<W, DM>::unpack_idx(idx, i, ip);
                const int j = i / nu, cc = i - j * nu, j2 = ip / nu, c2 = ip - j2 * nu;
                double acc = 0.0;
                for (int t = 0; t <= d.Hp; ++t)
                    for (int iw = 0; iw < nw; ++iw) {
                        const double dk = dW[t * nw + iw];
                        if (dk != 0.0) acc += dk * qp.Ew_at(t, iw, j, cc) * qp.Ew_at(t, iw, j2, c2);
                    }
                Phi[pk(i, ip)] += acc;
            }
        }
    }

    // ---- Phi (+)= G' diag(dd) G, dd(Row&) evaluated on finite rows (see phi_direct) ---------------
    template <class Fn>
    MPCQP_HD void add_GtDG(Fn dd) {
        MPCQP_RELANE(3);
        MPCQP_TIC();
        const int nu = d.nu, nDU = d.nDU, nZ = d.nZ;
        double ee = 0.0;
        for_pairs([&](int p, int k, Row* r0, Row* r1) {
            double dmin = 0.0, dmax = 0.0, cmin = 0.0, cmax = 0.0;
            if (r0 && fin(*r0)) { dmin = dd(*r0); cmin = r0->cs; }
            if (r1 && fin(*r1)) { dmax = dd(*r1); cmax = r1->cs; }
            double hw = 0.0;        // H̃'s own term of this primitive (fold_H)
            if (fold_H) {
                if (p == P_Y) hw = hwy_cached() ? hwy(k) : (k < d.nY ? 2.0 * m.Mdiag[(size_t)b * d.nY + k] : 0.0);
                else if (p == P_U) hw = H2L(k);
            }
            sm[c.tA[p] + k] = dmin + dmax + hw;
            if (p != P_BOX) {
                sm[c.tB[p] + k] = cmin * dmin - cmax * dmax;
                ee += cmin * cmin * dmin + cmax * cmax * dmax;
            }
        });
        ee = w.sum(ee);
        w.sync();
        if (fold_H && !etde_overwrites()) {      // (scalar E'DE adds in place: start from zero instead of from H̃)
            for (int i = w.lane; i < d.npk; i += WAVE) Phi[i] = 0.0;
            w.sync();
        }
        bool epsY = qp.pair_on(P_Y), epsU = qp.pair_on(P_U);     // who feeds the ϵ row below
        if constexpr (has_w<DM>()) {
            if (qp.pair_on(P_W) && d.neps) {      // ϵ row of the custom rows: E_w' tB_W through Y and U
                qp.W_fold(sm + c.tB[P_W], sm + c.tB[P_Y], !epsY, sm + c.tB[P_U], !epsU);
                epsY = epsU = true;
            }
        }
        MPCQP_TOC(3);
        // dense E' dY E (+ the ϵ row of the Ŷ rows on the matrix-core path)
        int eps_t0 = -1;          // first step whose Ŷ rows the matrix-core path put into the ϵ row
        if (qp.pair_on(P_Y)) {
            MPCQP_TIC();
            if (!(MPCQP_ABLATE & 1))
            eps_t0 = qp.EtDE_add(sm + c.tA[P_Y], Phi, 1.0, d.neps ? sm + c.tB[P_Y] : nullptr,
                                 (phi_direct() && !fold_H) ? m.Hpk + (size_t)b * d.npk : nullptr, fold_H && etde_overwrites());
            w.sync();      // the MFMA write-back uses its own entry->lane map
            MPCQP_TOC(4);
        }
        MPCQP_TICK(tic5_);
        // U rows: Pu' dU Pu has entry ((j,c),(j',c)) = sum_{jj >= max(j,j')} dU[jj,c]: lane (j,c)
        // forms its suffix sum once and adds it along its own row of the lower triangle
        if (qp.pair_on(P_U) && nDU <= WAVE && !(MPCQP_ABLATE & 64)) {
            // one row per lane: suffix sum by log steps, then unconditional read-modify-writes along
            // the lane's row (block columns right of the diagonal go to the trash slot)
            const double* tU = sm + c.tA[P_U];
            const int k = w.lane < nDU ? w.lane : 0;
            const int j = k / nu, cc = k - j * nu;
            const double suf = qp.block_suffix(w.lane < nDU ? tU[k] : 0.0);
            double* const trash = sm + c.zero + 4;
            double* const row = Phi + pk(k, cc);
            if constexpr (DM::is_static) {
#if MPCQP_CHOL_RMW_BATCH
                // all reads, then all writes (entry by entry the trash slot's possible aliasing serialises Hc LDS round trips)
                double* q_[DM::Hc];
                double old_[DM::Hc];
                MPCQP_UNROLL
                for (int j2 = 0; j2 < DM::Hc; ++j2) {
                    q_[j2] = (j2 <= j && w.lane < nDU) ? row + j2 * nu : trash;
                    old_[j2] = *q_[j2];
                }
                MPCQP_UNROLL
                for (int j2 = 0; j2 < DM::Hc; ++j2) *q_[j2] = old_[j2] + suf;
#else
                MPCQP_UNROLL
                for (int j2 = 0; j2 < d.Hc; ++j2) {
                    double* const q_ = (j2 <= j && w.lane < nDU) ? row + j2 * nu : trash;
                    *q_ += suf;
                }
#endif
            } else {
                for (int j2 = 0; j2 < d.Hc; ++j2) {
                    double* const q_ = (j2 <= j && w.lane < nDU) ? row + j2 * nu : trash;
                    *q_ += suf;
                }
            }
        } else if (qp.pair_on(P_U)) {
            double* tU = sm + c.tA[P_U];
            w.sync();
            block_scan(tU, tU, true);              // (nothing reads the per-block sums after this point: in place)
            if constexpr (W::NTEAM > 1) w.post(TJ_UROWS);
            GtDG_urows_share();
            if constexpr (W::NTEAM > 1) w.join();
        }
        if (qp.pair_on(P_X)) {
            w.sync();
            if constexpr (W::NTEAM > 1) w.post(TJ_XROWS);
            GtDG_xrows_share();
            if constexpr (W::NTEAM > 1) w.join();
        }
        if constexpr (has_w<DM>()) {
            if (qp.pair_on(P_W)) {        // E_w' dW E_w, rows formed on the fly (set-up-grade path)
                w.sync();
                if constexpr (W::NTEAM > 1) w.post(TJ_WROWS);
                GtDG_wrows_share();
                if constexpr (W::NTEAM > 1) w.join();
            }
        }
        w.sync();
        for (int k = w.lane; k < nZ; k += WAVE) {
            double acc = 0.0;
            if (qp.pair_on(P_BOX)) acc += sm[c.tA[P_BOX] + k];
            if (k < nDU && qp.pair_on(P_DU)) acc += sm[c.tA[P_DU] + k];
            if (k == nZ - 1 && d.neps) acc += ee;
            if (fold_H) acc += one_row_per_lane<DM>() ? h2n_ : H2N_load(k);
            Phi[pk(k, k)] += acc;
        }
        // ϵ row: Phi[eps, k] += sum_pairs L_P' tB     (staged in dz, which is free here)
        if (d.neps) {
            double* st = dz;
            double sufB = 0.0;
            if (epsU && nDU <= WAVE) sufB = qp.block_suffix(w.lane < nDU ? sm[c.tB[P_U] + w.lane] : 0.0);
            for (int k = w.lane; k < nDU; k += WAVE) {
                double acc = 0.0;
                if (qp.pair_on(P_DU)) acc += sm[c.tB[P_DU] + k];
                if (epsU && nDU <= WAVE) acc += sufB;
                else if (epsU) {
                    const int j = k / nu, cc = k - j * nu;
                    const double* tU = sm + c.tB[P_U];
                    MPCQP_UNROLL4
                    for (int jj = j; jj < d.Hc; ++jj) acc += tU[jj * nu + cc];
                }
                if (qp.pair_on(P_X)) {
                    const double* tX = sm + c.tB[P_X];
                    for (int i = 0; i < d.nxh; ++i) acc += qp.Xat(i, k) * tX[i];
                }
                st[k] = acc;
            }
            if (epsY && eps_t0 != 0) qp.Et_apply_add(sm + c.tB[P_Y], st, 1.0, eps_t0);   // same lane owns st[k]
            for (int k = w.lane; k < nDU; k += WAVE) Phi[pk(nZ - 1, k)] += st[k];
        }
        w.sync();
        MPCQP_TOCK(5, tic5_);
    }

    MPCQP_HD static long long clock64_() {
#if defined(MPCQP_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        return clock64();
#else
        return 0;
#endif
    }

    // four consecutive doubles from a 32-byte aligned LDS address (two ds_read_b128 / ds_write_b128)
    MPCQP_HD static void load4(const double* p, double* x) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef double v2d_ __attribute__((ext_vector_type(2)));
        const v2d_ a = reinterpret_cast<const v2d_*>(p)[0], b2 = reinterpret_cast<const v2d_*>(p)[1];
        x[0] = a.x; x[1] = a.y; x[2] = b2.x; x[3] = b2.y;
#else
        x[0] = p[0]; x[1] = p[1]; x[2] = p[2]; x[3] = p[3];
#endif
    }
    MPCQP_HD static void store4(double* p, const double* x) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef double v2d_ __attribute__((ext_vector_type(2)));
        reinterpret_cast<v2d_*>(p)[0] = v2d_{x[0], x[1]};
        reinterpret_cast<v2d_*>(p)[1] = v2d_{x[2], x[3]};
#else
        p[0] = x[0]; p[1] = x[1]; p[2] = x[2]; p[3] = x[3];
#endif
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // Blocked (right-looking at 16-column granularity) part of the factorisation on the matrix
    // cores: before panel P (columns 16P..16P+15) is factored, the contribution of all finished
    // columns j < 16P to the panel's columns is removed from Phi,
    //   Phi[i][16P + c] -= sum_{j < 16P} L[i][j] L[16P + c][j]          (rows i >= 16P),
    // one v_mfma_f64_16x16x4 per (row tile, 4 columns of j).  Operand of row tile T at K step kk is
    // X_T = L[16T + (lane & 15)][4 kk + (lane >> 4)], used as A for the tile's rows and as B (T = P)
    // for the panel's columns.  Rows >= nZ are clamped to the last row: they only produce tile
    // rows / columns that are not written back.
    template <int P>
    __device__ __forceinline__ void chol_panel_update() {
        // (a team of wavefronts splits the row tiles below the panel: W::NTEAM, mpcqp_devwave.h)
        if constexpr (W::NTEAM > 1) w.post(TJ_PANEL, P);
        chol_panel_share<P>();
        if constexpr (W::NTEAM > 1) w.join();
    }
    template <int P>
    __device__ __forceinline__ void chol_panel_share() {
        typedef double v4d_ __attribute__((ext_vector_type(4)));
        constexpr int n = DM::nZ, NT = (n + 15) / 16;
        if constexpr (P < NT) {
            const int li = w.lane & 15, lk = w.lane >> 4;
            auto mine = [](int I) { return team_mine_mfma<W>(I - P); };       // row tile I of this wavefront (compile-time after unrolling)
            const double* X[NT];
            v4d_ acc[NT];
            MPCQP_UNROLL
            for (int I = P; I < NT; ++I) {
                const int row = 16 * I + li < n ? 16 * I + li : n - 1;
                X[I] = Phi + pk(row, 0) + lk;
                acc[I] = v4d_{0.0, 0.0, 0.0, 0.0};
            }
            if constexpr (team_any_mfma<W>(NT - P)) {
            MPCQP_PRAGMA(unroll MPCQP_PANEL_UNROLL)
            for (int kk = 0; kk < 4 * P; ++kk) {
                const double bb = X[P][4 * kk];
                // (one row per lane, M D M' form: one operand carries d of its column, the d vector sits in gt; the
                //  several-rows-per-lane factorisation that shares this function is LL')
                const double bs = (MPCQP_CHOL_LDL && !MPCQP_CHOL_REDUNDANT && one_row_per_lane<DM>()) ? bb * gt[4 * kk + lk] : bb;
                if (mine(P)) acc[P] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb, bs, acc[P], 0, 0, 0);
                MPCQP_UNROLL
                for (int I = P + 1; I < NT; ++I)
                    if (mine(I)) acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[I][4 * kk], bs, acc[I], 0, 0, 0);
            }
            }
            double* const trash = sm + c.zero + 4;        // unconditional write-back, see EtDE_add_mfma
            MPCQP_UNROLL
            for (int I = P; I < NT; ++I) {
                if (!mine(I)) continue;
                double* pp_[4];
                double old_[4];
                MPCQP_UNROLL
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * I + lk + 4 * reg, col = 16 * P + li;
                    pp_[reg] = (row < n && col <= row) ? Phi + pk(row, col) : trash;
                    old_[reg] = *pp_[reg];
                }
                MPCQP_UNROLL
                for (int reg = 0; reg < 4; ++reg) *pp_[reg] = old_[reg] - acc[I][reg];
            }
            w.sync();
        }
    }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
    // helper wavefronts of a team: the share of the panel update the mailbox names
    template <int P>
    __device__ __forceinline__ void panel_dispatch(int p) {
        if constexpr (P < (DM::nZ + 15) / 16) {
            if (p == P) chol_panel_share<P>();
            else panel_dispatch<P + 1>(p);
        }
    }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
    // In-panel update on the matrix cores: after the four columns K0..K0+3 of panel P (block Bk of
    // the panel) are final, their contribution to the later columns of the panel is removed,
    //   Phi[i][c] -= sum_{k < 4} L[i][K0 + k] L[c][K0 + k]        (c = 16P + 4r + ., r > Bk; i >= c),
    // one v_mfma_f64_16x16x4 per row tile with the TRANSPOSED product D' = X_P X_T' (X_T as in
    // chol_panel_update with K step K0/4): lane (li, lk) then holds, in register r, the update of
    // the entry (row 16T + li, column 16P + 4r + lk) -- four consecutive columns per quarter-wave, so
    // one read-modify-write per (tile, later block) with all lanes busy.  Replaces the left-looking
    // sweep over the panel's finished columns (up to 12 columns x 4 FMA chains per lane and block).
    template <int P, int Bk>
    __device__ __forceinline__ void chol_block_update() {
        typedef double v4d_ __attribute__((ext_vector_type(4)));
        constexpr int n = DM::nZ, NT = (n + 15) / 16, K0 = 16 * P + 4 * Bk;
        const int li = w.lane & 15, lk = w.lane >> 4;
        double x[NT];
        MPCQP_UNROLL
        for (int T = P; T < NT; ++T) {
            const int row = 16 * T + li < n ? 16 * T + li : n - 1;
            x[T] = Phi[pk(row, 0) + K0 + lk];
        }
        const double xPs = (MPCQP_CHOL_LDL && !MPCQP_CHOL_REDUNDANT && one_row_per_lane<DM>()) ? x[P] * gt[K0 + lk] : x[P];   // (M D M': d of the block's columns, written by chol_static)
        double* const trash = sm + c.zero + 4;
#if MPCQP_CHOL_RMW_BATCH
        // Every entry to be updated is requested BEFORE the matrix-core instructions are issued and written after them:
        // written as `*q -= acc[r]` entry by entry, the possible aliasing of the trash slot made each read-modify-write
        // wait for the one before it (ds_read, s_waitcnt lgkmcnt(0), v_add, ds_write: 49 dependent LDS round trips per
        // factorisation at C3, ~100 cycles each, on the critical path of the pivot chain).
        double* q_[NT][4];
        double old_[NT][4];
        MPCQP_UNROLL
        for (int T = P; T < NT; ++T) {
            MPCQP_UNROLL
            for (int r = Bk + 1; r < 4; ++r) {
                if (16 * P + 4 * r >= n) continue;
                const int row = 16 * T + li, col = 16 * P + 4 * r + lk;
                q_[T][r] = (row < n && col <= row) ? Phi + pk(row, col) : trash;
                old_[T][r] = *q_[T][r];
            }
        }
        MPCQP_UNROLL
        for (int T = P; T < NT; ++T) {
            const v4d_ acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xPs, x[T], v4d_{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            MPCQP_UNROLL
            for (int r = Bk + 1; r < 4; ++r) {
                if (16 * P + 4 * r >= n) continue;
                *q_[T][r] = old_[T][r] - acc[r];
            }
        }
#else
        MPCQP_UNROLL
        for (int T = P; T < NT; ++T) {
            const v4d_ acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xPs, x[T], v4d_{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            MPCQP_UNROLL
            for (int r = Bk + 1; r < 4; ++r) {
                if (16 * P + 4 * r >= n) continue;
                const int row = 16 * T + li, col = 16 * P + 4 * r + lk;
                double* const q_ = (row < n && col <= row) ? Phi + pk(row, col) : trash;
                *q_ -= acc[r];
            }
        }
#endif
        w.sync();
    }

    // One block of four columns of the compile-time-dims factorisation (see cholesky()); the recursion
    // over K0 unrolls the whole factorisation: every row offset, lane index and panel step is a constant.
    // MPCQP_CHOL_REDUNDANT: the 4 x 4 diagonal block is read by EVERY lane (wave-uniform LDS addresses: broadcast reads,
    // issued together with the lane's own four entries) and factored redundantly in registers; a lane then finishes its
    // own row with the block's uniform entries as plain operands.  Same operations in the same order as "every lane
    // takes the reciprocal square root of its own entry, lane k's is broadcast" (bit-identical factor), but no
    // v_readlane (and no hazard nops) on the column-to-column dependency: the chain is four rsq + a few FMAs per block.
    // The pivot thresholds of the block's rows come from an LDS vector (dz, free during a factorisation).
    template <int K0>
    __device__ __forceinline__ void chol_static(double thr) {
        constexpr int n = DM::nZ, CB = 4, P = K0 / 16, Bk = (K0 % 16) / 4;
        const int i = w.lane;
        const bool act = i < n;
        const int rowi = pk(act ? i : 0, 0);
        if constexpr (Bk == 0 && P > 0) chol_panel_update<P>();      // contribution of the finished panels
        const bool mine = act && i >= K0;
        double v[CB], lk[CB];
        load4(mine ? Phi + rowi + K0 : sm + c.zero, v);
#if MPCQP_CHOL_REDUNDANT
        constexpr int NC = (n - K0 < CB) ? n - K0 : CB;              // columns of this block that exist
        double A[NC][CB], th[CB], d_[NC], L[NC][NC];
        MPCQP_UNROLL
        for (int r = 0; r < NC; ++r) load4(Phi + pk(K0, 0) + r * (K0 + CB) + K0, A[r]);   // rows K0..K0+3 share a stride
        load4(dz + K0, th);                                          // thresholds of rows K0..K0+3 (written by cholesky())
        MPCQP_UNROLL
        for (int cc = 0; cc < NC; ++cc) {
            double t = A[cc][cc];
            MPCQP_UNROLL
            for (int k = 0; k < cc; ++k) t = fma(-L[cc][k], L[cc][k], t);
            d_[cc] = (t > th[cc]) ? rsqrt_(t) : 0.0;                 // 0: pivot below its threshold (zero column)
            MPCQP_UNROLL
            for (int r = cc + 1; r < NC; ++r) {
                double x = A[r][cc];
                MPCQP_UNROLL
                for (int k = 0; k < cc; ++k) x = fma(-L[r][k], L[cc][k], x);
                L[r][cc] = x * d_[cc];
            }
        }
        MPCQP_UNROLL
        for (int cc = 0; cc < CB; ++cc) {
            if (cc < NC) {
                double x = v[cc];
                MPCQP_UNROLL
                for (int k = 0; k < cc; ++k) x = fma(-lk[k], L[cc][k], x);
                lk[cc] = x * d_[cc];
            } else {
                lk[cc] = 0.0;                                        // a column k >= n does not exist
            }
        }
#if MPCQP_CHOL_INVD == 0
        MPCQP_UNROLL
        for (int cc = 0; cc < NC; ++cc) myinvd = (i == K0 + cc) ? d_[cc] : myinvd;
#elif MPCQP_CHOL_INVD == 1
        if (i == 0) {
            MPCQP_UNROLL
            for (int cc = 0; cc < NC; ++cc) gt[K0 + cc] = d_[cc];
        }
#else
        MPCQP_UNROLL
        for (int cc = 0; cc < NC; ++cc) gt[K0 + cc] = d_[cc];        // every lane, same value, same address
#endif
#elif MPCQP_CHOL_LDL
        // Phi = M D M' with unit lower-triangular M, stored NEGATED (m_ik = -a_ik / d_k: the substitution chains of
        // solve_static add), no square root: the pivot chain is v_rcp_f64 + two multiply-adds per column (LL': floor,
        // v_rsq_f64 + three), and -- the point -- the sweeps of solve_static use the stored entries as they are: no scaling of
        // the factor by 1/L_ii in every solve (176 multiplications per interior-point iteration).  A column's 1/d_k is known
        // when its entries are stored, which is what a row scaling of L never is.  No guard on the chain: a pivot <= thr
        // (or NaN) is told from d afterwards and the factor discarded (Step::run).
        constexpr int NCL = (n - K0 < CB) ? n - K0 : CB;             // columns of this block that exist
        MPCQP_UNROLL
        for (int cc = 0; cc < CB; ++cc) {
            if (cc < NCL) {
                const int k = K0 + cc;
                const double nidb = -w.bcast(rcp(v[cc]), k);          // -1/d_k (every lane takes the reciprocal of its own entry, lane k's is used)
                lk[cc] = v[cc] * nidb;                               // m_ik
                MPCQP_UNROLL
                for (int c2 = cc + 1; c2 < NCL; ++c2) v[c2] = fma(v[cc], w.bcast(lk[cc], K0 + c2), v[c2]);   // a_i,c2 -= a_ik a_c2,k / d_k
            } else {
                lk[cc] = 0.0;
            }
        }
#elif MPCQP_CHOL_DIAG
        // The pivot guard is a floor (v_max with the lane's threshold: one instruction on the pivot chain instead of a
        // compare and two selects after it) and L_kk = v_k / sqrt(v_k) stays in the diagonal slot: cholesky() turns it
        // into 1/L_kk -- and tells a floored pivot by L_kk^2 <= thr -- once, after the last column (one LDS read, one
        // reciprocal, one store per lane instead of a three-instruction select per column).  A factor with a floored
        // pivot is finite garbage: Step::run discards it (larger dual regularisation, no step taken).
        MPCQP_UNROLL
        for (int cc = 0; cc < CB; ++cc) {
            const int k = K0 + cc;                       // k <= 63; a column k >= n only sees zeros
            const double idl = rsqrt_(fmx(v[cc], thr));  // (only lane k's value is used)
            const double idb = w.bcast(idl, k);
            lk[cc] = v[cc] * idb;
            MPCQP_UNROLL
            for (int c2 = cc + 1; c2 < CB; ++c2) v[c2] -= lk[cc] * w.bcast(lk[cc], K0 + c2);
        }
#else
        MPCQP_UNROLL
        for (int cc = 0; cc < CB; ++cc) {
            const int k = K0 + cc;                       // k <= 63; a column k >= n only sees zeros
            const double idl = (v[cc] > thr) ? rsqrt_(v[cc]) : 0.0;
            const double idb = w.bcast(idl, k);
            lk[cc] = v[cc] * idb;
            myinvd = (i == k) ? idl : myinvd;
            MPCQP_UNROLL
            for (int c2 = cc + 1; c2 < CB; ++c2) v[c2] -= lk[cc] * w.bcast(lk[cc], K0 + c2);
        }
#endif
        if (mine) {
#if MPCQP_CHOL_LDL && !MPCQP_CHOL_REDUNDANT
            // lane K0 + a still holds its pivot in v[a] (a column's entry is not touched after its own step); the lanes
            // beyond the block pick up a meaningless value here and the right one in their own block, the last one
            // they take part in.  gt doubles as the d vector the matrix-core updates scale one operand with.
            const double t0 = (i & 1) ? v[1] : v[0], t1 = (i & 1) ? v[3] : v[2];
            myd_ = (i & 2) ? t1 : t0;
            gt[i] = myd_;
            MPCQP_UNROLL
            for (int cc = 0; cc < CB; ++cc) lk[cc] = (i > K0 + cc) ? lk[cc] : 0.0;
#else
            MPCQP_UNROLL
            for (int cc = 0; cc < CB; ++cc) lk[cc] = (MPCQP_CHOL_DIAG && !MPCQP_CHOL_REDUNDANT ? i >= K0 + cc : i > K0 + cc) ? lk[cc] : 0.0;
#endif
            store4(Phi + rowi + K0, lk);
        }
        w.sync();
        if constexpr (K0 + CB < n) {
            if constexpr (Bk < 3) chol_block_update<P, Bk>();
            chol_static<K0 + CB>(thr);
        }
    }
#endif

    // ---- in-place Cholesky of packed Phi (row-major lower, see pk()), one row per lane, nZ <= 64
    // Left-looking, four columns at a time.  The part of the four dot products that only needs
    // finished columns (j < k0) is accumulated in one sweep over the lane's own row (four
    // independent FMA chains, own-row entries loaded once for four columns); the 4 x 4 triangle
    // inside the block is then finished from registers with v_readlane broadcasts -- straight-line
    // code, no LDS round trip and no branch on the column-to-column dependency: every lane takes
    // the reciprocal square root of its own entry and lane k's is broadcast.
    // The factor is stored strictly below the diagonal; the diagonal slot and the pad entries of
    // a row are (re)written as zeros and 1/L[k][k] stays in a register of lane k (myinvd).
    // Pivot guard: a non-positive pivot (float64 breakdown of the normal equations) freezes that
    // coordinate for this Newton step (zero column, 1/L = 1e-32) instead of poisoning the factor.
    MPCQP_HD_CHOL void cholesky() {
        if (d.nZ > WAVE) { cholesky_big(); return; }      // (a compile-time branch with compile-time dims)
        MPCQP_SETPRIO(1, 1);
        cholesky_();
        MPCQP_SETPRIO(1, 0);
    }
    MPCQP_HD_CHOL void cholesky_() {
        MPCQP_RELANE(8);
        MPCQP_TIC();
        const int n = d.nZ;
        const int i = w.lane;
        const bool act = i < n;
        const int rowi = pk(act ? i : 0, 0);
        const double dia = act ? Phi[rowi + i] : 1.0;
        const double thr = 1e-14 * fabs(dia);        // lanes >= n: entries 0, never above thr
        const double* zero4 = sm + c.zero;
        myinvd = 0.0;
        constexpr int CB = 4;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (one_row_per_lane<DM>()) {
#if MPCQP_CHOL_REDUNDANT
            if (act) dz[i] = thr;                 // read back four at a time by chol_static (dz is free here)
            w.sync();
#endif
            chol_static<0>(thr);
#if MPCQP_CHOL_REDUNDANT && MPCQP_CHOL_INVD != 0
            myinvd = act ? gt[i] : 0.0;           // (gt is free during a factorisation)
            w.sync();
#endif
#if MPCQP_CHOL_LDL && !MPCQP_CHOL_REDUNDANT
            {
                chol_broke = w.any(act && !(myd_ > thr));            // a pivot at or below its threshold, or NaN
                myinvd = act ? rcp(myd_) : 0.0;
                w.sync();
                MPCQP_TOC(6);
                return;
            }
#elif MPCQP_CHOL_DIAG && !MPCQP_CHOL_REDUNDANT
            {
                // L_ii from the diagonal slot -> 1/L_ii in the lane's register, zero in the slot (what the sweeps expect)
                const double Ld = act ? Phi[rowi + i] : 1.0;
                if (act) Phi[rowi + i] = 0.0;
                chol_broke = w.any(act && !(Ld * Ld > thr));         // a floored (or NaN) pivot
                myinvd = act ? rcp(Ld) : 0.0;
                w.sync();
                MPCQP_TOC(6);
                return;
            }
#endif
            myinvd = fmx(myinvd, 1e-32);
            chol_broke = w.any(act && myinvd <= 1e-32);
            MPCQP_TOC(6);
            return;
        }
#endif
        MPCQP_NOUNROLL
        for (int k0 = 0; k0 < n; k0 += CB) {
            const bool mine = act && i >= k0;
            int j0 = 0;                                  // first column the sweep below has to cover
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (one_row_per_lane<DM>()) {
                if (k0 == 16) chol_panel_update<1>();
                else if (k0 == 32) chol_panel_update<2>();
                else if (k0 == 48) chol_panel_update<3>();
                j0 = k0 & ~15;                           // columns < 16P are already accounted for
            }
#endif
            double v[CB];
            load4(mine ? Phi + rowi + k0 : zero4, v);     // entries right of the diagonal are pads = 0
            if (mine && k0 > j0) {
                const double* Li = Phi + rowi;
                const int rb = pk(k0, 0), rs = k0 + CB;   // rows k0..k0+3 share the stride k0+4
                const double* Lk[CB];
                MPCQP_UNROLL
                for (int cc = 0; cc < CB; ++cc) Lk[cc] = Phi + rb + (k0 + cc < n ? cc : n - 1 - k0) * rs;
                _Pragma("unroll 2")
                for (int j = j0; j < k0; j += 2) {       // j0, k0 multiples of CB: pairs are aligned
                    const double a0 = Li[j], a1 = Li[j + 1];
                    MPCQP_UNROLL
                    for (int cc = 0; cc < CB; ++cc) {
                        v[cc] -= a0 * Lk[cc][j];
                        v[cc] -= a1 * Lk[cc][j + 1];
                    }
                }
            }
            double lk[CB];
            MPCQP_UNROLL
            for (int cc = 0; cc < CB; ++cc) {
                const int k = k0 + cc;                   // k <= 63; a column k >= n only sees zeros
                const double idl = (v[cc] > thr) ? rsqrt_(v[cc]) : 0.0;   // lane k: 1/sqrt(pivot), 0 if bad
                const double idb = w.bcast(idl, k);
                lk[cc] = v[cc] * idb;                    // lane i > k: L[i][k]
                if (i == k) myinvd = idl;
                MPCQP_UNROLL
                for (int c2 = cc + 1; c2 < CB; ++c2) v[c2] -= lk[cc] * w.bcast(lk[cc], k0 + c2);   // L[i][k] L[k2][k]
            }
            if (mine) {
                MPCQP_UNROLL
                for (int cc = 0; cc < CB; ++cc) lk[cc] = (i > k0 + cc) ? lk[cc] : 0.0;
                store4(Phi + rowi + k0, lk);
            }
            w.sync();
        }
        myinvd = fmx(myinvd, 1e-32);
        chol_broke = w.any(act && myinvd <= 1e-32);
        MPCQP_TOC(6);
    }

    MPCQP_HD static double rsqrt_(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
        const double r = __builtin_amdgcn_rsq(x);          // 5e-8 relative
        return r * fma(-0.5 * x * r, r, 1.5);              // one Newton step: 4e-15 (see rcp())
#else
        return 1.0 / sqrt(x);
#endif
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // ---- triangular solves of the compile-time-dims kernels (nZ <= 64), blocked by the 16-lane DPP rows -----------
    // Lane i owns unknown i, so tile T (unknowns 16T .. 16T+15) sits in DPP row T.  Inside a tile the substitution is a
    // chain of v_fmac_f64_dpp: "r += r[lane k of my row] * c_k" -- the broadcast is the DPP modifier of the multiply-add,
    // the row mask keeps the other rows out, no v_readlane / scalar operand / hazard nops of a cross-wave broadcast.
    // The finished tile is then copied to every row (one ds_bpermute pair) and the rows still to come (forward: below,
    // backward: above) take their whole 16-column update from that copy, again with the row broadcast in the
    // multiply-add and two independent accumulators.  Same arithmetic per unknown as the column sweeps of
    // solve_into_dz (sums in a different order).  c: the lane's own factor entries times -1/L_ii.
    // raw factor entries of tile T for the forward sweep: the lane's own row, columns 16T.. (zeros on and right of the
    // diagonal; rows above the tile read the zero slot)
    template <int T>
    __device__ __forceinline__ void solve_fwd_load(double (*cf)[4], const int rowi, const bool act) {
        constexpr int n = DM::nZ, K0 = 16 * T, KT = (n - K0 < 16) ? n - K0 : 16, NC = (KT + 3) / 4;
        const int i = w.lane;
        MPCQP_UNROLL
        for (int u = 0; u < NC; ++u) load4((act && i >= K0 + 4 * u) ? Phi + rowi + K0 + 4 * u : sm + c.zero, cf[u]);
    }
    // ... for the backward sweep: L[k][i] of the rows k of tile T (0 on and right of the diagonal: zero diagonal slot and
    // pads; lanes i >= k0 + 4 are beyond the stored row: they read column 0, which solve_bwd_tile scales by zero)
    template <int T>
    __device__ __forceinline__ void solve_bwd_load(double (*cb)[4]) {
        constexpr int n = DM::nZ, K0 = 16 * T, KT = (n - K0 < 16) ? n - K0 : 16, NC = (KT + 3) / 4;
        const int i = w.lane;
        MPCQP_UNROLL
        for (int u = 0; u < NC; ++u) {
            const int k0 = K0 + 4 * u;
            // (M D M' form with enough zero blocks in front of the Sigma table: entries used as stored, the lanes beyond the
            //  chunk's rows read zeros there -- at every offset e (k0 + 4) of the strided reads)
            const double* p = solve_zero_region() ? ((i < k0 + 4 && i < n) ? Phi + pk(k0, 0) + i : sm + c.S)
                                                  : Phi + pk(k0, 0) + (i < k0 + 4 ? i : 0);
            MPCQP_UNROLL
            for (int e = 0; e < 4; ++e) cb[u][e] = (k0 + e < n) ? p[e * (k0 + 4)] : 0.0;
        }
    }
    template <int T>
    __device__ __forceinline__ void solve_fwd_tile(double& r, double (*cf)[4], const int rowi, const bool act, const double nm) {
        constexpr int n = DM::nZ, NT = (n + 15) / 16, K0 = 16 * T, KT = (n - K0 < 16) ? n - K0 : 16, NC = (KT + 3) / 4;
        double nx[4][4];                                    // the next tile (or the first one of the backward sweep), in flight
        if constexpr (T + 1 < NT) solve_fwd_load<T + 1>(nx, rowi, act);
        else solve_bwd_load<NT - 1>(nx);
        MPCQP_SCHED_FENCE();
        MPCQP_UNROLL
        for (int u = 0; u < NC; ++u) chain_sw<1 << T, false>(u, r, cf[u]);
        if constexpr (T + 1 < NT) {
            const double y = w.fetch(r, K0 + (w.lane & 15));
            double a0 = 0.0, a1 = 0.0;
            MPCQP_UNROLL
            for (int u = 0; u < NC; ++u) rows_sw<0xf & ~((2 << T) - 1)>(u, (u & 1) ? a1 : a0, y, cf[u]);
            r += a0 + a1;
            MPCQP_SCHED_FENCE();
            if constexpr (!solve_unscaled()) {
                MPCQP_UNROLL
                for (int u = 0; u < 4; ++u) {
                    MPCQP_UNROLL
                    for (int e = 0; e < 4; ++e) nx[u][e] *= nm;
                }
            }
            solve_fwd_tile<T + 1>(r, nx, rowi, act, nm);
        } else {
            r *= myinvd;                                    // L'x = y
            solve_bwd_tile<NT - 1>(r, nx, act, nm);
        }
    }
    template <int T>
    __device__ __forceinline__ void solve_bwd_tile(double& r, double (*cb)[4], const bool act, const double nm) {
        constexpr int n = DM::nZ, K0 = 16 * T, KT = (n - K0 < 16) ? n - K0 : 16, NC = (KT + 3) / 4;
        const int i = w.lane;
        double nx[4][4];
        if constexpr (T > 0) solve_bwd_load<T - 1>(nx);
        MPCQP_SCHED_FENCE();
        if constexpr (!solve_zero_region()) {
            MPCQP_UNROLL
            for (int u = 0; u < NC; ++u) {
                const double sc = (act && i < K0 + 4 * u + 4) ? (solve_unscaled() ? 1.0 : nm) : 0.0;
                MPCQP_UNROLL
                for (int e = 0; e < 4; ++e) cb[u][e] *= sc;
            }
        }
        MPCQP_UNROLL
        for (int u = NC - 1; u >= 0; --u) chain_sw<1 << T, true>(u, r, cb[u]);
        if constexpr (T > 0) {
            const double x = w.fetch(r, K0 + (i & 15));
            double a0 = 0.0, a1 = 0.0;
            MPCQP_UNROLL
            for (int u = 0; u < NC; ++u) rows_sw<(1 << T) - 1>(u, (u & 1) ? a1 : a0, x, cb[u]);
            r += a0 + a1;
            solve_bwd_tile<T - 1>(r, nx, act, nm);
        }
    }
    // (the DPP lane is an immediate: one instantiation per chunk of four, selected by the unrolled loop index)
    template <int ROWS, bool REV>
    __device__ __forceinline__ void chain_sw(int u, double& r, const double* cc) {
        switch (u) {
            case 0: W::template chain4<0, ROWS, REV>(r, cc[0], cc[1], cc[2], cc[3]); break;
            case 1: W::template chain4<4, ROWS, REV>(r, cc[0], cc[1], cc[2], cc[3]); break;
            case 2: W::template chain4<8, ROWS, REV>(r, cc[0], cc[1], cc[2], cc[3]); break;
            default: W::template chain4<12, ROWS, REV>(r, cc[0], cc[1], cc[2], cc[3]); break;
        }
    }
    template <int ROWS>
    __device__ __forceinline__ void rows_sw(int u, double& a, double x, const double* cc) {
        switch (u) {
            case 0: W::template fmabc4<0, ROWS>(a, x, cc[0], cc[1], cc[2], cc[3]); break;
            case 1: W::template fmabc4<4, ROWS>(a, x, cc[0], cc[1], cc[2], cc[3]); break;
            case 2: W::template fmabc4<8, ROWS>(a, x, cc[0], cc[1], cc[2], cc[3]); break;
            default: W::template fmabc4<12, ROWS>(a, x, cc[0], cc[1], cc[2], cc[3]); break;
        }
    }
    // the factor is M D M' with the negated unit-triangular M stored (chol_static, MPCQP_CHOL_LDL): the sweeps use its entries
    // as they are.  Needs zeros at the strided offsets of solve_bwd_load, i.e. enough zero blocks in front of the Sigma table.
    static constexpr bool solve_unscaled() { return MPCQP_CHOL_LDL && !MPCQP_CHOL_REDUNDANT; }
    // ... and the lanes beyond a chunk's rows of the backward sweep read zeros from the zero blocks in front of the Sigma
    // table (enough of them for the strided offsets e (k0 + 4)); otherwise their entries are multiplied by zero
    static constexpr bool solve_zero_region() { return solve_unscaled() && DM::zpad * DM::sp >= 3 * (DM::nZ + 3) + 4; }
    __device__ __forceinline__ void solve_static() {
        constexpr int n = DM::nZ;
        const int i = w.lane;
        const bool act = i < n;
        const int rowi = pk(act ? i : 0, 0);
        const double nm = act ? -myinvd : 0.0;
        double r = act ? gt[i] : 0.0;
        if constexpr (!(MPCQP_CHOL_LDL && !MPCQP_CHOL_REDUNDANT)) r *= myinvd;      // LL': L y = r in the variable scaled by 1/L_ii
        double cf[4][4];
        solve_fwd_load<0>(cf, rowi, act);
        if constexpr (!solve_unscaled()) {
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u) {
                MPCQP_UNROLL
                for (int e = 0; e < 4; ++e) cf[u][e] *= nm;
            }
        }
        solve_fwd_tile<0>(r, cf, rowi, act, nm);            // (runs on into the backward sweep)
        if (act) dz[i] = r;
        w.sync();
    }
#endif

    // ---- dz <- Phi^{-1} gt  (factor in Phi / myinvd) --------------------------------------------
    // Column sweeps in chunks of four, all lanes in lock step, no exec masking: a lane whose row is
    // already finished reads the four-zero slot instead of the factor, and inside the diagonal
    // block the zero diagonal slot / pad entries do the masking.  The chunk of the next group is fetched ahead of the dependent chain, which is then
    // v_mul -> v_readlane -> v_fma per column.
    MPCQP_HD void solve_into_dz() {
        MPCQP_RELANE(4);          // (in front of the several-rows-per-lane form too: without it the nZ~ = 106 kernel spills 748 registers instead of 6)
        if (d.nZ > WAVE) { solve_big(); return; }
        MPCQP_SETPRIO(2, 1);
        solve_into_dz_();
        MPCQP_SETPRIO(2, 0);
    }
    MPCQP_HD void solve_into_dz_() {
        MPCQP_TIC();
#if defined(__HIP_DEVICE_COMPILE__) && MPCQP_SOLVE_DPP
        if constexpr (one_row_per_lane<DM>()) {
            solve_static();
            MPCQP_TOC(7);
            return;
        }
#endif
        const int n = d.nZ;
        const int i = w.lane;
        const bool act = i < n;
        const int rowi = pk(act ? i : 0, 0);
        const int nfull = n >> 2, rem = n & 3;
        const double* zero4 = sm + c.zero;
        // The sweeps run in the scaled variable rs_i = r_i / L_ii with the lane's factor entries
        // scaled by its own 1/L_ii when they are fetched (off the dependent chain), which leaves
        // v_readlane -> v_fma per column on the chain:  y_i = rs_i after the forward sweep.
        double r = (act ? gt[i] : 0.0) * myinvd;
        double c0[4], c1[4];
        // L y = r: x[u] = L[i][k0+u] / L[i][i] (0 on/right of the diagonal; finished rows i < k0 read zeros)
        auto ldf = [&](int k0, double* x) {
            load4((act && i >= k0) ? Phi + rowi + k0 : zero4, x);
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u) x[u] *= myinvd;
        };
        ldf(0, c0);
        auto fstep = [&](int g) {
            const int k0 = 4 * g;
            if (k0 + 4 < n) ldf(k0 + 4, c1);
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u) r -= c0[u] * w.bcast(r, k0 + u);
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u) c0[u] = c1[u];
            MPCQP_SCHED_FENCE();             // one chunk fetched ahead, not all of them
        };
        if constexpr (DM::is_static) {       // straight-line code: constant row offsets and lane indices
            MPCQP_UNROLL
            for (int g = 0; g < nfull; ++g) fstep(g);
        } else {
            _Pragma("unroll 2")
            for (int g = 0; g < nfull; ++g) fstep(g);
        }
        MPCQP_UNROLL
        for (int u = 0; u < 3; ++u)
            if (u < rem) r -= c0[u] * w.bcast(r, 4 * nfull + u);
        // L' x = y in xs_i = x_i (right-hand side y_i / L_ii): x[u] = L[k0+u][i] / L[i][i] (0 for the
        // rows k0+u <= i of the group; finished lanes i >= k0+4 read zeros); rows of a group are k0+4 apart
        r *= myinvd;
        auto ldb = [&](int k0, int cnt, double* x) {
            // finished lanes (i >= k0 + 4) read a valid entry (column 0) and scale it by zero
            const bool on = act && i < k0 + 4;
            const double sc = on ? myinvd : 0.0;
            const double* p = Phi + pk(k0, 0) + (i < k0 + 4 ? i : 0);
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u) x[u] = (u < cnt) ? p[u * (k0 + 4)] * sc : 0.0;
        };
        if (rem) {
            ldb(4 * nfull, rem, c0);
            MPCQP_UNROLL
            for (int u = 2; u >= 0; --u)
                if (u < rem) r -= c0[u] * w.bcast(r, 4 * nfull + u);
        }
        if (nfull > 0) ldb(4 * (nfull - 1), 4, c0);
        auto bstep = [&](int g) {
            const int k0 = 4 * g;
            if (g > 0) ldb(k0 - 4, 4, c1);
            MPCQP_UNROLL
            for (int u = 3; u >= 0; --u) r -= c0[u] * w.bcast(r, k0 + u);
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u) c0[u] = c1[u];
            MPCQP_SCHED_FENCE();
        };
        if constexpr (DM::is_static) {
            MPCQP_UNROLL
            for (int g = nfull - 1; g >= 0; --g) bstep(g);
        } else {
            _Pragma("unroll 2")
            for (int g = nfull - 1; g >= 0; --g) bstep(g);
        }
        if (act) dz[i] = r;
        w.sync();
        MPCQP_TOC(7);
    }

    // ---- nZ > 64 (runtime-dims kernel only): lane l owns the rows l, l + 64, ...  ----------------
    // Left-looking column Cholesky in place.  For column k every lane forms, for each of its rows
    // i >= k, the dot product of row i and row k over the finished columns j < k (both contiguous in
    // the packed layout, row k a broadcast read; register accumulation, no LDS read-modify-write
    // chain) and stores Phi[i][k] - dot; the owner of row k turns its entry into 1/L[k][k] (LDS vector
    // dinv, same pivot guard as cholesky()), then the column is scaled.  The factor is stored
    // strictly below the diagonal.
    // columns kb .. ke-1, the dot products over the columns j0 .. k-1 (j0 a multiple of four; the columns before j0 have
    // been accounted for by chol_panel_update)
    MPCQP_HD void chol_big_columns(int kb, int ke, int j0, bool& broke) {
        const int n = d.nZ;
        double* dinv = sm + c.dinv;
        MPCQP_NOUNROLL
        for (int k = kb; k < ke; ++k) {
            const double* Lk = Phi + pk(k, 0);
            const int k4 = k & ~3;
            for (int i = w.lane + ((k - w.lane + WAVE - 1) / WAVE) * WAVE; i < n; i += WAVE) {   // first owned row >= k
                const double* Li = Phi + pk(i, 0);
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                MPCQP_UNROLL4
                for (int j = j0; j < k4; j += 4) {
                    double x[4], y[4];
                    load4(Li + j, x); load4(Lk + j, y);
                    a0 = fma(x[0], y[0], a0); a1 = fma(x[1], y[1], a1);
                    a2 = fma(x[2], y[2], a2); a3 = fma(x[3], y[3], a3);
                }
                for (int j = k4; j < k; ++j) a0 = fma(Li[j], Lk[j], a0);
                const double orig = Li[k];
                const double v = orig - ((a0 + a1) + (a2 + a3));
                if (i == k) {
                    const bool ok = v > 1e-14 * fabs(orig);
                    broke = broke || !ok;
                    dinv[k] = ok ? rsqrt_(v) : 0.0;
                    Phi[pk(i, k)] = 0.0;         // (zero diagonal slot, like the one-row-per-lane factor: solve_big_static)
                } else {
                    Phi[pk(i, k)] = v;
                }
            }
            w.sync();
            const double idl = dinv[k];
            for (int i = w.lane + ((k + 1 - w.lane + WAVE - 1) / WAVE) * WAVE; i < n; i += WAVE) Phi[pk(i, k)] *= idl;
            w.sync();
            if (w.lane == 0) dinv[k] = fmax(idl, 1e-32);
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // compile-time dims: 16-column panels; the contribution of the finished panels to a panel's columns is removed on the
    // matrix cores (chol_panel_update: every row tile below, one v_mfma_f64_16x16x4 per four finished columns), the
    // left-looking dot products then only run inside the panel (at most 15 terms instead of up to nZ~)
    // The 16 columns of a panel, factored in registers: lane l holds the panel entries of its rows l, l + 64, .. (16
    // doubles per row); the pivot row of column k = 16P + c lives in slot (16P)/64, lane (16P)%64 + c -- compile-time
    // constants -- so the column step is chol_static's: reciprocal square root of the lane's own entry, v_readlane
    // broadcasts of the pivot lane's, one multiply-add per later column and row slot.  No LDS round trip and no fence
    // inside the panel (the column-at-a-time loop of the runtime dims needs two fences and several dependent LDS reads
    // per column).  thr: pivot thresholds of the lane's rows (1e-14 of the original diagonal).
    // A team of wavefronts (W::NTEAM > 1): wavefront 0 factors the pivot slot only (NSE = so + 1: the 16 x 16 diagonal block and
    // the other rows of its slot); the row slots below take their 16 panel entries through a triangular solve with the
    // diagonal block read back from LDS (wave-uniform addresses: broadcast reads, no v_readlane), one slot per wavefront
    // (chol_big_panel_rows).  profiles/r6c: the one-wavefront form took 11k cycles per panel at nZ~ = 151 (three slots,
    // 240 v_readlane + 408 FP64 instructions per panel on the pivot chain's wavefront), 110k of the 347k cycles of an iteration.
    template <int P, int NS>
    __device__ __forceinline__ void chol_big_panel_rows() {
        constexpr int n = DM::nZ, K0 = 16 * P, KC = (n - K0 < 16) ? n - K0 : 16, so = K0 / WAVE;
        if constexpr (KC == 16) {                      // (a shorter last panel has no rows below it)
            const double* dinv = sm + c.dinv;
            auto lk = [&](int r, int cc) { return Phi[pk(K0 + r, 0) + K0 + cc]; };      // L[K0 + r][K0 + cc], wave-uniform
            // (diagonal-block form: the pivot slot's rows below the block are rows like any other, chol_big_panel_diag)
            constexpr int s0 = (MPCQP_TEAM_DIAG && W::NTEAM >= 3) ? so : so + 1;
            MPCQP_UNROLL
            for (int s_ = s0; s_ < NS; ++s_) {
                if ((s_ - s0 + 1) % W::NTEAM != W::WV) continue;         // (first slot to wavefront 1, ..: wavefront 0 last)
                const int i = w.lane + WAVE * s_;
                const bool mine = i < n && i >= K0 + 16;
                double* row = Phi + pk(mine ? i : 0, 0) + K0;
                double x[16];
                MPCQP_UNROLL
                for (int u = 0; u < 4; ++u) load4(mine ? row + 4 * u : sm + c.zero, &x[4 * u]);
#if MPCQP_PANELROWS_RL
                // right-looking, one column per scheduling region: the 15 - cc updates of a column are independent
                // multiply-adds (the compiler otherwise orders the work by target column: 120 multiply-adds in chains on one
                // accumulator each); the block's next column is requested while the current one is applied
                double lc[16], ln[16];
                MPCQP_UNROLL
                for (int c2 = 1; c2 < 16; ++c2) lc[c2] = lk(c2, 0);
                MPCQP_UNROLL
                for (int cc = 0; cc < 16; ++cc) {
                    MPCQP_UNROLL
                    for (int c2 = cc + 2; c2 < 16; ++c2) ln[c2] = lk(c2, cc + 1);
                    const double dv = dinv[K0 + cc];
                    x[cc] *= dv > 1e-32 ? dv : 0.0;                   // (a pivot below its threshold: zero column, as in the pivot slot)
                    MPCQP_UNROLL
                    for (int c2 = cc + 1; c2 < 16; ++c2) x[c2] = fma(-x[cc], lc[c2], x[c2]);
                    MPCQP_SCHED_FENCE();
                    MPCQP_UNROLL
                    for (int c2 = cc + 2; c2 < 16; ++c2) lc[c2] = ln[c2];
                }
#else
                MPCQP_UNROLL
                for (int cc = 0; cc < 16; ++cc) {
                    const double dv = dinv[K0 + cc];
                    x[cc] *= dv > 1e-32 ? dv : 0.0;                   // (a pivot below its threshold: zero column, as in the pivot slot)
                    MPCQP_UNROLL
                    for (int c2 = cc + 1; c2 < 16; ++c2) x[c2] = fma(-x[cc], lk(c2, cc), x[c2]);
                    if (cc % 4 == 3) MPCQP_SCHED_FENCE();            // (bounds the broadcast reads in flight: registers)
                }
#endif
                if (mine) {
                    MPCQP_UNROLL
                    for (int u = 0; u < 4; ++u) store4(row + 4 * u, &x[4 * u]);
                }
            }
        }
    }
    template <int P>
    __device__ __forceinline__ void panel_rows_dispatch(int p) {
        if constexpr (P < (DM::nZ + 15) / 16) {
            if (p == P) chol_big_panel_rows<P, (DM::nZ + WAVE - 1) / WAVE>();
            else panel_rows_dispatch<P + 1>(p);
        }
    }
    // The 16 x 16 diagonal block of panel P alone, in the 16 lanes of its DPP row (16 P is a multiple of 16, so the block's rows
    // are one row of lanes): the column step broadcasts inside the row with the DPP modifier of the instructions themselves
    // (v_mov_b64_dpp / v_fmac_f64_dpp row_newbcast) -- one instruction per (column, later column) pair where the v_readlane
    // form of chol_big_panel_slots takes three.  The other lanes run along on zeros.
    template <int P, int NS>
    __device__ __forceinline__ void chol_big_panel_diag(const double (&thr)[NS], bool& broke) {
        constexpr int K0 = 16 * P, so = K0 / WAVE, lb = K0 % WAVE;
        const int lr = w.lane & 15;
        const bool inblk = (w.lane >> 4) == (lb >> 4);
        double* row = Phi + pk(K0 + lr, 0) + K0;
        double* dinv = sm + c.dinv;
        double v[16];
        MPCQP_UNROLL
        for (int u = 0; u < 4; ++u) load4((inblk && lr >= 4 * u) ? row + 4 * u : sm + c.zero, &v[4 * u]);
        double mydinv = 0.0;
        MPCQP_UNROLL
        for (int cc = 0; cc < 16; ++cc) chol_diag_col(cc, v, thr[so], lr, mydinv);
        MPCQP_UNROLL
        for (int cc = 0; cc < 16; ++cc) v[cc] = (lr > cc) ? v[cc] : 0.0;       // zeros on and right of the diagonal
        MPCQP_UNROLL
        for (int u = 0; u < 4; ++u)
            if (inblk && lr >= 4 * u) store4(row + 4 * u, &v[4 * u]);
        if (inblk) dinv[K0 + lr] = fmx(mydinv, 1e-32);
        broke = broke || (inblk && mydinv <= 1e-32);
        w.sync();
    }
    __device__ __forceinline__ void chol_diag_col(int cc, double (&v)[16], double thr, int lr, double& mydinv) {
        // (cc is a compile-time value after unrolling; the DPP lane is an immediate: one case per column)
        switch (cc) {
#define MPCQP_DIAG_COL_(C) case C: chol_diag_col_<C>(v, thr, lr, mydinv); break;
            MPCQP_DIAG_COL_(0) MPCQP_DIAG_COL_(1) MPCQP_DIAG_COL_(2) MPCQP_DIAG_COL_(3) MPCQP_DIAG_COL_(4) MPCQP_DIAG_COL_(5)
            MPCQP_DIAG_COL_(6) MPCQP_DIAG_COL_(7) MPCQP_DIAG_COL_(8) MPCQP_DIAG_COL_(9) MPCQP_DIAG_COL_(10) MPCQP_DIAG_COL_(11)
            MPCQP_DIAG_COL_(12) MPCQP_DIAG_COL_(13) MPCQP_DIAG_COL_(14) MPCQP_DIAG_COL_(15)
#undef MPCQP_DIAG_COL_
            default: break;
        }
    }
    template <int C>
    __device__ __forceinline__ void chol_diag_col_(double (&v)[16], double thr, int lr, double& mydinv) {
        const double piv = v[C];
        const double idl = (piv > thr) ? rsqrt_(piv) : 0.0;          // lane C of the row: 1/sqrt(pivot), 0 if bad
        const double idb = W::template rowbc<C>(idl);
        mydinv = (lr == C) ? idl : mydinv;
        v[C] *= idb;                                                 // rows below the pivot: L[i][k]
        chol_diag_upd_<C, C + 1>(v);
    }
    template <int C, int C2>
    __device__ __forceinline__ void chol_diag_upd_(double (&v)[16]) {
        if constexpr (C2 < 16) {
            W::template rowbc_fms<C2, C2 == C + 1>(v[C2], v[C], v[C]);       // v[c2] -= L[K0 + c2][k] * L[i][k]
            chol_diag_upd_<C, C2 + 1>(v);
        }
    }
    template <int P, int NS>
    __device__ __forceinline__ void chol_big_panel_reg(const double (&thr)[NS], bool& broke) {
        if constexpr (W::NTEAM > 1) {
            constexpr int so_ = (16 * P) / WAVE;
            // three or more wavefronts: the diagonal block alone on wavefront 0, every other row through the team's triangular
            // solves; two: the whole pivot slot on wavefront 0, the slots below on the team (measured, profiles/r6j: nZ~ = 151,
            // four wavefronts 31.3 -> 29.7 ms per 4096 with the diagonal-block form; nZ~ = 106, two wavefronts 19.3 -> 20.0 ms)
            constexpr bool DIAG = MPCQP_TEAM_DIAG && W::NTEAM >= 3;
            if constexpr (DM::nZ - 16 * P >= 16) {
                if constexpr (DIAG) chol_big_panel_diag<P, NS>(thr, broke);                 // (ends with a wave fence: its stores are issued)
                else chol_big_panel_slots<P, NS, (so_ + 1 < NS ? so_ + 1 : NS)>(thr, broke);
                if constexpr ((DIAG ? DM::nZ - 16 * P > 16 : so_ + 1 < NS) && !(MPCQP_ABLATE & 1024)) {
                    w.post(TJ_PANELROWS, P);
                    chol_big_panel_rows<P, NS>();
                    w.join();
                }
            } else {
                chol_big_panel_slots<P, NS, (so_ + 1 < NS ? so_ + 1 : NS)>(thr, broke);      // (a shorter last panel: nothing below it)
            }
        } else {
            chol_big_panel_slots<P, NS, NS>(thr, broke);
        }
    }
    // (row slots so .. NSE-1 of the panel)
    template <int P, int NS, int NSE>
    __device__ __forceinline__ void chol_big_panel_slots(const double (&thr)[NS], bool& broke) {
        constexpr int n = DM::nZ, K0 = 16 * P, KC = (n - K0 < 16) ? n - K0 : 16, NCH = (KC + 3) / 4;
        constexpr int so = K0 / WAVE, lb = K0 % WAVE;
        const double* zero4 = sm + c.zero;
        double* dinv = sm + c.dinv;
        double v[NS][16];
        bool mine[NS];
        int rowo[NS];
        MPCQP_UNROLL
        for (int s_ = so; s_ < NSE; ++s_) {
            const int i = w.lane + WAVE * s_;
            mine[s_] = i < n && i >= K0;
            rowo[s_] = pk(mine[s_] ? i : 0, 0) + K0;
            MPCQP_UNROLL
            for (int u = 0; u < 4; ++u) {
                if (u < NCH) load4((mine[s_] && i >= K0 + 4 * u) ? Phi + rowo[s_] + 4 * u : zero4, &v[s_][4 * u]);
                else { v[s_][4 * u] = v[s_][4 * u + 1] = v[s_][4 * u + 2] = v[s_][4 * u + 3] = 0.0; }
            }
        }
        double mydinv = 0.0;
        MPCQP_UNROLL
        for (int cc = 0; cc < KC; ++cc) {
            const double piv = v[so][cc];
            const double idl = (piv > thr[so]) ? rsqrt_(piv) : 0.0;       // lane lb + cc: 1/sqrt(pivot), 0 if bad
            const double idb = w.bcast(idl, lb + cc);
            mydinv = (w.lane == lb + cc) ? idl : mydinv;
            MPCQP_UNROLL
            for (int s_ = so; s_ < NSE; ++s_) v[s_][cc] *= idb;            // rows below the pivot: L[i][k]
            MPCQP_UNROLL
            for (int c2 = cc + 1; c2 < KC; ++c2) {
                const double bv = w.bcast(v[so][cc], lb + c2);            // L[K0 + c2][k]
                MPCQP_UNROLL
                for (int s_ = so; s_ < NSE; ++s_) v[s_][c2] = fma(-v[s_][cc], bv, v[s_][c2]);
            }
        }
        // rows of the panel itself keep zeros on and right of the diagonal; stores only where the row is that long
        MPCQP_UNROLL
        for (int cc = 0; cc < KC; ++cc) v[so][cc] = (w.lane > lb + cc) ? v[so][cc] : 0.0;
        MPCQP_UNROLL
        for (int s_ = so; s_ < NSE; ++s_) {
            const int i = w.lane + WAVE * s_;
            MPCQP_UNROLL
            for (int u = 0; u < NCH; ++u)
                if (mine[s_] && i >= K0 + 4 * u) store4(Phi + rowo[s_] + 4 * u, &v[s_][4 * u]);
        }
        const bool own = w.lane >= lb && w.lane < lb + KC;
        if (own) dinv[K0 + w.lane - lb] = fmx(mydinv, 1e-32);
        broke = broke || (own && mydinv <= 1e-32);
        w.sync();
    }
    template <int P, int NS>
    __device__ __forceinline__ void chol_big_panels(const double (&thr)[NS], bool& broke) {
        constexpr int n = DM::nZ, K0 = 16 * P;
        if constexpr (K0 < n) {
            if constexpr (P > 0) { if (!(MPCQP_ABLATE & 256)) chol_panel_update<P>(); }
            if (!(MPCQP_ABLATE & 512)) chol_big_panel_reg<P, NS>(thr, broke);
            chol_big_panels<P + 1, NS>(thr, broke);
        }
    }
#endif
    MPCQP_HD void cholesky_big() {
        MPCQP_TIC();
        bool broke = false;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DM::is_static) {
            constexpr int NS = (DM::nZ + WAVE - 1) / WAVE;
            double thr[NS];
            MPCQP_UNROLL
            for (int s_ = 0; s_ < NS; ++s_) {
                const int i = w.lane + WAVE * s_;
                thr[s_] = i < DM::nZ ? 1e-14 * fabs(Phi[pk(i, i)]) : 1.0;
            }
            chol_big_panels<0, NS>(thr, broke);
        } else {
            chol_big_columns(0, d.nZ, 0, broke);
        }
#else
        chol_big_columns(0, d.nZ, 0, broke);
#endif
        w.sync();
        chol_broke = w.any(broke);
        MPCQP_TOC(6);
    }

    // dz <- Phi^{-1} gt with the factor of cholesky_big(): column sweeps on the LDS vector dz
#if defined(__HIP_DEVICE_COMPILE__)
    // Compile-time dims, nZ~ > 64: the rows lane, lane + 64, .. of the right-hand side stay in registers and the
    // substitutions run like solve_into_dz's -- chunks of four columns, the lane's own factor entries (forward: its row,
    // backward: the column read along the rows of a four-row group) scaled by its 1/L_ii, v_readlane broadcasts of the
    // four unknowns -- instead of one LDS round trip (vector dz) and two fences per column.
    __device__ __forceinline__ void solve_big_static() {
        constexpr int n = DM::nZ, NS = (n + WAVE - 1) / WAVE;
        const double* dinv = sm + c.dinv;
        const double* zero4 = sm + c.zero;
        double r[NS], di[NS];
        int rowo[NS];
        bool act[NS];
        MPCQP_UNROLL
        for (int s_ = 0; s_ < NS; ++s_) {
            const int i = w.lane + WAVE * s_;
            act[s_] = i < n;
            rowo[s_] = pk(act[s_] ? i : 0, 0);
            di[s_] = act[s_] ? dinv[i] : 0.0;
            r[s_] = (act[s_] ? gt[i] : 0.0) * di[s_];
        }
        // L y = r in the variable scaled by 1/L_ii
        MPCQP_UNROLL
        for (int so = 0; so < NS; ++so) {
            const int nch = (((n - WAVE * so < WAVE) ? n - WAVE * so : WAVE) + 3) / 4;
            MPCQP_PRAGMA(unroll MPCQP_SOLVEBIG_UNROLL)
            for (int g = 0; g < nch; ++g) {
                const int k0 = WAVE * so + 4 * g;
                double x[NS][4];
                MPCQP_UNROLL
                for (int s_ = so; s_ < NS; ++s_) {
                    const int i = w.lane + WAVE * s_;
                    load4((act[s_] && i >= k0) ? Phi + rowo[s_] + k0 : zero4, x[s_]);
                    MPCQP_UNROLL
                    for (int u = 0; u < 4; ++u) x[s_][u] *= di[s_];
                }
                MPCQP_UNROLL
                for (int u = 0; u < 4; ++u) {
                    const double bv = w.bcast(r[so], 4 * g + u);
                    MPCQP_UNROLL
                    for (int s_ = so; s_ < NS; ++s_) r[s_] = fma(-x[s_][u], bv, r[s_]);
                }
            }
        }
        // L'x = y
        MPCQP_UNROLL
        for (int s_ = 0; s_ < NS; ++s_) r[s_] *= di[s_];
        MPCQP_UNROLL
        for (int so = NS - 1; so >= 0; --so) {
            const int nch = (((n - WAVE * so < WAVE) ? n - WAVE * so : WAVE) + 3) / 4;
            MPCQP_PRAGMA(unroll MPCQP_SOLVEBIG_UNROLL)
            for (int g = nch - 1; g >= 0; --g) {
                const int k0 = WAVE * so + 4 * g;
                const double* pg = Phi + pk(k0, 0);
                double x[NS][4];
                MPCQP_UNROLL
                for (int s_ = 0; s_ <= so; ++s_) {
                    const int i = w.lane + WAVE * s_;
                    const bool on = act[s_] && i < k0 + 4;        // (beyond the stored rows: column 0, scaled by zero)
                    const double sc = on ? di[s_] : 0.0;
                    const double* p_ = pg + (i < k0 + 4 ? i : 0);
                    MPCQP_UNROLL
                    for (int u = 0; u < 4; ++u) x[s_][u] = (k0 + u < n) ? p_[u * (k0 + 4)] * sc : 0.0;
                }
                MPCQP_UNROLL
                for (int u = 3; u >= 0; --u) {
                    const double bv = w.bcast(r[so], 4 * g + u);
                    MPCQP_UNROLL
                    for (int s_ = 0; s_ <= so; ++s_) r[s_] = fma(-x[s_][u], bv, r[s_]);
                }
            }
        }
        MPCQP_UNROLL
        for (int s_ = 0; s_ < NS; ++s_)
            if (act[s_]) dz[w.lane + WAVE * s_] = r[s_];
        w.sync();
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    // Round 6: the same substitutions blocked by the 16-lane DPP rows, like solve_static's (one row per lane): unknown
    // i = lane + 64 s sits in DPP row (lane >> 4) of slot s, i.e. tile (s, T) of 16 unknowns is one DPP row of register r[s].
    // Inside a tile the substitution is a chain of v_fmac_f64_dpp row_newbcast (the broadcast is the multiply-add's own
    // modifier: one instruction and two wait states per unknown); the finished tile is copied to every row (one ds_bpermute
    // pair) and all unknowns still to come take their 16-column update from the copy, again with the row broadcast in the
    // multiply-add.  The column-at-a-time form above pays two v_readlane, their scalar-operand hazards and a multiply-add on
    // the dependent chain of EVERY unknown: 80 cycles per unknown and sweep measured at nZ~ = 151 (profiles/r6i).
    template <int NS, int SO, int T>
    __device__ __forceinline__ void sbd_fwd(double (&r)[NS], const double (&ndi)[NS], const int (&rowo)[NS], const bool (&act)[NS]) {
        constexpr int n = DM::nZ, K0 = WAVE * SO + 16 * T;
        if constexpr (K0 < n) {
            constexpr int KT = (n - K0 < 16) ? n - K0 : 16, NC = (KT + 3) / 4;
            const double* zero4 = sm + c.zero;
            double cf[NS][4][4];
            MPCQP_UNROLL
            for (int s_ = SO; s_ < NS; ++s_) {
                const int i = w.lane + WAVE * s_;
                MPCQP_UNROLL
                for (int u = 0; u < NC; ++u) {
                    load4((act[s_] && i >= K0 + 4 * u) ? Phi + rowo[s_] + K0 + 4 * u : zero4, cf[s_][u]);
                    MPCQP_UNROLL
                    for (int e = 0; e < 4; ++e) cf[s_][u][e] *= ndi[s_];
                }
            }
            MPCQP_UNROLL
            for (int u = 0; u < NC; ++u) chain_sw<1 << T, false>(u, r[SO], cf[SO][u]);
            if constexpr (K0 + 16 < n) {
                const double y = w.fetch(r[SO], 16 * T + (w.lane & 15));
                if constexpr (T < 3 && WAVE * SO + 16 * (T + 1) < n) {
                    double a0 = 0.0, a1 = 0.0;
                    MPCQP_UNROLL
                    for (int u = 0; u < NC; ++u) rows_sw<0xf & ~((2 << T) - 1)>(u, (u & 1) ? a1 : a0, y, cf[SO][u]);
                    r[SO] += a0 + a1;
                }
                MPCQP_UNROLL
                for (int s_ = SO + 1; s_ < NS; ++s_) {
                    double a0 = 0.0, a1 = 0.0;
                    MPCQP_UNROLL
                    for (int u = 0; u < NC; ++u) rows_sw<0xf>(u, (u & 1) ? a1 : a0, y, cf[s_][u]);
                    r[s_] += a0 + a1;
                }
            }
            MPCQP_SCHED_FENCE();
            if constexpr (T < 3) sbd_fwd<NS, SO, T + 1>(r, ndi, rowo, act);
            else sbd_fwd<NS, SO + 1, 0>(r, ndi, rowo, act);
        }
    }
    template <int NS, int SO, int T>
    __device__ __forceinline__ void sbd_bwd(double (&r)[NS], const double (&ndi)[NS], const bool (&act)[NS]) {
        constexpr int n = DM::nZ, K0 = WAVE * SO + 16 * T;
        if constexpr (K0 < n) {
            constexpr int KT = (n - K0 < 16) ? n - K0 : 16, NC = (KT + 3) / 4;
            // column entries L[k0 + e][i] of the rows k0 .. k0 + 3 of a chunk (they share the stride k0 + 4; zeros on and right
            // of the diagonal), for the lane's unknown of every slot up to SO; a lane whose unknown lies beyond the chunk's rows
            // reads column 0 and scales it by zero
            double cb[NS][4][4];
            MPCQP_UNROLL
            for (int s_ = 0; s_ <= SO; ++s_) {
                const int i = w.lane + WAVE * s_;
                MPCQP_UNROLL
                for (int u = 0; u < NC; ++u) {
                    const int k0 = K0 + 4 * u;
                    const bool on = act[s_] && i < k0 + 4;
                    const double sc = on ? ndi[s_] : 0.0;
                    const double* p_ = Phi + pk(k0, 0) + (i < k0 + 4 ? i : 0);
                    MPCQP_UNROLL
                    for (int e = 0; e < 4; ++e) cb[s_][u][e] = (k0 + e < n) ? p_[e * (k0 + 4)] * sc : 0.0;
                }
            }
            MPCQP_UNROLL
            for (int u = NC - 1; u >= 0; --u) chain_sw<1 << T, true>(u, r[SO], cb[SO][u]);
            if constexpr (K0 > 0) {
                const double x = w.fetch(r[SO], 16 * T + (w.lane & 15));
                if constexpr (T > 0) {
                    double a0 = 0.0, a1 = 0.0;
                    MPCQP_UNROLL
                    for (int u = 0; u < NC; ++u) rows_sw<(1 << T) - 1>(u, (u & 1) ? a1 : a0, x, cb[SO][u]);
                    r[SO] += a0 + a1;
                }
                MPCQP_UNROLL
                for (int s_ = 0; s_ < SO; ++s_) {
                    double a0 = 0.0, a1 = 0.0;
                    MPCQP_UNROLL
                    for (int u = 0; u < NC; ++u) rows_sw<0xf>(u, (u & 1) ? a1 : a0, x, cb[s_][u]);
                    r[s_] += a0 + a1;
                }
            }
            MPCQP_SCHED_FENCE();
        }
        if constexpr (T > 0) sbd_bwd<NS, SO, T - 1>(r, ndi, act);
        else if constexpr (SO > 0) sbd_bwd<NS, SO - 1, 3>(r, ndi, act);
    }
    __device__ __forceinline__ void solve_big_dpp() {
        constexpr int n = DM::nZ, NS = (n + WAVE - 1) / WAVE;
        const double* dinv = sm + c.dinv;
        double r[NS], di[NS], ndi[NS];
        int rowo[NS];
        bool act[NS];
        MPCQP_UNROLL
        for (int s_ = 0; s_ < NS; ++s_) {
            const int i = w.lane + WAVE * s_;
            act[s_] = i < n;
            rowo[s_] = pk(act[s_] ? i : 0, 0);
            di[s_] = act[s_] ? dinv[i] : 0.0;
            ndi[s_] = -di[s_];
            r[s_] = (act[s_] ? gt[i] : 0.0) * di[s_];            // L y = r in the variable scaled by 1/L_ii
        }
        sbd_fwd<NS, 0, 0>(r, ndi, rowo, act);
        MPCQP_UNROLL
        for (int s_ = 0; s_ < NS; ++s_) r[s_] *= di[s_];          // L'x = y
        sbd_bwd<NS, NS - 1, 3>(r, ndi, act);
        MPCQP_UNROLL
        for (int s_ = 0; s_ < NS; ++s_)
            if (act[s_]) dz[w.lane + WAVE * s_] = r[s_];
        w.sync();
    }
#endif
    MPCQP_HD void solve_big() {
        MPCQP_TIC();
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DM::is_static) {
#if MPCQP_SOLVEBIG_DPP
            solve_big_dpp();
#else
            solve_big_static();
#endif
            MPCQP_TOC(7);
            return;
        }
#endif
        const int n = d.nZ;
        const double* dinv = sm + c.dinv;
        for (int i = w.lane; i < n; i += WAVE) dz[i] = gt[i];
        w.sync();
        MPCQP_NOUNROLL
        for (int k = 0; k < n; ++k) {                    // L y = g
            const double yk = dz[k] * dinv[k];
            w.sync();
            for (int i = w.lane; i < n; i += WAVE) {
                if (i > k) dz[i] -= Phi[pk(i, k)] * yk;
                else if (i == k) dz[i] = yk;
            }
            w.sync();
        }
        MPCQP_NOUNROLL
        for (int k = n - 1; k >= 0; --k) {               // L' x = y
            const double xk = dz[k] * dinv[k];
            w.sync();
            const double* Lk = Phi + pk(k, 0);
            for (int j = w.lane; j <= k; j += WAVE) {
                if (j < k) dz[j] -= Lk[j] * xk;
                else dz[j] = xk;
            }
            w.sync();
        }
        MPCQP_TOC(7);
    }

    // rd = H̃ z + q + gt (gt = G' multipliers) with H̃ packed at Hp_ (LDS or global memory);
    // returns max |rd| and 1 + the largest term of the sum
    // rd[k] <- (H̃ z)[k] for the rows k of this wavefront's slots (team form of the loop in dual_residual)
    MPCQP_HD void Hz_rows_share(const double* Hp_) {
        const int n = d.nZ;
        for (int k = w.lane + WAVE * ((W::WV + W::NTEAM - 1) % W::NTEAM); k < n; k += WAVE * W::NTEAM) {
            double h0 = 0.0, h1 = 0.0;
            const double* Hk = Hp_ + pk(k, 0);
            int j = 0;
            MPCQP_PRAGMA(unroll MPCQP_HZ_UNROLL)
            for (; j + 1 <= k; j += 2) { h0 += Hk[j] * z[j]; h1 += Hk[j + 1] * z[j + 1]; }
            if (j <= k) h0 += Hk[j] * z[j];
            MPCQP_PRAGMA(unroll MPCQP_HZ_UNROLL)
            for (int jj = k + 1; jj < n; ++jj) h1 += Hp_[pk(jj, k)] * z[jj];
            rd[k] = h0 + h1;
        }
    }
    MPCQP_HD void dual_residual(const double* Hp_, double& rdn, double& nd_) {
        const int n = d.nZ;
        double mx = 0.0, sc = 0.0;
        if constexpr (W::NTEAM > 1) {
            if (Hp_) {                  // the rows of H̃ z over the team, then the rest as if rd held H̃ z already
                w.post(TJ_HZ, Hp_ == Phi ? 1 : 0);
                if (Hp_ == Phi) Hz_rows_share(Phi);
                else Hz_rows_share(m.Hpk + (size_t)b * d.npk);          // (the only other caller's argument: polish)
                w.join();
                Hp_ = nullptr;
            }
        }
        for (int k = w.lane; k < n; k += WAVE) {
            // H̃ z with the packed lower triangle: row k up to the diagonal is contiguous,
            // the rest of the (symmetric) row comes from column k of the rows below
            // (Hp_ == nullptr: rd holds H̃ z already, Hz_structured)
            double h0 = 0.0, h1 = 0.0;
            if (Hp_) {
                const double* Hk = Hp_ + pk(k, 0);
                int j = 0;
                MPCQP_PRAGMA(unroll MPCQP_HZ_UNROLL)
                for (; j + 1 <= k; j += 2) { h0 += Hk[j] * z[j]; h1 += Hk[j + 1] * z[j + 1]; }
                if (j <= k) h0 += Hk[j] * z[j];
                MPCQP_PRAGMA(unroll MPCQP_HZ_UNROLL)
                for (int jj = k + 1; jj < n; ++jj) h1 += Hp_[pk(jj, k)] * z[jj];
            } else {
                h0 = rd[k];
            }
            const double hz = h0 + h1;
            const double r = hz + q[k] + gt[k];
            rd[k] = r;
            mx = fmx_abs(mx, r);
            sc = fmx_abs(fmx_abs(fmx_abs(sc, q[k]), hz), gt[k]);
        }
        rdn = w.maxv(mx);
        nd_ = 1.0 + w.maxv(sc);
        w.sync();
    }

    // ---- rp, mu ; then rd = H̃ z + q + G' lam with H̃ freshly staged in Phi ---------------------
    MPCQP_HD void residuals(double& mu, double& rpn, double& rdn, double& nd_) {
        const int n = d.nZ;
        double musum = 0.0, rpmax = 0.0;
        apply_G(z, [&](Row& r, double gz) {
            const double v = gz + r.s - r.h;
            r.rp = v;
            rpmax = fmx_abs(rpmax, v);
            musum += r.s * r.lam;
        });
        mu = w.sum(musum) / wsum;
        rpn = w.maxv(rpmax);
        if (fold_H) {
            Hz_structured(true);            // (E z is still in the scratch of apply_G)
            apply_Gt([&](Row& r) { return r.lam; });
            dual_residual(nullptr, rdn, nd_);
            return;
        }
        apply_Gt([&](Row& r) { return r.lam; });
        MPCQP_TIC();
        load_H();
        dual_residual(Phi, rdn, nd_);
        MPCQP_TOC(2);
    }

    // ---- active-set polish of an interior-point iterate ------------------------------------------
    // Method of multipliers (augmented Lagrangian, rho = 1e10) for the inequality-constrained QP,
    // started from the interior-point partition A = {i: lam_i > s_i}, l = lam on A.  With the rows of
    // A treated as equalities and the first-order multiplier estimate l^ = l + rho r, r = G z - h,
    // a round is
    //     (H̃ + rho G_A'G_A) dz = -(H̃ z + q̃ + G_A'l^),   z += dz,   l <- l^ + rho G_A dz,   r += G dz
    // i.e. one G'l^, one H̃ z (H̃ from global memory), one pair of triangular solves and one G dz.
    // After a round the working set follows the multiplier rule of the method, A <- {i: l_i + rho r_i > 0}:
    // a row of A whose multiplier went negative is dropped, a violated row outside A is added, and
    // the matrix is factorised again (at most three times; interior-point partitions of degenerate
    // vertices, where weakly active rows have s_i ~ lam_i, need it; most problems never do).
    // Once r_A is at the float64 floor the last round evaluates r_d = H̃ z + q̃ + G_A'l instead, and the
    // point is accepted if r_d is at its floor too, r_A re-evaluated exactly still is, l >= 0 on A and
    // every other row is feasible: it then satisfies the KKT conditions of the QP, whatever iterate the
    // polish started from.  The residuals are evaluated exactly every round, so the rounds are
    // self-correcting and end at the floor of the residual evaluation, not at that of the
    // ill-conditioned interior-point normal equations.  If the polish fails, z is restored and the
    // interior-point iteration goes on from exactly re-evaluated residuals.
    // Row registers during the polish: rp = 1/0 (row in A), gd = r, pp = l; s and lam are left alone.
    // Vectors dz, gt, rd and Phi are used.
    // Measured (C port, 65536 C3 instances): mean 12.4 instead of 13.5 factorisations including the
    // polish's own, worst error against the certified optimum 6e-9 (1e-8 without).
    MPCQP_HD bool polish(int& nfact) {
        const int n = d.nZ;
        const double rho = 1e10;
        double zkeep = 0.0;                       // compile-time dims: nZ <= 64, one entry per lane
        double* const zb = sm + c.zb;
        constexpr bool ZREG = one_row_per_lane<DM>();      // one entry per lane: the kept iterate is a register
        if constexpr (ZREG) zkeep = (w.lane < n) ? z[w.lane] : 0.0;
        else
            for (int k = w.lane; k < n; k += WAVE) zb[k] = z[k];
        for_rows([&](int, int, Row& r) {
            const bool on = fin(r) && r.lam > r.s;
            r.rp = on ? 1.0 : 0.0;
            r.pp = on ? r.lam : 0.0;
        });
        bool ok = false, fresh = true;
        MPCQP_NOUNROLL
        for (int fact = 0; fact < MPCQP_POLISH_FACTS && !ok; ++fact) {
            ++nfact;
            if (!phi_direct()) load_H();
            add_GtDG([&](Row& r) { return rho * r.rp; });
            cholesky();
            if (chol_broke) break;
            double rpa = 0.0;
            if (fresh) {                      // r evaluated exactly once, then carried as r += G dz
                apply_G(z, [&](Row& r, double gz) {
                    r.gd = gz - r.h;
                    rpa = fmax(rpa, r.rp * fabs(r.gd));
                });
                fresh = false;
            } else {
                for_rows([&](int, int, Row& r) { if (fin(r)) rpa = fmax(rpa, r.rp * fabs(r.gd)); });
            }
            rpa = w.maxv(rpa);
            bool retry = false, stagnated = false;
            double rpa_prev = 1e300, lmax = 0.0;
            MPCQP_NOUNROLL
            for (int round = 0; round < 8; ++round) {
                if (!(rpa == rpa)) break;
                // r_A that stops shrinking above its floor: the rows of A cannot all be tight
                if (round >= 2 && rpa > 1e-13 * nh && rpa >= 0.25 * rpa_prev) { stagnated = true; break; }
                rpa_prev = rpa;
                // r_A above the floor: the step needs G_A'l^; at the floor: the test needs G_A'l
                const bool last = rpa <= 1e-13 * nh && !retry;
                if (fold_H) Hz_structured(false);
                apply_Gt([&](Row& r) { return last ? r.pp : r.rp * fma(rho, r.gd, r.pp); });
                double rdn2, ndd2;
                if (fold_H) {
                    dual_residual(nullptr, rdn2, ndd2);
                } else {
                    MPCQP_TIC();
                    dual_residual(m.Hpk + (size_t)b * d.npk, rdn2, ndd2);
                    MPCQP_TOC(2);
                }
                if (!(rdn2 == rdn2)) break;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_DEBUG_POLISH)
                if (w.lane == 0) printf("  polish b=%d fact %d round %d rpa %.3e (nh %.2e) rdn %.3e ndd %.3e last %d\n", b, fact, round, rpa, nh, rdn2, ndd2, (int)last);
#endif
                if (last) {
                    if (rdn2 <= 1e-14 * ndd2) {
                        // exact r once more: r_A at its floor, multipliers >= 0 on A, every other row feasible
                        double lmin = 0.0, rchk = 0.0;
                        bool infeas = false;
                        lmax = 0.0;
                        apply_G(z, [&](Row& r, double gz) {
                            r.gd = gz - r.h;
                            if (r.rp != 0.0) {
                                rchk = fmax(rchk, fabs(r.gd));
                                lmax = fmax(lmax, fabs(r.pp));
                                lmin = fmin(lmin, r.pp);
                            } else {
                                infeas = infeas || (r.gd > 1e-11 * nh);
                            }
                        });
                        lmax = w.maxv(lmax);
                        ok = w.maxv(rchk) <= 1e-12 * nh && !w.any(infeas) && w.minv(lmin) >= -1e-12 * (1.0 + lmax);
                        break;
                    }
                    retry = true;             // r_d not there yet: one more step (G_A'l^ is needed for it)
                    continue;
                }
                retry = false;
                for (int k = w.lane; k < n; k += WAVE) gt[k] = -rd[k];
                w.sync();
                solve_into_dz();
                for (int k = w.lane; k < n; k += WAVE) z[k] += dz[k];
                w.sync();
                rpa = 0.0;
                lmax = 0.0;
                apply_G(dz, [&](Row& r, double g) {
                    if (r.rp != 0.0) r.pp = fma(rho, r.gd + g, r.pp);
                    r.gd += g;
                    rpa = fmax(rpa, r.rp * fabs(r.gd));
                    lmax = fmax(lmax, fabs(r.pp));
                });
                rpa = w.maxv(rpa);
                lmax = w.maxv(lmax);
            }
            if (ok || fact + 1 >= MPCQP_POLISH_FACTS) break;
            // Next working set.  Rows of A that cannot all be tight (r_A stagnated: its limit is the
            // least-squares residual of G_A z = h_A): the rows on the slack side of it are not active --
            // drop them and start over from the interior-point multipliers (the multipliers of this
            // attempt have drifted by rho r_A per round).  Equality-constrained problem solved but not
            // the QP: the multiplier rule of the method, l_i + rho r_i > 0 (for a row of A, l holds it
            // already), with the acceptance thresholds as dead bands.
            bool chg = false;
            if (stagnated) {
                for_rows([&](int, int, Row& r) {
                    if (!fin(r) || r.rp == 0.0) return;
                    if (r.gd < -1e-13 * nh) { r.rp = 0.0; r.pp = 0.0; chg = true; }
                    else r.pp = r.lam;
                });
                if constexpr (ZREG) { if (w.lane < n) z[w.lane] = zkeep; }
                else
                    for (int k = w.lane; k < n; k += WAVE) z[k] = zb[k];
                w.sync();
                fresh = true;
            } else {
                for_rows([&](int, int, Row& r) {
                    if (!fin(r)) return;
                    if (r.rp != 0.0) {
                        if (r.pp < -1e-12 * (1.0 + lmax)) { r.rp = 0.0; r.pp = 0.0; chg = true; }
                    } else if (r.gd > 1e-11 * nh) {
                        r.rp = 1.0; chg = true;
                    }
                });
            }
            const bool changed = w.any(chg);
            if (!changed) break;              // converged (ok) or failed without a new working set
        }
        if (!ok) {
            if constexpr (ZREG) { if (w.lane < n) z[w.lane] = zkeep; }
            else
                for (int k = w.lane; k < n; k += WAVE) z[k] = zb[k];
            w.sync();
        }
        return ok;
    }

    // Row part of one Newton step of the dual-regularised system.  With D = lam/s, w = 1/(1+δD) and
    // D~ = D w, everything a row needs is one reciprocal  wi = w/s = 1/(s + δ lam):
    //   D~ = lam wi,   dl = w (D (rp + G dz) - rc/s) = wi (lam a - rc),   a = rp + G dz
    //   ds = -(rc + s dl)/lam = -wi (s a + δ rc)            (no division by lam, no cancellation)
    // The rows satisfy  s dl + lam ds = -rc  and  rp + G dz + ds = δ dl.
    MPCQP_HD double row_wi(const Row& r) const { return rcp(fma(delta, r.lam, r.s)); }
    // The same quantity five times per iteration (D~, two right-hand sides, two row steps): with compile-time dims it
    // is formed once, when Phi is assembled (row_wi_fresh), and kept in a register until the iterate moves.
    MPCQP_HD double row_wi_fresh(Row& r) const { const double wi = row_wi(r); if (r.wic) *r.wic = wi; return wi; }
    MPCQP_HD double row_wi_cached(const Row& r) const { return r.wic ? *r.wic : row_wi(r); }
    MPCQP_HD void row_step(const Row& r, double rc, double& ds, double& dl) const {
        const double wi = row_wi_cached(r), a = r.rp + r.gd;
        dl = wi * fma(r.lam, a, -rc);
        ds = -wi * fma(r.s, a, delta * rc);
    }

    // (H + G'D~G) dz = -rd + G'(w rc/s - D~ rp); then gd = G dz.  rc(Row&) given by functor.
    // rowfn(Row&) runs on every finite row right after its gd = (G dz)[row] is known -- the row step of the caller in the
    // same pass over the rows as G dz (one traversal of the row slots less per Newton solve).
    template <class Fn, class RowFn>
    MPCQP_HD void newton(Fn rc, RowFn rowfn) {
        apply_Gt([&](Row& r) {
            return row_wi_cached(r) * (rc(r) - r.lam * r.rp);
        });
        for (int k = w.lane; k < d.nZ; k += WAVE) gt[k] -= rd[k];
        w.sync();
        if (!(MPCQP_ABLATE & 4)) solve_into_dz();
        apply_G(dz, [&](Row& r, double g) { r.gd = g; rowfn(r); });
    }

    MPCQP_HD int run(const StepIO& io, int& iters_out) {
        const int n = d.nZ;
        // warm start: Z̃s = [Z̃prev[nu+1:nΔU]; 0; ϵprev]       transcription.jl:1001-1004
        const double* Zg = io.Z + (size_t)b * n;
        const bool cold = d.flags & 2u;
        for (int k = w.lane; k < n; k += WAVE) {
            double v = 0.0;
            if (!cold) {
                if (k < d.nDU - d.nu) v = Zg[k + d.nu];
                else if (k >= d.nDU) v = Zg[k];
            }
            z[k] = v;
        }
        w.sync();
        if (mact != 0) init_fold_H();
        if (mact == 0) {
            // no finite row at all: Z̃ = -H̃^{-1} q̃ (what ExplicitMPC computes, explicitmpc.jl:216)
            load_H();
            cholesky();
            for (int k = w.lane; k < n; k += WAVE) gt[k] = -q[k];
            w.sync();
            solve_into_dz();
            for (int k = w.lane; k < n; k += WAVE) z[k] = dz[k];
            w.sync();
            iters_out = 0;
            if (io.audit && w.lane == 0) {      // closed form: exact up to the Cholesky solve
                double* au = io.audit + (size_t)b * 4;
                au[0] = 0.0; au[1] = 0.0; au[2] = 0.0; au[3] = 1.0;
            }
            return ST_OPTIMAL;
        }
        // warm start kept in a register for the error path (nZ <= 64: one entry per lane; larger
        // problems read it again from the caller's Z̃, which is only overwritten after the solve)
        const double zws = (w.lane < n) ? z[w.lane] : 0.0;
        double mu, rpn, rdn, ndd;
        // ---- starting point (no factorisation): slacks of the warm start pushed to >= 1,
        //      multipliers on the central path of mu = 10:  s = max(h - G z, 1), lam = 10 / s.
        //      Measured against an affine-step start (one extra factorisation) on the BASELINE
        //      configs: 1-2 fewer factorisations per solve and a shorter tail.
        //      With MPCQP_FLAG_WARM_DUAL and the multipliers of the previous period at hand (closed
        //      loop), the start is placed on the central path of mu0 = 1e-3 around them instead:
        const double* lam_prev = ((d.flags & 8u) && !cold) ? io.lam_prev : nullptr;
        if (lam_prev) {
            const double* lp = lam_prev + (size_t)b * d.nrows();
            apply_G(z, [&](Row& r, double gz) { r.s = r.h - gz; });
            for_rows([&](int g, int k, Row& r) {
                if (!fin(r)) return;
                // s = max(h - G z, 1e-3), lam = max(lam_prev, mu0/s), s = max(s, mu0/lam), mu0 = 1e-3.
                // (Measured on C3 closed loops: -2.3 iterations per period once the loop has settled,
                // +4 in the period after a large estimate correction; clamping the multipliers to a
                // band around the central path made the tail worse.)
                const double mu0 = 1e-3;
                double si = fmax(r.s, 1e-3);
                const double li = fmax(lp[d.rowoff(g) + k], r.wt * mu0 * rcp(si));
                si = fmax(si, r.wt * mu0 * rcp(li));
                r.s = si; r.lam = li;
            });
        } else {
            apply_G(z, [&](Row& r, double gz) {
                r.s = fmax(r.h - gz, 1.0);
                r.lam = 10.0 * r.wt * rcp(r.s);
            });
        }
        int status = ST_ITERATION_LIMIT;
        int it = 0;
        // Residuals are evaluated exactly (G z, G'lam, H̃ z) at the first iterate and whenever the
        // recursively updated ones claim convergence; in between they follow the Newton identities
        //   r_p <- (1 - alpha) r_p + alpha δ dlam,      r_d <- (1 - alpha) r_d
        // (block rows G dz + ds - δ dlam = -r_p and H dz + G'dlam = -r_d), which saves one G z, one
        // G'lam and one H̃ z per iteration.  A verification that fails simply continues from the
        // exact values.
        bool exact = true, verified = false;
        double musum_c = 0.0, rpmax_c = 0.0, rdscale_c = 1.0;    // carried from the update pass
        double step_c = 1e300, zabs_c = 0.0;                     // |alpha dU_k|, |dU_k| of this lane's entry
        double rd_exact_prev = 1e300, scale_since_exact = 1.0;   // stall detection of the exact dual residual
        bool rd_stalled = false, rp_stalled = false;
        double polmu_next = MPCQP_POLISH_MU;
        int npolish = 0;
        bool polished = false;
        double rpn_last = 1e300;
        while (it < (MPCQP_FIXED_ITERS ? MPCQP_FIXED_ITERS : d.max_iter)) {
            // A pivot below its threshold in the last factorisation means Phi = H̃ + G'D~G left
            // float64's range (rows held at D~ = 1/δ stack up to 1e14 on the diagonal of long
            // horizons).  The guarded factor kept that step finite (and the step-length rules kept
            // s, lam positive); from here on the dual regularisation is 100 times larger (at most
            // twice), which caps D~ lower and biases nothing -- δ multiplies the multiplier step,
            // which vanishes at the optimum -- and the residuals are re-evaluated exactly.  (Not met
            // on the BASELINE configs; about one family in 20 at nZ~ ~ 100.)
            if (chol_broke && delta < 1e-8 && !MPCQP_FIXED_ITERS) {
                delta *= 100.0;
                exact = true;
                chol_broke = false;
            }
            MPCQP_MTIC();
            if (exact) {
                residuals(mu, rpn, rdn, ndd);      // also stages H̃ in Phi
                exact = false;
                verified = true;
                // The dual residual has hit the floor of the float64 normal equations when an exact
                // evaluation finds it where the previous exact evaluation left it although the
                // steps in between were (nearly) full: rows held at D = 1/δ make Phi's condition
                // number ~1e13 and r_d stops near 1e-16 * 1e12 * |dz| however long one iterates.
                rd_stalled = rdn >= 0.5 * rd_exact_prev && scale_since_exact <= 0.1;
                rd_exact_prev = rdn;
                scale_since_exact = 1.0;
            } else {
                // sum s lam and max |r_p| were accumulated by the update pass of the previous
                // iteration; max |r_d| scales with the dual residual itself
                mu = w.sum(musum_c) / wsum;
                rpn = w.maxv(rpmax_c);
                rdn *= rdscale_c;
                scale_since_exact *= rdscale_c;
                verified = false;
                // a primal residual that no longer follows (1 - alpha) -- it sits on alpha δ dlam, a
                // few 1e-10 nh on rows with 1e6-size multipliers -- has stalled too, once below 1e-9 nh
                // (degenerate vertices turn a violation of 1e-8 into an error of 3e-4 in z: measured on
                // C3 instances 36072 of seed 0 and 70686 of seed 3, both fixed by these two targets)
                rp_stalled = rpn >= 0.5 * rpn_last && rdscale_c <= 0.1 && rpn <= 1e-9 * nh;
                rpn_last = rpn;
            }
#if !MPCQP_FIXED_ITERS
            if (!(mu == mu) || !(rdn == rdn) || !(rpn == rpn)) { status = ST_ERROR; break; }
            // Converged: gap and residuals below their targets AND the last Newton step no longer moves
            // the inputs, alpha |dU|_inf <= 1e-6 max(1, |dU|_inf).  (Residual targets alone leave 1e-6-size
            // errors in z on instances with 1e5-size multipliers or weakly active rows.  Measured on
            // 4084 certified C3 optima: worst dU error 6.7e-6 -> 3.1e-8 for +0.26 iterations.)
            // The dual residual is measured against a gradient scale that reaches 1e5 (soft rows):
            // its target is res_tol; the primal one is 10 res_tol (degenerate vertices, see above).
            // A dual residual stalled at its floor counts as converged (the step criterion is what
            // vouches for z then).
            if (mu <= d.gap_tol && (rdn <= d.res_tol * ndd || (verified && rd_stalled)) &&
                (rpn <= 10.0 * d.res_tol * nh || rp_stalled) &&
                w.maxv(step_c) <= 1e-6 * fmax(1.0, w.maxv(zabs_c))) {
                if (verified) { status = ST_OPTIMAL; break; }
                exact = true;                      // re-evaluate exactly at the same iterate
                continue;
            }
            // Active-set polish once the gap is small: first at mu <= 1e-6, again after every further
            // factor 100 if it was not accepted (wrong active set: weakly active or degenerate rows).
            if (mu <= polmu_next && rpn <= MPCQP_POLISH_RP * nh && npolish < MPCQP_POLISH_BUDGET && !(d.flags & 16u)) {
                polmu_next = 1e-2 * mu;
                MPCQP_TICK(tic14_);
                const bool pol_ok = polish(npolish);
                MPCQP_TOCK(14, tic14_);
                if (pol_ok) { polished = true; status = ST_OPTIMAL; break; }
                exact = true;                      // Phi, rd, gt were used: start over from exact residuals
                continue;
            }
#endif
            MPCQP_MTOC("looptop");
            if (!verified && !phi_direct()) load_H();
            add_GtDG([&](Row& r) {
                return r.lam * row_wi_fresh(r);             // D~ = D / (1 + δ D)
            });
            if (!(MPCQP_ABLATE & 2)) cholesky();
#if (MPCQP_CHOL_DIAG || MPCQP_CHOL_LDL) && !MPCQP_FIXED_ITERS && defined(__HIP_DEVICE_COMPILE__)
            // (chol_static's floored pivots leave a finite but meaningless factor: no step is taken with it -- the loop top
            //  raises the dual regularisation and re-evaluates the residuals; with the regularisation at its cap the solve
            //  has failed.  The runtime-dimension factorisation freezes the coordinate instead and goes on as before.)
            if constexpr (one_row_per_lane<DM>()) {
                if (chol_broke) {
                    if (delta < 1e-8) continue;
                    status = ST_ERROR;
                    break;
                }
            }
#endif
            // Two Newton solves with the same factor, one pass of the loop each (one copy of the
            // code): pass 0 the predictor, rc = s lam; pass 1 the corrector,
            // rc = s lam + ds_aff dl_aff - sigma mu (sigma = (mu_aff/mu)^3 from the predictor).
            double amin = 1.0, smu = 0.0, tmax = 1.0;
            MPCQP_NOUNROLL
            for (int pass = 0; pass < 2; ++pass) {
                const double cpp = pass ? 1.0 : 0.0;
                double ppsum = 0.0;
                tmax = pass ? 1e-300 : 1.0;       // 1 / (largest step that keeps s, lam >= 0), capped at 1 for the predictor
                newton([&](Row& r) { return fma(cpp, r.pp, fma(r.s, r.lam, -r.wt * smu)); },
                [&](Row& r) {
                    if (MPCQP_ABLATE & 32) return;
                    double ds, dl;
                    row_step(r, fma(cpp, r.pp, fma(r.s, r.lam, -r.wt * smu)), ds, dl);
                    // step to the boundary as 1 / max(-ds/s, -dl/lam): no compare, no select; raw reciprocals
                    tmax = fmx(tmax, -ds * rcp_fast(r.s));
                    tmax = fmx(tmax, -dl * rcp_fast(r.lam));
                    // predictor: pp <- ds dl.  corrector: the step is kept in the row (pp <- ds,
                    // gd <- dl): the update below needs nothing else, since the primal residual
                    // follows r_p <- (1 - alpha) r_p + alpha δ dl.
                    r.pp = pass ? ds : ds * dl;
                    r.gd = pass ? dl : r.gd;
                    ppsum += ds * dl;
                });
                if (pass == 0) {
                    const double aaff = rcp(w.maxv(tmax));
                    // mu after the affine step: sum (s + a ds)(lam + a dl) = sum s lam (1 - a) + a^2 sum ds dl,
                    // because s dl + lam ds = -s lam on every row of the predictor
                    const double muaff = (1.0 - aaff) * mu + aaff * aaff * w.sum(ppsum) / wsum;
                    double sig = muaff / mu;
                    sig = sig * sig * sig;
                    smu = sig * mu;
                }
            }
            // Fraction to the boundary: 0.9999 when the iterate it leads to stays in the wide
            // neighbourhood min_i s_i lam_i >= 0.01 mu, otherwise 0.99.  (An unguarded 0.999 jams
            // about one instance in 20000; with the guard no instance of 65536 needs more
            // iterations than with 0.99 throughout and the mean drops by about 0.9.)
            MPCQP_TICK(tic13_);
            amin = rcp(w.maxv(tmax));
            const double ahi = fmin(1.0, 0.9999 * amin);
            double pmin = 1e300, psum = 0.0;
            if (!(MPCQP_ABLATE & 32))
            for_rows([&](int, int, Row& r) {
                if (!fin(r)) return;
                const double p = (r.s + ahi * r.pp) * (r.lam + ahi * r.gd);
                pmin = fmn(pmin, r.wt == 1.0 ? p : p * rcp_fast(r.wt));           // per copy of a merged row
                psum += p;
            });
            pmin = w.minv(pmin);
            psum = w.sum(psum);
            const double alpha = (pmin * wsum >= 0.01 * psum) ? ahi : fmin(1.0, 0.99 * amin);
            musum_c = 0.0; rpmax_c = 0.0; rdscale_c = 1.0 - alpha;
            if (!(MPCQP_ABLATE & 32))
            for_rows([&](int, int, Row& r) {
                if (!fin(r)) return;
                r.s += alpha * r.pp;
                r.lam += alpha * r.gd;
                r.rp = fma(alpha, delta * r.gd - r.rp, r.rp);
                musum_c += r.s * r.lam;
                rpmax_c = fmx_abs(rpmax_c, r.rp);
            });
            // (a step cut short by the boundary, alpha < 1/2, says nothing about convergence: only a nearly full Newton step
            //  that no longer moves the inputs does -- instance 99 of shape 8,2,2,60,40 passed the test on a blocked step,
            //  4.9e-4 from the optimum adjudicated in 60-digit arithmetic, tests/golden/hp_optima.json)
            step_c = alpha >= 0.5 ? 0.0 : 1e300; zabs_c = 0.0;
            for (int k = w.lane; k < n; k += WAVE) {
                const double st = alpha * dz[k];
                if (k < d.nDU) { step_c = fmx_abs(step_c, st); zabs_c = fmx_abs(zabs_c, z[k]); }
                z[k] += st;
                rd[k] *= (1.0 - alpha);
            }
            w.sync();
            MPCQP_TOCK(13, tic13_);
            ++it;
        }
        // never primal-feasible (or NaN) => the reference's error branch (execute.jl:484-489)
        // (MPCQP_FLAG_KEEP_ITERATE: diagnostics -- the iterate after exactly max_iter iterations is what the caller wants)
        if (status == ST_ITERATION_LIMIT && !(rpn <= 1e-6 * nh) && !(d.flags & 32u)) status = ST_ERROR;
        if (status == ST_ERROR) {
            if (w.lane < n) z[w.lane] = zws;          // mpc.Z̃ .= Z̃s   execute.jl:499-500
            for (int k = w.lane + WAVE; k < n; k += WAVE)
                z[k] = cold ? 0.0 : (k < d.nDU - d.nu) ? Zg[k + d.nu] : (k >= d.nDU) ? Zg[k] : 0.0;
            w.sync();
        }
        if ((d.flags & 8u) && io.lam_out) {           // multipliers for the next period's start
            double* lo = io.lam_out + (size_t)b * d.nrows();
            const bool good = status != ST_ERROR;
            for_rows([&](int g, int k, Row& r) { lo[d.rowoff(g) + k] = (good && fin(r)) ? (polished ? fmax(r.pp, 0.0) : r.lam) : 0.0; });
        }
        iters_out = it + npolish;      // factorisations: interior-point iterations + polish attempts
        if (io.audit && w.lane == 0) {      // what the convergence test saw last (callers can audit an OPTIMAL)
            double* au = io.audit + (size_t)b * 4;
            au[0] = mu; au[1] = rdn / ndd; au[2] = rpn / nh; au[3] = polished ? 1.0 : 0.0;
        }
        return status;
    }

    static constexpr int ST_OPTIMAL = 0, ST_ITERATION_LIMIT = 1, ST_ERROR = 2;
};

template <class W, class DM>
MPCQP_HD void step_body(W& w, const DM& d, const Model& m, const StepIO& io, int b, double* sm) {
    MPCQP_SETPRIO(4, 1);
#ifdef MPCQP_TEAM_PINGTEST      // (measurement only: the cost of a job hand-off -- that many empty jobs in front of the step)
    if constexpr (W::NTEAM > 1) {
        for (int i_ = 0; i_ < MPCQP_TEAM_PINGTEST; ++i_) { w.post(99); w.join(); }
    }
#endif
    const long long t_in_ = Step<W, DM>::clock64_();
    Qp<W, DM> qp(w, d, m, b, sm);
    qp.load_tables();
    const long long t_tab_ = Step<W, DM>::clock64_();
    // x̂0 of this period into LDS; with kf_y0m the SteadyKalmanFilter correction first
    // (correct_estimate_obsv!, src/estimator/kalman.jl:284-295; same arithmetic order as kf_correct_lane)
    {
        const int nx = d.nxh, ny = d.ny, nd = d.nd;
        double* xh = sm + qp.c.xh;
        for (int i = w.lane; i < nx; i += WAVE) xh[i] = io.xhat0[(size_t)b * nx + i];
        w.sync();
        if (io.kf_y0m) {
            const double* Cm = m.C + (size_t)b * ny * nx;
            const double* K = io.kf_K + (size_t)b * io.kf_nym * nx;
            double acc[4];                                     // rows i = lane + 64 q (nx̂ <= 256)
            int nq = 0;
            for (int i = w.lane; i < nx; i += WAVE, ++nq) {
                double a_ = xh[i];
                for (int mm = 0; mm < io.kf_nym; ++mm) {
                    const int a = io.kf_iym[mm];
                    double v = io.kf_y0m[(size_t)b * io.kf_nym + mm];
                    for (int k = 0; k < nx; ++k) v -= Cm[a + ny * k] * xh[k];
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
    }
    Step<W, DM> st(qp);
    const long long t_b0_ = Step<W, DM>::clock64_();
    st.build(io);
    const long long t_b1_ = Step<W, DM>::clock64_();
    if ((d.flags & 4u) && io.q_keep) {
        for (int k = w.lane; k < d.nZ; k += WAVE) io.q_keep[(size_t)b * d.nZ + k] = st.q[k];
        for (int r = w.lane; r < d.nY; r += WAVE) io.F_keep[(size_t)b * d.nY + r] = st.F[r];
    }
    if (io.Yhat0)      // park F in the output buffer: its LDS copy is scratch from here on
        for (int r = w.lane; r < d.nY; r += WAVE) io.Yhat0[(size_t)b * d.nY + r] = st.F[r];
    int iters = 0;
    const long long t_run0 = Step<W, DM>::clock64_();
    const int status = st.run(io, iters);
    if (io.prof) {
        st.prof_[15] = (double)(Step<W, DM>::clock64_() - t_run0);
#ifdef MPCQP_PROFILE_SETUP      // (developer switch: the set-up of the step in the slots of the Newton sub-phases)
        st.prof_[8] = (double)(t_tab_ - t_in_);       // table staging
        st.prof_[9] = (double)(t_b0_ - t_tab_);       // x̂0 (+ Kalman correction), Step construction
        st.prof_[10] = (double)(t_b1_ - t_b0_);       // build(): F, q̃, bounds, rows
        st.prof_[11] = (double)(t_run0 - t_b1_);      // between build() and run()
#endif
        if (w.lane == 0)
            for (int i = 0; i < 16; ++i) io.prof[(size_t)b * 16 + i] = st.prof_[i];
    }
    // outputs: Z̃, u0 = Z̃[1:nu] + lastu0 (getinput!), Ŷ0 = Ẽ Z̃ + F (predict!)
    for (int k = w.lane; k < d.nZ; k += WAVE) io.Z[(size_t)b * d.nZ + k] = st.z[k];
    for (int k = w.lane; k < d.nu; k += WAVE)
        io.u0[(size_t)b * d.nu + k] = st.z[k] + io.lastu0[(size_t)b * d.nu + k];
    if (io.Yhat0) {
        double* tY = sm + st.c.tA[P_Y];
        qp.E_apply(st.z, tY);
        for (int r = w.lane; r < d.nY; r += WAVE) io.Yhat0[(size_t)b * d.nY + r] += tY[r];      // same lane parked F[r]
    }
    if (io.kf_predict) {
        // updatestate! (predict_estimate_obsv!, kalman.jl:298-309) with the input just computed:
        // x̂0 <- Â x̂0 + B̂u u0 + B̂d d0 + (f̂op - x̂op)
        const int nx = d.nxh, nu = d.nu, nd = d.nd;
        const double* xh = sm + st.c.xh;
        const double* A = m.Ahat + (size_t)b * nx * nx;
        for (int i = w.lane; i < nx; i += WAVE) {
            double acc = m.dop ? m.dop[(size_t)b * nx + i] : 0.0;
            for (int k = 0; k < nx; ++k) acc += A[i + nx * k] * xh[k];
            for (int cc = 0; cc < nu; ++cc)
                acc += m.Bu[(size_t)b * nx * nu + i + nx * cc] * (st.z[cc] + io.lastu0[(size_t)b * nu + cc]);
            for (int e = 0; e < nd; ++e) acc += m.Bd[(size_t)b * nx * nd + i + nx * e] * io.d0[(size_t)b * nd + e];
            io.xhat0_out[(size_t)b * nx + i] = acc;
        }
    }
    if (w.lane == 0) {
        io.status[b] = status;
        if (io.iters) io.iters[b] = iters;
    }
}


#if defined(__HIP_DEVICE_COMPILE__)
// Wavefronts 1 .. T-1 of a team (DevWaveT, mpcqp_devwave.h): wait for a job, run this wavefront's share, meet wavefront 0.
template <class W, class DM>
__device__ __forceinline__ void team_helper(W& w, const DM& d, const Model& m, int b, double* sm) {
    Qp<W, DM> qp(w, d, m, b, sm);
    Step<W, DM> st(qp);
    for (;;) {
        w.join();
        const int* mi = reinterpret_cast<const int*>(w.mbox);
        const int job = __builtin_amdgcn_readfirstlane(mi[0]);
        if (job == TJ_EXIT) break;
        // (the lane id laundered per job: otherwise the address arithmetic of EVERY job's share is hoisted in front of this
        //  loop and kept in -- spilled -- registers: 500 spill slots and 87 scratch stores in the helper's prologue, nZ~ = 151)
        w.relane();
        const int a0 = __builtin_amdgcn_readfirstlane(mi[1]), a1 = __builtin_amdgcn_readfirstlane(mi[2]);
        const int a2 = __builtin_amdgcn_readfirstlane(mi[3]), a3 = __builtin_amdgcn_readfirstlane(mi[4]);
        const double sc = w.mbox[3];
        switch (job) {
            case TJ_ETDE:
                if constexpr (DM::is_static)
                    (void)qp.EtDE_share(sm + a0, sm + a1, sc, sm + a2, (a3 & 2) ? m.Hpk + (size_t)b * d.npk : nullptr, (a3 & 1) != 0, (a3 & 4) != 0);
                break;
            case TJ_PANEL: st.template panel_dispatch<1>(a0); break;
            case TJ_EV: qp.E_apply_share(sm + a0, sm + a1); break;
            case TJ_ETW: qp.Et_apply_share(sm + a0, sm + a1, sc, a2); break;
            case TJ_UROWS: st.GtDG_urows_share(); break;
            case TJ_XROWS: st.GtDG_xrows_share(); break;
            case TJ_WROWS: st.GtDG_wrows_share(); break;
            case TJ_PANELROWS: st.template panel_rows_dispatch<0>(a0); break;
            case TJ_LOADH: st.load_H_share(); break;
            case TJ_HZ:          // (two calls: a select between an LDS and a global pointer would make every access a flat one)
                if (a0) st.Hz_rows_share(st.Phi);
                else st.Hz_rows_share(m.Hpk + (size_t)b * d.npk);
                break;
            default: break;
        }
        w.join();
    }
}

#endif

// wavefronts per problem of a specialised kernel: MPCQP_TEAM when given, else by the LDS footprint -- one problem per CU
// (beyond 80 KB): four, one per SIMD; two problems per CU: two; otherwise one wavefront per problem as before
template <class DM>
constexpr int auto_team() {
    if constexpr (!DM::is_static) return 1;
    else {
        if (DM::nZ <= WAVE) return 1;
        const long bytes = 8L * ((long)(DM::Hp + DM::zpad) * DM::sp + DM::npk + 8L * DM::nZ + 3L * DM::nY + 2L * DM::nDU);
        return bytes > 80 * 1024 ? 4 : bytes > 160 * 1024 / 3 ? 2 : 1;
    }
}

// ------------------------------------------------------------------------------------------
// SteadyKalmanFilter steps.  One lane per state of one problem; a problem's nx̂
// lanes sit in one wavefront, so "all lanes read, then all lanes write" needs no barrier on the
// GPU (xin == xhat0 there; the CPU emulator passes a copy as xin).
// `slot` = problem handled by this lane, `i` = state index (inactive when i >= nxh).
// ------------------------------------------------------------------------------------------
MPCQP_HD inline void kf_correct_lane(const Dims& d, const Model& m, const KfParams& kf, int slot, int i,
                                     const double* xin, double* xhat0, const double* y0m, const double* d0) {
    if (slot >= d.B || i >= d.nxh) return;
    const int nx = d.nxh, ny = d.ny, nd = d.nd;
    const double* x = xin + (size_t)slot * nx;
    const double* Cm = m.C + (size_t)slot * ny * nx;            // (ny,nx̂) col-major
    const double* K = kf.Khat + (size_t)slot * kf.nym * nx;     // (nx̂,nym) col-major
    double acc = x[i];
    for (int mm = 0; mm < kf.nym; ++mm) {
        const int a = kf.i_ym[mm];
        double v = y0m[(size_t)slot * kf.nym + mm];             // innovation y0m - Ĉm x̂0 - D̂dm d0
        for (int k = 0; k < nx; ++k) v -= Cm[a + ny * k] * x[k];
        for (int e = 0; e < nd; ++e) v -= m.Dd[(size_t)slot * ny * nd + a + ny * e] * d0[(size_t)slot * nd + e];
        acc += K[i + nx * mm] * v;
    }
    // (all loads of the wave above precede this store in program order)
    xhat0[(size_t)slot * nx + i] = acc;
}

MPCQP_HD inline void kf_predict_lane(const Dims& d, const Model& m, int slot, int i, const double* xin,
                                     double* xhat0, const double* u0, const double* d0) {
    if (slot >= d.B || i >= d.nxh) return;
    const int nx = d.nxh, nu = d.nu, nd = d.nd;
    const double* x = xin + (size_t)slot * nx;
    const double* A = m.Ahat + (size_t)slot * nx * nx;
    double acc = m.dop ? m.dop[(size_t)slot * nx + i] : 0.0;
    for (int k = 0; k < nx; ++k) acc += A[i + nx * k] * x[k];
    for (int c = 0; c < nu; ++c) acc += m.Bu[(size_t)slot * nx * nu + i + nx * c] * u0[(size_t)slot * nu + c];
    for (int e = 0; e < nd; ++e) acc += m.Bd[(size_t)slot * nx * nd + i + nx * e] * d0[(size_t)slot * nd + e];
    xhat0[(size_t)slot * nx + i] = acc;
}

}  // namespace mpcqp
