// mpcqp_types.h -- plain-old-data shared by the HIP kernels and the host library.
//
// Row groups of the inequality system A Z̃ <= b of one LinMPC, in the reference's order
// (init_matconstraint_mpc, src/controller/transcription.jl:686-703) with the variable bounds
// Z̃min/Z̃max (init_boxconstraint_mpc, src/controller/construct.jl:1209-1234) put first:
//   pair 0  box   : -z_k <= -Z̃min_k            |  z_k <= Z̃max_k                (k < nZ)
//   pair 1  U     : -Pu ΔU - C_umin ϵ <= ...     |  Pu ΔU - C_umax ϵ <= ...       (nDU rows: the nb_j
//                   identical rows of a move-blocking interval are merged into their tightest one)
//   pair 2  ΔU    : soft ΔU rows only (hard ones are the box)                    (nDU rows)
//   pair 3  Ŷ     : -E ΔU - C_ymin ϵ <= ...      |  E ΔU - C_ymax ϵ <= ...        (nY rows)
//   pair 4  x̂end : -ex̂ ΔU - c_x̂min ϵ <= ...     |  ex̂ ΔU - c_x̂max ϵ <= ...      (nx̂ rows)
// group id = 2*pair + (0 = min rows, 1 = max rows).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
// always_inline: a real call inside the one-wave kernels makes `this` escape: the solver state (the Step / Qp objects)
// then lives in scratch memory and its LDS pointer members come back from memory as GENERIC pointers, so every LDS access
// turns into flat_load / flat_store.  The backend folds constant displacements into the flat instruction's offset field
// and the hardware selects the aperture from the 64-bit BASE: for the zero blocks in front of the Sigma table (LDS offset
// 0) the base lies below the LDS aperture and the access leaves it -- wrong data or a memory aperture violation (DESIGN 4,
// profiles/r3/outline_rocgdb.txt; tests/test_gpu_parity.py::test_families_near_wave_limit).  The inliner's size
// heuristics had left cholesky() / EtDE_add() out of line in the larger specialisations (nZ~ > 48).
#define MPCQP_HD __host__ __device__ __attribute__((always_inline))
// (investigation of that miscompilation only: -DMPCQP_OUTLINE=1 puts cholesky(), =2 EtDE_add(), =3 both out of line)
#if defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_OUTLINE) && (MPCQP_OUTLINE & 1)
#define MPCQP_HD_CHOL __host__ __device__ __attribute__((noinline))
#else
#define MPCQP_HD_CHOL MPCQP_HD
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(MPCQP_OUTLINE) && (MPCQP_OUTLINE & 2)
#define MPCQP_HD_ETDE __host__ __device__ __attribute__((noinline))
#else
#define MPCQP_HD_ETDE MPCQP_HD
#endif
#define MPCQP_UNROLL _Pragma("unroll")
#define MPCQP_UNROLL4 _Pragma("unroll 4")
#define MPCQP_NOUNROLL _Pragma("nounroll")
#if defined(__HIP_DEVICE_COMPILE__)
// stop the instruction scheduler from interleaving the (fully unrolled) per-slot row updates:
// their temporaries would otherwise be live all at once and cost a wave of occupancy
#define MPCQP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MPCQP_SCHED_FENCE() ((void)0)
#endif
#else
#define MPCQP_SCHED_FENCE() ((void)0)
#define MPCQP_HD
#define MPCQP_HD_CHOL
#define MPCQP_HD_ETDE
#define MPCQP_UNROLL
#define MPCQP_UNROLL4
#define MPCQP_NOUNROLL
#endif

// Revision of the kernel sources / device structs: part of the name of cached on-demand
// specialisations, so that objects built from older sources are never loaded.
#define MPCQP_KERNEL_REV 12       // 12: matrix-core operands of E'DE in registers (MPCQP_ETDE_VREG); 11: on-demand objects compiled with the pragma-unroll threshold lifted (no scratch arrays / flat accesses from eight tile rows on); 10: the row eps >= 0 rides in a Ŷ group (eps_host_group); 9: a blocked step (alpha < 1/2) no longer passes the last-step test; 8: MPCQP_FLAG_KEEP_ITERATE

namespace mpcqp {

// jobs a team's wavefront 0 hands to its helpers (DevWaveT, mpcqp_devwave.h; W::NTEAM == 1: never posted)
enum TeamJob { TJ_EXIT = 0, TJ_ETDE = 1, TJ_PANEL = 2, TJ_EV = 3, TJ_ETW = 4, TJ_UROWS = 5, TJ_WROWS = 6, TJ_XROWS = 7, TJ_PANELROWS = 8, TJ_LOADH = 9, TJ_HZ = 10 };


enum { P_BOX = 0, P_U = 1, P_DU = 2, P_Y = 3, P_X = 4, P_W = 5, NPAIR = 6, NGROUP = 12 };

constexpr int WAVE = 64;          // gfx950 wavefront
constexpr double BIG = 1e300;     // |h| >= BIG  <=>  row absent (bound was +-Inf)

// The row  -eps <= 0  (Z̃min[end] = 0, construct.jl:1216) is formally a box row, and the only one of its group when no hard
// dUmin exists -- a whole row slot (64 lanes of per-row state and algebra in every pass of the interior-point iteration)
// for ONE row.  It is the same inequality as an output-bound row with a zero row of E, softness 1 and bound 0:
// 0'dU - 1 eps <= 0.  So when a Ŷ group exists the row rides there as its row k = nY (the group then has nY + 1 rows and
// the box-lower group only exists for hard dUmin rows); eps_host_group says where.  -1: it stays in the box group.
MPCQP_HD constexpr int eps_host_group(uint32_t gmask, int neps) {
    return !neps ? -1 : ((gmask >> (2 * P_Y + 1)) & 1u) ? 2 * P_Y + 1 : ((gmask >> (2 * P_Y)) & 1u) ? 2 * P_Y : -1;
}

// Runtime dimensions (the generic kernels read these; the specialised kernels take the same
// values as template constants, see StaticDims in mpcqp_bodies.h).
struct Dims {
    int B, nxh, nu, ny, nd, Hp, Hc, neps;
    int nZ, nDU, nU, nY, nD;
    int nw, nW;              // custom linear constraints: rows per step, nw (Hp+1) in total (0: none)
    int npk;                 // pk_size(nZ): packed lower triangle, rows padded in groups of four
    uint32_t gmask;          // bit g set <=> row group g may hold finite rows (handle level)
    int rowoff_[NGROUP + 1]; // first row of group g in the per-problem row arrays (inactive: empty)
    int cnt_[NPAIR];         // rows per pair: nZ, nDU, nDU, nY (+ 1 with the eps row, eps_host_group), nxh, nW
    int default_nb;          // 1 iff nb = [1,..,1,Hp-Hc+1]
    int dense_w;             // 1 iff a dense M_Hp or L_Hp is set: the step runs on an on-demand variant with the dense products or on the runtime-dimension kernel
    int max_iter;
    double gap_tol, res_tol, dual_reg;
    uint32_t flags;
    static constexpr bool is_static = false;
    MPCQP_HD int cnt(int p) const { return cnt_[p]; }
    // length per problem of the caller's arrays of pair p (bounds, softness): cnt(p) without the hosted eps row
    MPCQP_HD int len(int p) const { return p == P_Y ? nY : cnt_[p]; }
    MPCQP_HD int eps_host() const { return eps_host_group(gmask, neps); }
    MPCQP_HD int rowoff(int g) const { return rowoff_[g]; }
    MPCQP_HD int nrows() const { return rowoff_[NGROUP]; }
};

// device-resident, per-handle data (all problem-major; "col-major inside a problem" where the
// ABI exposes it, internal tables in whatever order the kernels like)
struct Model {
    // inputs of set_model (ABI layout)
    const double *Ahat, *Bu, *C, *Bd, *Dd, *dop;     // dop = f̂op - x̂op, may be null
    // K1 outputs
    double* Stab;    // [B][Hp][ny][nu]   Σ_m = Ĉ S(m) B̂u
    double* Ktab;    // [B][nxh][nY]      K, column-major (nY fastest)
    double* Bvec;    // [B][nY]
    double* Gdtab;   // [B][Hp][ny][nd]   Ĉ Â^m B̂d                     (nd > 0)
    double* exT;     // [B][Hc][nxh][nu]  ex̂ block j = S(Hp-j_j-1) B̂u   (terminal rows)
    double* kxT;     // [B][nxh][nxh]     Â^Hp, column-major              (terminal rows)
    double* bxv;     // [B][nxh]          S(Hp-1)(f̂op - x̂op)
    double* Xdtab;   // [B][Hp][nxh][nd]  Â^m B̂d                        (terminal rows, nd > 0)
    // K2 output
    double* Hpk;     // [B][npk]          H̃, packed lower triangle row-major
    // weights
    const double *Mdiag, *Ndiag, *Ldiag, *Cwt;
    const double* Mblk;   // optional [B][Hp][ny][ny]: block-diagonal M_Hp (symmetric blocks), replaces Mdiag
    // optional dense (symmetric) weights, column-major per problem; they replace the diagonals (runtime-dimension kernels and
    // on-demand specialisations built with MPCQP_SPEC_DENSE)
    const double* Mfull;  // [B][nY][nY]    M_Hp coupling different prediction steps
    const double* Ndense; // [B][nDU][nDU]  N_Hc
    const double* Ldense; // [B][nU][nU]    L_Hp
    // bounds + softness (null = group absent / default softness)
    const double *U0min, *U0max, *DUmin, *DUmax, *Y0min, *Y0max, *x0min, *x0max;
    const double *C_umin, *C_umax, *C_dumin, *C_dumax, *C_ymin, *C_ymax, *c_x0min, *c_x0max;
    // custom linear constraints (relaxW, construct.jl:1086-1160): W = Wy ŷe + Wu ue + Wd d̂e + Wr r̂e
    const double *Wy, *Wu, *Wd, *Wr;   // [B][ny|nu|nd|ny][nw] (ABI (nw,·,B)); Wd, Wr may be null
    const double* w_op;                // [B][nw] operating-point part of W, may be null
    const double* ry_now;              // [B][ny] current set point ry(k) - yop for the Wr term of step k (null: first block of R̂y)
    const double *Wmin, *Wmax, *C_wmin, *C_wmax;   // [B][nW]
    // horizon tables
    const int* jl;   // [Hc+1] block starts j_l (move_blocking, construct.jl:597-660)
    const int* blk;  // [Hp]   index of the block that holds step t
};

// steady-state Kalman filter of the batch (estimator/kalman.jl:284-309)
struct KfParams {
    const double* Khat;   // [B][nym][nxh]  (ABI (nx̂,nym,B): nx̂ fastest)
    const int* i_ym;      // [nym]
    int nym;
};

struct StepIO {
    const double *xhat0, *lastu0, *Ry, *Ru, *d0, *Dhat0;
    double *Z, *u0, *Yhat0;
    int32_t *status, *iters;
    double *q_keep, *F_keep;   // optional (MPCQP_FLAG_KEEP_QP)
    const double* lam_prev;    // optional [B][nrows]: multipliers of the previous period (MPCQP_FLAG_WARM_DUAL)
    double* lam_out;           // optional [B][nrows]: multipliers of this period
    double *prof;              // optional [B][16] per-phase cycle counts (-DMPCQP_PROFILE builds only)
    double *audit;             // optional [B][4]: final gap mu, dual residual / its scale, primal residual / its scale, 1 if the
                               // returned point is an accepted active-set polish (KKT conditions of the QP checked)
    // optional: the SteadyKalmanFilter steps on both sides of moveinput! inside the same launch
    // (mpcqp_loop_device): kf_y0m != null => preparestate! first, x̂0 += K̂ (y0m - Ĉm x̂0 - D̂dm d0);
    // kf_predict != 0 => updatestate! last, x̂0 <- Â x̂0 + B̂u u0 + B̂d d0 + (f̂op - x̂op); both write xhat0_out
    const double* kf_K;        // [B][nym][nxh]
    const int* kf_iym;         // [nym]
    const double* kf_y0m;      // [B][nym]
    double* xhat0_out;         // [B][nxh]  (may alias xhat0)
    int kf_nym, kf_predict;
};

// Packed lower triangle, row-major.  Rows are grouped in fours; every row of group g = i/4 is padded
// to 4(g+1) entries, so that
//   * a row starts on a 32-byte boundary and every aligned 4-column chunk of it is two
//     ds_read_b128 (a 16-byte DS access off its natural alignment is replayed at ~64 cycles per
//     wave instruction, cdna_hip_programming.md Guideline 17);
//   * the four rows of a group share one stride, 4(g+1);
//   * the entries (i, j) with j > i inside the last chunk of row i exist and hold ZERO.  The factor
//     keeps a zero in the diagonal slot as well (1/L[i][i] lives in a register of lane i), so the
//     triangular sweeps read whole chunks without masking the diagonal block.
// Row i starts at 4(g+1)(2g + i%4); n rows take pk(n, 0) doubles.
MPCQP_HD constexpr int pk(int i, int j) { return 4 * ((i >> 2) + 1) * (2 * (i >> 2) + (i & 3)) + j; }   // i >= j
MPCQP_HD constexpr int pk_size(int n) { return pk(n, 0); }

}  // namespace mpcqp
