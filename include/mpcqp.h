/*
 * mpcqp.h -- C-ABI of the MI355X-native batched LinMPC step (condense + QP solve).
 *
 * Drop-in boundary for the LinMPC `moveinput!` hot path of JuliaControl/ModelPredictiveControl.jl
 * v2.11.0 (citations are relative to /root/reference).  The reference has no FFI of its own for
 * this path: the seam is the `optim::JuMP.GenericModel` field (src/controller/linmpc.jl:14,245)
 * driven by five JuMP calls -- set_normalized_rhs (src/controller/transcription.jl:845),
 * set_objective_coefficient (src/controller/execute.jl:513), set_start_value
 * (src/controller/transcription.jl:1005), optimize! (src/controller/execute.jl:472) and value
 * (:502).  This library replaces, for a batch of B independent controllers of identical
 * dimensions:
 *
 *     mpcqp_set_model    <->  init_predmat   src/controller/transcription.jl:115-194   (kernel K1)
 *     mpcqp_set_weights  <->  init_quadprog  src/controller/construct.jl:837-845       (kernel K2)
 *     mpcqp_set_bounds   <->  setconstraint! src/controller/construct.jl:324-559 (bound vectors,
 *                             softness columns, Inf = absent row: transcription.jl:692-700)
 *     mpcqp_step         <->  initpred! + linconstraint! + optim_objective! + getinput!
 *                             src/controller/execute.jl:247-277, transcription.jl:811-848,
 *                             execute.jl:466-505, execute.jl:536-546                   (kernel K3)
 *
 * Conventions
 *   - plain C, no C++/torch types; every array is float64 unless stated.
 *   - batch layout: problem-major, COLUMN-major inside a problem, i.e. exactly a Julia
 *     Array{Float64,3} of size (rows, cols, B) / Array{Float64,2} of size (n, B) passed as
 *     Ptr{Float64} with zero copy.
 *   - all signals are DEVIATION variables (operating points already subtracted), as the
 *     reference stores them (con.U0min = umin - Uop, src/controller/construct.jl:359).
 *   - +-Inf in a bound means "row absent" (the i_b rule).  A NULL bound pointer means the whole
 *     group is absent; a NULL softness pointer means the reference defaults (0 for u and Δu,
 *     1 for y and x̂end, src/controller/construct.jl:909-913).
 *   - host entry points (`*_host` suffix omitted) take HOST pointers and are synchronous, like
 *     `moveinput!`; the `_device` twins take DEVICE pointers of the handle's GPU, enqueue on the
 *     given hipStream_t (passed as void*) and return immediately -- this is what a resident
 *     closed loop (and bench.py) uses.
 *   - return value: 0 = ok, negative = API misuse (mirrors the DimensionMismatch/ArgumentError
 *     of validate_args, src/controller/construct.jl:702-710).  Nothing throws across the ABI.
 *   - per-problem solver outcome in status[]: see MPCQP_STATUS_*; on MPCQP_STATUS_ERROR the
 *     returned Z̃ is the shifted warm start (src/controller/execute.jl:499-500).
 *   - a handle is NOT thread-safe (neither is a LinMPC: its buffers are mutated in place,
 *     src/controller/construct.jl:4-16); distinct handles are independent.
 */
#ifndef MPCQP_H
#define MPCQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpcqp_handle_s* mpcqp_handle;

/* error codes (return values) */
#define MPCQP_OK                 0
#define MPCQP_ERR_NULL          -1   /* required pointer is NULL                              */
#define MPCQP_ERR_DIMS          -2   /* inconsistent dimensions (DimensionMismatch)           */
#define MPCQP_ERR_ARG           -3   /* illegal value (ArgumentError)                         */
#define MPCQP_ERR_UNSUPPORTED   -4   /* legal in the reference, not built here (see DESIGN)   */
#define MPCQP_ERR_ORDER         -5   /* call order: model/weights/bounds before step          */
#define MPCQP_ERR_DEVICE        -6   /* HIP runtime error (mpcqp_last_hip_error for detail)   */
#define MPCQP_ERR_NOMEM         -7

/* per-problem solver status, cf. issolved/iserror src/general.jl:52-61 */
#define MPCQP_STATUS_OPTIMAL          0
#define MPCQP_STATUS_ITERATION_LIMIT  1   /* solution kept (@warn branch, execute.jl:491-496)  */
#define MPCQP_STATUS_ERROR            2   /* infeasible / numerical: warm start returned       */

/* flags of mpcqp_dims.flags */
#define MPCQP_FLAG_RY_CONSTANT   (1u << 0)  /* Ry is (ny,B): ry held over Hp, like R̂y=repeat(ry,Hp) */
#define MPCQP_FLAG_COLD_START    (1u << 1)  /* ignore Ztilde on input: warm start = 0              */
#define MPCQP_FLAG_KEEP_QP       (1u << 2)  /* keep q̃ and F of the last step for mpcqp_get         */
#define MPCQP_FLAG_WARM_DUAL     (1u << 3)  /* closed loop: keep the multipliers between steps and  */
                                            /* start the next solve around them (and the shifted Z̃) */
#define MPCQP_FLAG_NO_POLISH     (1u << 4)  /* skip the active-set polish of the interior-point iterate */
                                            /* (measurements only: the polish is what bounds the error) */
#define MPCQP_FLAG_KEEP_ITERATE  (1u << 5)  /* diagnostics: a solve that stops at max_iter returns its iterate as it is  */
                                            /* (status ITERATION_LIMIT) instead of taking the error branch -- with       */
                                            /* mpcqp_set_iteration_limit(k) this exposes the k-th interior-point iterate; */
                                            /* mpcqp_prepare compares the first iterates of an on-demand kernel with the  */
                                            /* runtime-dimension kernel's this way                                        */

typedef struct {
    int32_t  batch;     /* B: number of independent controllers                              */
    int32_t  nxhat;     /* augmented states nx̂ (estimator/construct.jl:305-323)               */
    int32_t  nu, ny, nd;
    int32_t  Hp;        /* prediction horizon                                                 */
    int32_t  Hc;        /* number of free moves = length(nb) (construct.jl:629-660)           */
    const int32_t* nb;  /* [Hc] move-blocking lengths, sum == Hp; NULL = [1,..,1,Hp-Hc+1]     */
    int32_t  neps;      /* 1 iff Cwt is finite (slack variable ϵ present, construct.jl:903)   */
    int32_t  device;    /* HIP device ordinal                                                 */
    uint32_t flags;     /* MPCQP_FLAG_*                                                       */
    int32_t  max_iter;  /* interior-point iteration cap, 0 = default (80)                      */
    double   gap_tol;   /* absolute mean complementarity target, 0 = default (1e-12)          */
    double   res_tol;   /* relative primal/dual residual target, 0 = default (1e-11)          */
    double   dual_reg;  /* dual proximal regularisation δ, 0 = default (1e-12)                */
} mpcqp_dims;

/* sizes derived from dims (for allocating caller-side arrays) */
typedef struct {
    int32_t nZ;      /* nu*Hc + neps  (decision vector Z̃ = [ΔU; ϵ])                         */
    int32_t nDU;     /* nu*Hc                                                                */
    int32_t nU;      /* nu*Hp                                                                */
    int32_t nY;      /* ny*Hp                                                                */
    int32_t nD;      /* nd*Hp                                                                */
} mpcqp_sizes;

const char* mpcqp_version(void);
const char* mpcqp_strerror(int code);
const char* mpcqp_last_hip_error(void);

int mpcqp_create(const mpcqp_dims* dims, mpcqp_handle* out);
int mpcqp_destroy(mpcqp_handle h);
int mpcqp_get_sizes(mpcqp_handle h, mpcqp_sizes* out);

/* Augmented model of every controller (estim.Â, B̂u, Ĉ, B̂d, D̂d, f̂op - x̂op;
 * src/controller/transcription.jl:118,190).  Ahat (nx̂,nx̂,B), Bu (nx̂,nu,B), C (ny,nx̂,B),
 * Bd (nx̂,nd,B) / Dd (ny,nd,B) (NULL iff nd == 0), fop_minus_xop (nx̂,B) or NULL (= 0).
 * Runs K1 (prediction tables) and, when weights are already set, K2 (Hessian) -- this is also
 * the `setmodel!` path (src/controller/execute.jl:684-790).                                  */
int mpcqp_set_model(mpcqp_handle h, const double* Ahat, const double* Bu, const double* C,
                    const double* Bd, const double* Dd, const double* fop_minus_xop);

/* Diagonal weights: Mdiag (nY,B), Ndiag (nDU,B), Ldiag (nU,B), Cwt (B) (ignored when neps == 0).
 * (`Diagonal(repeat(Mwt,Hp))` etc., src/controller/linmpc.jl:236-238.)  Runs K2 when the model
 * is set.  Block-diagonal M_Hp: mpcqp_set_output_weight_blocks; dense M_Hp / N_Hc / L_Hp: mpcqp_set_dense_weights. */
int mpcqp_set_weights(mpcqp_handle h, const double* Mdiag, const double* Ndiag,
                      const double* Ldiag, const double* Cwt);

/* Transcription of the handle (the `transcription` keyword of LinMPC, src/controller/linmpc.jl:205-216):
 *   MPCQP_SINGLE_SHOOTING    Z = ΔU, the condensed QP (default; kernels K1-K3)
 *   MPCQP_MULTIPLE_SHOOTING  Z = [ΔU; X̂0], the model as equality constraints (src/controller/transcription.jl:196-240,
 *                            303-414, 913-928): solved in its stage form by a Riccati recursion inside the interior-point
 *                            iteration -- what the reference recommends when cond(H̃) is large (unstable plants, long
 *                            horizons; src/controller/construct.jl:855-866).  Same objective, bounds and softness, hence the
 *                            same optimal ΔU; mpcqp_step writes Z̃ = [ΔU; ϵ] as always and X̂0 is read with
 *                            mpcqp_get(MPCQP_GET_XHAT_MS).
 * Not every handle can run MultipleShooting: mpcqp_transcription_supported returns 0 when it can, else a bit mask
 * (1 block / dense weight matrices, 2 custom linear constraints, 4 stage data beyond 160 KB of LDS, 8 KEEP_QP / WARM_DUAL
 * flags); a step of an unsupported MultipleShooting handle returns MPCQP_ERR_UNSUPPORTED, and so do mpcqp_prepare and
 * mpcqp_kernel_kind for it (the Python mirror then keeps the SingleShooting kernels and says so).  The fused Kalman loop
 * (mpcqp_loop_device) runs on the stage-structured kernel too (round 6; MPCQP_ERR_UNSUPPORTED before).
 *
 * Size: a SingleShooting handle with nZ̃ = nu Hc + nϵ > 256 has no condensed kernel (its Newton matrix does not fit the
 * LDS); it runs the same QP on the stage-structured kernel, whose cost is linear in the horizons -- the reference has no
 * size limit (transcription.jl:2-4).  mpcqp_kernel_kind reports MPCQP_KERNEL_MS for it; the condensed tables
 * (MPCQP_GET_HESSIAN, _STEPRESP, _KMAT, _BVEC) do not exist for such a handle (MPCQP_ERR_UNSUPPORTED).  The same holds for a
 * handle of nZ̃ <= 256 whose condensed problem does not fit the 160 KB of LDS of a CU (nZ̃ beyond ~165 at nu = ny = 3 ... 4): its
 * steps run on the stage-structured kernel (MPCQP_KERNEL_MS), its prediction tables exist, its packed Hessian does not. */
#define MPCQP_SINGLE_SHOOTING    0
#define MPCQP_MULTIPLE_SHOOTING  1
int mpcqp_set_transcription(mpcqp_handle h, int32_t transcription);
int mpcqp_transcription_supported(mpcqp_handle h);

/* Interior-point iteration cap of the following steps (0 = default, 80); the analogue of the solver time limit the
 * reference sets from Ts (src/controller/linmpc.jl:329, src/general.jl:110-121).  With MPCQP_FLAG_KEEP_ITERATE a capped
 * solve returns its iterate (diagnostics). */
int mpcqp_set_iteration_limit(mpcqp_handle h, int32_t max_iter);

/* Replace the MPCQP_FLAG_* set of the handle (they are read at every step): e.g. switch between a
 * set point held over the horizon (Ry (ny,B), MPCQP_FLAG_RY_CONSTANT) and a full R̂y (nY,B), or
 * between cold and warm starts.  Unknown bits: MPCQP_ERR_ARG.                                    */
int mpcqp_set_flags(mpcqp_handle h, uint32_t flags);

/* Block-diagonal output weight M_Hp = blkdiag(M_1, ..., M_Hp) with symmetric ny x ny blocks,
 * Mblk (ny,ny,Hp,B); replaces Mdiag of mpcqp_set_weights (call that first: N, L, C come from it).
 * This is the `M_Hp=` keyword of LinMPC (src/controller/linmpc.jl:205-214) for the weights the
 * reference's own tests use: a terminal cost, M_Hp = blkdiag(M, ..., M, P)
 * (test/3_test_predictive_control.jl:498-527).  NULL returns to the diagonal weight.  A weight that
 * couples different prediction steps goes through mpcqp_set_dense_weights.                        */
int mpcqp_set_output_weight_blocks(mpcqp_handle h, const double* Mblk);

/* Dense (Hermitian) weight matrices, the reference's M_Hp=, N_Hc=, L_Hp= keywords in full generality
 * (src/controller/construct.jl:45-93, 837-845; linmpc.jl:205-214): M_Hp (nY,nY,B) may couple different
 * prediction steps, N_Hc (nDU,nDU,B) different moves, L_Hp (nU,nU,B) different inputs/steps.  Each replaces the
 * corresponding diagonal of mpcqp_set_weights (NULL: keep the diagonal / block form); call after
 * mpcqp_set_weights.  H~ is rebuilt (K2, runtime-dimension kernel).  With a dense M_Hp or L_Hp the step
 * evaluates M (F - R^y) and L (Tu lastu0 - R^u) densely: on an on-demand specialisation that carries these products
 * (mpcqp_prepare AFTER this call; the kernels compiled into the library do not) or on the runtime-dimension kernel; a
 * dense N_Hc only changes H~ and keeps every specialised step kernel.                                           */
int mpcqp_set_dense_weights(mpcqp_handle h, const double* M_Hp, const double* N_Hc, const double* L_Hp);

/* Custom linear inequality constraints over k .. k+Hp (keywords Wy, Wu, Wd, Wr of LinMPC,
 * src/controller/construct.jl:666-694; relaxW :1086-1160; linconstraint_custom!,
 * src/controller/execute.jl:337-364):
 *     wmin <= Wy ŷe + Wu ue + Wd d̂e + Wr r̂e <= wmax,     nw rows per step, Hp+1 steps
 * Wy (nw,ny,B), Wu (nw,nu,B), Wd (nw,nd,B) or NULL, Wr (nw,ny,B) or NULL.  The constraints are
 * written on engineering values while this ABI works in deviation variables: w_op (nw,B) =
 * Wy yop + Wu uop + Wd dop + Wr yop (NULL = 0) carries the operating points.  r̂e(k): see
 * mpcqp_set_current_setpoint.  nw = 0 removes them.
 * Bounds and softness (default 1, like c_wmin/c_wmax): Wmin, Wmax, C_wmin, C_wmax (nw (Hp+1), B),
 * NULL = absent (±Inf).  Problems with custom constraints run on an on-demand specialisation that has the rows
 * compiled in (mpcqp_prepare after this call) or on the runtime-dimension kernel.                   */
int mpcqp_set_custom_constraints(mpcqp_handle h, int nw, const double* Wy, const double* Wu,
                                 const double* Wd, const double* Wr, const double* w_op);
int mpcqp_set_custom_bounds(mpcqp_handle h, const double* Wmin, const double* Wmax,
                            const double* C_wmin, const double* C_wmax);
/* r̂e(k), the first ny entries of the extended set point vector the Wr term multiplies, is the CURRENT
 * set point ry(k) (src/controller/execute.jl:351).  With a held set point (MPCQP_FLAG_RY_CONSTANT) the
 * library has it already; with a full R̂y trajectory pass ry(k) - yop here, (ny,B), before the step
 * (NULL: back to the first block of R̂y).                                                          */
int mpcqp_set_current_setpoint(mpcqp_handle h, const double* ry_now);

/* Bounds (deviation values) and softness (ECR) vectors; shapes (nU,B), (nDU,B), (nY,B), (nx̂,B).
 * Field order follows ControllerConstraint (src/controller/construct.jl:126-199).            */
typedef struct {
    const double *U0min, *U0max;       /* (nU,B)  */
    const double *DUmin, *DUmax;       /* (nDU,B) */
    const double *Y0min, *Y0max;       /* (nY,B)  */
    const double *x0min, *x0max;       /* (nx̂,B)  terminal */
    const double *C_umin, *C_umax;     /* (nU,B)  softness, NULL = 0 */
    const double *C_dumin, *C_dumax;   /* (nDU,B) NULL = 0 */
    const double *C_ymin, *C_ymax;     /* (nY,B)  NULL = 1 */
    const double *c_x0min, *c_x0max;   /* (nx̂,B)  NULL = 1 */
} mpcqp_bounds;
int mpcqp_set_bounds(mpcqp_handle h, const mpcqp_bounds* bounds);

/* One control period for the whole batch (== B calls of moveinput!, execute.jl:59-80).
 *   xhat0  (nx̂,B)  estim.x̂0                       lastu0 (nu,B)  u(k-1) - uop
 *   Ry     (nY,B) or (ny,B) with MPCQP_FLAG_RY_CONSTANT: R̂y - Yop (deviation setpoints)
 *   Ru     (nU,B) R̂u - Uop, or NULL (= 0, the default R̂u = Uop)
 *   d0     (nd,B), Dhat0 (nD,B): measured disturbance and its preview (NULL iff nd == 0)
 *   Ztilde (nZ,B) in: previous optimum (shifted inside, transcription.jl:997-1007);
 *                 out: optimum (or the shifted warm start when status == ERROR)
 *   u0     (nu,B) out: u(k) - uop = Z̃[1:nu] + lastu0 (execute.jl:536-546)
 *   status (B) int32, iters (B) int32 (NULL allowed for iters)
 *   Yhat0  (nY,B) optional out: Ŷ0 = Ẽ Z̃ + F (predict!, transcription.jl:1136-1145); NULL ok
 */
int mpcqp_step(mpcqp_handle h, const double* xhat0, const double* lastu0, const double* Ry,
               const double* Ru, const double* d0, const double* Dhat0, double* Ztilde,
               double* u0, int32_t* status, int32_t* iters, double* Yhat0);

/* Same with DEVICE pointers on the handle's GPU, asynchronous on `stream` (a hipStream_t). */
int mpcqp_step_device(mpcqp_handle h, const double* xhat0, const double* lastu0,
                      const double* Ry, const double* Ru, const double* d0, const double* Dhat0,
                      double* Ztilde, double* u0, int32_t* status, int32_t* iters,
                      double* Yhat0, void* stream);

/* One control period of a resident closed loop in ONE launch: preparestate! (SteadyKalmanFilter
 * correction with the measurement y0m (nym,B), kalman.jl:284-295), moveinput! and updatestate!
 * (kalman.jl:298-309, with the input just computed) inside the step kernel.  xhat0 (nx̂,B) is updated
 * in place: in = x̂0(k|k-1), out = x̂0(k+1|k); everything else as mpcqp_step_device (u0 of one period
 * is lastu0 of the next: swap the two buffers).  Needs mpcqp_kf_set.  DEVICE pointers.          */
int mpcqp_loop_device(mpcqp_handle h, double* xhat0, const double* y0m, const double* lastu0,
                      const double* Ry, const double* Ru, const double* d0, const double* Dhat0,
                      double* Ztilde, double* u0, int32_t* status, int32_t* iters, double* Yhat0,
                      void* stream);

/* Re-run K1+K2 on the resident model/weights (timing of the setmodel! path). */
int mpcqp_recondense_device(mpcqp_handle h, void* stream);

/* Read back condensed quantities for parity tests / getinfo (host pointers).
 *   HESSIAN  H̃ (nZ,nZ,B)                       construct.jl:837-845
 *   STEPRESP Σ_m = Ĉ S(m) B̂u (ny,nu,Hp,B)       transcription.jl:134-139 (E is block-Toeplitz in it)
 *   KMAT     K (nY,nx̂,B)                        transcription.jl:143-147
 *   BVEC     B (nY,B)                           transcription.jl:184-192
 *   QTILDE   q̃ (nZ,B), FVEC F (nY,B) of the last step (needs MPCQP_FLAG_KEEP_QP)
 */
#define MPCQP_GET_HESSIAN   1
#define MPCQP_GET_STEPRESP  2
#define MPCQP_GET_KMAT      3
#define MPCQP_GET_BVEC      4
#define MPCQP_GET_QTILDE    5
#define MPCQP_GET_FVEC      6
#define MPCQP_GET_AUDIT     7   /* (4,B) of the last step: final complementarity gap mu, dual residual / its scale, primal
                                 * residual / its scale, 1.0 if the returned point passed the active-set polish's KKT check */
#define MPCQP_GET_XHAT_MS   8   /* (nx̂,Hp,B): X̂0(k+1..k+Hp) of the last MultipleShooting step -- the second block of the
                                 * reference's decision vector Z = [ΔU; X̂0] (src/controller/transcription.jl:5-7) */
#define MPCQP_GET_MS_DEFECT 9   /* (B): max |E_S Z + F_S| (defect of the model equations, transcription.jl:303-327) of it */
int mpcqp_get(mpcqp_handle h, int which, double* out);

/* ---- next row: the SteadyKalmanFilter steps on both sides of moveinput! --------
 * `preparestate!` -> correct_estimate_obsv! (src/estimator/kalman.jl:284-295):
 *       x̂0 += K̂ (y0m - Ĉm x̂0 - D̂dm d0)
 * `updatestate!`  -> predict_estimate_obsv! (src/estimator/kalman.jl:298-309):
 *       x̂0 <- Â x̂0 + B̂u u0 + B̂d d0 + (f̂op - x̂op)
 * on the model given to mpcqp_set_model, so that a closed loop keeps x̂0 resident on the GPU.
 * Khat (nx̂,nym,B) is the steady-state gain (the reference gets it from
 * ControlSystemsBase.kalman at construction, kalman.jl:204-236: host-side, once);
 * i_ym [nym] are the 0-based indices of the measured outputs.  nx̂ <= 64.                     */
int mpcqp_kf_set(mpcqp_handle h, const double* Khat, const int32_t* i_ym, int32_t nym);
/* xhat0 (nx̂,B) in/out, y0m (nym,B), d0 (nd,B) or NULL, u0 (nu,B): host pointers, synchronous */
int mpcqp_kf_correct(mpcqp_handle h, double* xhat0, const double* y0m, const double* d0);
int mpcqp_kf_predict(mpcqp_handle h, double* xhat0, const double* u0, const double* d0);
/* device pointers, asynchronous on `stream` */
int mpcqp_kf_correct_device(mpcqp_handle h, double* xhat0, const double* y0m, const double* d0, void* stream);
int mpcqp_kf_predict_device(mpcqp_handle h, double* xhat0, const double* u0, const double* d0, void* stream);


/* ---- which kernel runs a step; building specialised kernels ahead of the control loop -----------
 * A step runs on (MPCQP_KERNEL_AOT) a kernel specialised on the handle's dimensions that is compiled
 * into the library (the BASELINE shapes, csrc/mpcqp_dispatch.h), (MPCQP_KERNEL_ONDEMAND) a kernel
 * specialised on demand for these dimensions and this pattern of constraint groups, compiled with
 * the installation's hipcc and cached (csrc/mpcqp_kernels.hip), or (MPCQP_KERNEL_GENERIC) the
 * runtime-dimension kernel (same source and numerics, about 4x slower; also the kernel of problems
 * with nZ > 128).  mpcqp_step NEVER compiles: without a call of
 * mpcqp_prepare (or a cached object from an earlier run / from mpcqp_prebuild) it uses the generic
 * kernel.  The JuMP analogue is the one-time model build of init_optimization!
 * (src/controller/linmpc.jl:303-339), which also happens before the control loop.
 *   mpcqp_prepare      after mpcqp_set_bounds (the pattern of constraint groups is part of the key):
 *                      compiles if needed (seconds, once per shape and machine), loads, returns the
 *                      MPCQP_KERNEL_* kind the steps will run on (>= 0) or a negative error code.
 *                      An on-demand kernel that has not been checked on this machine is first compared
 *                      with the runtime-dimension kernel on a few controllers of the handle (one cold
 *                      step, pseudo-random states and set points); it is used only if the two agree -- optimum and
 *                      status, and the specialisation does not need half as many iterations again --
 *                      (marker <object>.ok), otherwise it is renamed <object>.bad, the generic kernel
 *                      is used and mpcqp_last_build_error says why.  Objects are keyed by the kernel
 *                      revision AND the identity of the compiler binary that built them.
 *   mpcqp_kernel_kind  the kind a step would run on right now; never compiles.
 *   mpcqp_row_groups   the handle's pattern of constraint groups (bit g: group g may hold finite rows;
 *                      0 box lower [hard dUmin; also the row eps >= 0 when no output-bound group exists -- with
 *                      one, that row rides there as an extra row with a zero row of E, softness 1, bound 0 (kernel
 *                      revision 10: a whole row slot less in the step kernel)], 1 box upper, 2 Umin, 3 Umax,
 *                      4 soft dUmin, 5 soft dUmax, 6 Ymin, 7 Ymax, 8 xhat-min, 9 xhat-max, 10 Wmin, 11 Wmax).
 *                      Use the value a handle of the same constraint pattern reports, e.g. BASELINE config 3 (hard
 *                      umin / umax, soft ymax): 0x8c.
 *   mpcqp_prebuild     compile-only (no GPU, no handle): for build pipelines; dims->batch is ignored.  Returns the
 *                      MPCQP_KERNEL_* kind of the shape (>= 0) or a negative error code.  The package ships a manifest
 *                      of shapes (spec_manifest.txt: `nu ny nxhat Hp Hc neps row_groups`) that its build and
 *                      `python -m mpcqp.prebuild [manifest]` turn into cached objects, so that a machine without
 *                      hipcc still runs a declared set of shapes on specialised kernels.  A row_groups value with bit 0
 *                      next to an output-bound group and without bit 1 (a manifest line written before kernel revision
 *                      10, e.g. 0x8d) is still compiled, but mpcqp_last_build_error then says that only a controller
 *                      with hard dUmin and no dUmax bounds matches the object (0x8d -> 0x8c).
 *                      Beyond one row per lane (nZ~ > 64) the object holds a kernel that runs a team of two or four
 *                      wavefronts per controller when the problem's LDS footprint leaves SIMDs idle (round 6: same
 *                      results, same ABI; -DMPCQP_TEAM=1 in MPCQP_JIT_FLAGS keeps one wavefront per controller).
 *   mpcqp_last_build_error  text of the last failed build of this thread (or the note above).
 * Environment: MPCQP_CACHE_DIR (default: <library dir>/spec_cache if writable, else
 * $XDG_CACHE_HOME/mpcqp or ~/.cache/mpcqp), HIPCC (compiler binary), MPCQP_JIT=0 (never specialise). */
#define MPCQP_KERNEL_GENERIC   0
#define MPCQP_KERNEL_AOT       1
#define MPCQP_KERNEL_ONDEMAND  2
#define MPCQP_KERNEL_SMALL     3   /* nZ~ <= 16 with box, input-bound and (<= 64 in all) output-bound / terminal rows: four controllers per wavefront
                                    * (csrc/mpcqp_small_bodies.h); a step that fuses the Kalman steps (mpcqp_loop_device)
                                    * runs on the kernel the other rules select, and so does a handle WITH output-bound / terminal
                                    * rows of at most 1024 controllers when its own specialisation is available (measured 1.5 times
                                    * faster there; MPCQP_SMALL_Y=1 keeps it on this kernel) */
#define MPCQP_KERNEL_MS        4   /* MultipleShooting handles (mpcqp_set_transcription): the stage-structured kernel
                                    * (csrc/ms_bodies.h): Riccati recursion over the horizon, H̃ and E never formed */
int mpcqp_prepare(mpcqp_handle h);
/* LDS bytes one problem needs in the step kernel (160 KB per CU: the number of problems resident per CU follows). */
int mpcqp_lds_bytes(mpcqp_handle h);
int mpcqp_kernel_kind(mpcqp_handle h);
int mpcqp_row_groups(mpcqp_handle h, uint32_t* row_groups);
int mpcqp_prebuild(const mpcqp_dims* dims, uint32_t row_groups);
const char* mpcqp_last_build_error(void);

/* ---- several GPUs of one node behind one handle (contiguous shards, no coupling) -------
 * mpcqp_multi_create makes one handle per entry of device_ids (an ordinal may repeat) and gives
 * device g the problems [off_g, off_g + B_g) of the batch, B_g = floor(B/ndev) (+1 for the first
 * B mod ndev devices).  Every array is problem-major, so a shard is ONE contiguous slice of each
 * caller array: the set_* / step entry points below take the whole-batch HOST arrays of their
 * single-device twins, slice them, and drive all devices concurrently (one stream per device; the
 * step is enqueued everywhere before anything is awaited; results are copied back into the caller's
 * arrays slice by slice -- the gather).  mpcqp_multi_handle gives the single-device handle of a shard
 * for everything else (mpcqp_get, kf_*, prepare ...); mpcqp_multi_shard its offset and size.
 * Device-resident data: call mpcqp_step_device on the shard handles with per-device pointers, and
 * mpcqp_multi_gather_device to collect Z (nZ,B), u0 (nu,B), status (B) of all shards into buffers on
 * one root device (hipMemcpyPeerAsync over xGMI; stream = a stream of the root device, or NULL);
 * mpcqp_multi_scatter_device is the way in: per-period inputs born on one device, sliced to the shards. */
typedef struct mpcqp_multi_s* mpcqp_multi;
int mpcqp_multi_create(const mpcqp_dims* dims, const int32_t* device_ids, int32_t ndev, mpcqp_multi* out);
int mpcqp_multi_destroy(mpcqp_multi mh);
int mpcqp_multi_ndev(mpcqp_multi mh);
mpcqp_handle mpcqp_multi_handle(mpcqp_multi mh, int32_t g);
int mpcqp_multi_shard(mpcqp_multi mh, int32_t g, int32_t* offset, int32_t* count);
int mpcqp_multi_set_model(mpcqp_multi mh, const double* Ahat, const double* Bu, const double* C,
                          const double* Bd, const double* Dd, const double* fop_minus_xop);
int mpcqp_multi_set_weights(mpcqp_multi mh, const double* Mdiag, const double* Ndiag,
                            const double* Ldiag, const double* Cwt);
int mpcqp_multi_set_bounds(mpcqp_multi mh, const mpcqp_bounds* bounds);
int mpcqp_multi_prepare(mpcqp_multi mh);
int mpcqp_multi_step(mpcqp_multi mh, const double* xhat0, const double* lastu0, const double* Ry,
                     const double* Ru, const double* d0, const double* Dhat0, double* Ztilde,
                     double* u0, int32_t* status, int32_t* iters, double* Yhat0);
int mpcqp_multi_gather_device(mpcqp_multi mh, int32_t root, const double* const* Z_shards,
                              const double* const* u0_shards, const int32_t* const* status_shards,
                              double* Z_root, double* u0_root, int32_t* status_root, void* stream);

/* The scatter of a period's inputs that were born on ONE device (the estimates of a plant-wide observer, set points of
 * a supervisory layer): x̂0 (nx̂,B), lastu0 (nu,B) and Ry (ry_rows,B; ry_rows = ny with MPCQP_FLAG_RY_CONSTANT, else ny Hp)
 * on `root` are sliced into the per-device buffers of the shards, hipMemcpyPeerAsync over xGMI on `stream` (a stream of
 * the root device, or NULL: the root handle's stream, synchronised before returning).  Twin of the gather above.     */
int mpcqp_multi_scatter_device(mpcqp_multi mh, int32_t root, const double* xhat0_root, const double* lastu0_root,
                               const double* Ry_root, int32_t ry_rows, double* const* xhat0_shards,
                               double* const* lastu0_shards, double* const* Ry_shards, void* stream);

/* Device time of the kernels of the last step / recondense on this handle, measured with HIP
 * events on the stream they ran on (milliseconds; < 0 if not available).                     */
double mpcqp_last_step_ms(mpcqp_handle h);
double mpcqp_last_condense_ms(mpcqp_handle h);
double mpcqp_last_predmat_ms(mpcqp_handle h);    /* K1 part of the last condensation (K2 = the rest) */

#ifdef __cplusplus
}
#endif
#endif /* MPCQP_H */
