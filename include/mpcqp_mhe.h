/* mpcqp_mhe.h -- C ABI of the batched linear MovingHorizonEstimator on MI355X (libmpcqp.so).
 *
 * SURVEY 8 row f2 / BASELINE configs[4]: B independent MovingHorizonEstimator objects (LinModel,
 * SingleShooting, default KalmanFilter arrival covariance), one QP per period each, solved by
 * hand-written gfx950 kernels (csrc/mhe_bodies.h).  The entry points replace, for a batch,
 *   MovingHorizonEstimator(model; He, ...)      src/estimator/mhe/construct.jl:255-460   -> mpcqp_mhe_create + set_model
 *   setconstraint!(estim; x̂min, ..., v̂max)       src/estimator/mhe/construct.jl:858-1049  -> mpcqp_mhe_set_bounds
 *   init_estimate_cov!                          src/estimator/mhe/execute.jl:2-36        -> mpcqp_mhe_init
 *   setstate!                                   src/estimator/execute.jl:424-429         -> mpcqp_mhe_set_state
 *   preparestate!(estim, ym, d)                 src/estimator/mhe/execute.jl:44-57       -> mpcqp_mhe_prepare
 *   updatestate!(estim, u, ym, d)               src/estimator/mhe/execute.jl:76-88       -> mpcqp_mhe_update
 *   getinfo(estim)                              src/estimator/mhe/execute.jl:116-200     -> mpcqp_mhe_get
 * All arrays are deviation variables (x̂0 = x̂ - x̂op, y0m = ym - yop[i_ym], u0, d0): the host keeps the
 * operating points, exactly as the reference's estimator fields do.  "(n,B)" means a Julia array of
 * that shape, i.e. estimator b's n values are contiguous; matrices are column-major inside an estimator.
 *
 * Supported: nx̂ <= 16, nym <= 16 (one estimator per 16-lane DPP row), any nu, nd, He; both forms
 * (direct = true/false); growing and moving windows; hard bounds on x̂ (arrival state and window), ŵ, v̂
 * given per channel, hard (Cwt = Inf, the reference's default) or relaxed by one slack variable ε (finite Cwt and
 * softness c per channel, mpcqp_mhe_set_softness, or per channel and stage, mpcqp_mhe_set_softness_window).  Bounds that
 * change along the window (X̂min / Ŵmin / V̂min vectors): mpcqp_mhe_set_bounds_window.
 * There is no CPU fallback: every compute entry point needs a HIP device. */
#ifndef MPCQP_MHE_H
#define MPCQP_MHE_H

#include <stdint.h>

#include "mpcqp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpcqp_mhe_s* mpcqp_mhe;

#define MPCQP_MHE_KEEP_WINDOWS (1u << 0)   /* also keep V̂ and X̂ of every solve for mpcqp_mhe_get */

typedef struct {
    int32_t batch;      /* B                                                                     */
    int32_t nxhat;      /* states of the augmented model (nx + integrators), <= 16               */
    int32_t nu, nym, nd;
    int32_t He;         /* estimation horizon                                                    */
    int32_t direct;     /* 1: current form (default of the reference), 0: predictor form         */
    int32_t device;
    uint32_t flags;     /* MPCQP_MHE_*                                                            */
    int32_t max_iter;   /* 0: default (80)                                                       */
    double gap_tol, res_tol, dual_reg;   /* 0: defaults 1e-12, 1e-11 (as the LinMPC step) and 1e-10 */
} mpcqp_mhe_dims;

int mpcqp_mhe_create(const mpcqp_mhe_dims* dims, mpcqp_mhe* out);
int mpcqp_mhe_destroy(mpcqp_mhe h);

/* Augmented model and covariances of every estimator (host arrays):
 *   Ahat (nx̂,nx̂,B)  Bhu (nx̂,nu,B)  Chm (nym,nx̂,B)  Bhd (nx̂,nd,B)  Dhdm (nym,nd,B)   [Bhd, Dhdm NULL iff nd = 0]
 *   fx = f̂op - x̂op (nx̂,B) or NULL;  Qhat (nx̂,nx̂,B), Rhat (nym,nym,B): Q̂, R̂ (symmetric positive definite)
 * Builds the constant blocks on the device (2Q̂⁻¹, 2R̂⁻¹, Â'2Q̂⁻¹Â, -2Q̂⁻¹Â, Ĉm'2R̂⁻¹Ĉm, ...).              */
int mpcqp_mhe_set_model(mpcqp_mhe h, const double* Ahat, const double* Bhu, const double* Chm, const double* Bhd,
                        const double* Dhdm, const double* fx, const double* Qhat, const double* Rhat);

/* setconstraint!: per-channel hard bounds in deviation variables, (nx̂,B) / (nym,B) host arrays, NULL or
 * +-Inf entries = no bound.  x̂ bounds apply to the arrival state and to every state of the window
 * (stage-dependent bounds: mpcqp_mhe_set_bounds_window). */
int mpcqp_mhe_set_bounds(mpcqp_mhe h, const double* xmin, const double* xmax, const double* wmin, const double* wmax,
                         const double* vmin, const double* vmax);

/* Window-long bounds, setconstraint!(estim; X̂min, X̂max, Ŵmin, Ŵmax, V̂min, V̂max) (construct.jl:858-935): one bound per
 * channel AND stage.  Xmin / Xmax (nx̂ (He+1), B): the arrival state first, then the He window states oldest first;
 * Wmin / Wmax (nx̂ He, B); Vmin / Vmax (nym He, B); deviation variables, +-Inf = no bound, NULL = class absent.  A window
 * that is not full yet (Nk < He) uses the LAST Nk blocks, like the reference (trunc_bounds).  Replaces the bounds of
 * mpcqp_mhe_set_bounds (and vice versa).  Softness: per channel (mpcqp_mhe_set_softness) or per channel and stage
 * (mpcqp_mhe_set_softness_window).                                                                                  */
int mpcqp_mhe_set_bounds_window(mpcqp_mhe h, const double* Xmin, const double* Xmax, const double* Wmin, const double* Wmax,
                                const double* Vmin, const double* Vmax);

/* Soft constraints (MovingHorizonEstimator(...; Cwt) + setconstraint!(estim; c_x̂min, ..., c_v̂max), construct.jl:858-1049,
 * 1151-1288): Cwt (B) finite weights of ε² (NULL: Cwt = Inf, hard constraints only, the reference's default) and the
 * softness c >= 0 of each channel's rows, (nx̂,B) / (nym,B) or NULL (= 0: hard).  A row reads  g'z - c ε <= h  with one
 * slack ε >= 0 per estimator.  Softness without a finite Cwt: MPCQP_ERR_ARG (ArgumentError in the reference).      */
int mpcqp_mhe_set_softness(mpcqp_mhe h, const double* Cwt, const double* c_xmin, const double* c_xmax, const double* c_wmin,
                           const double* c_wmax, const double* c_vmin, const double* c_vmax);

/* Window-long softness, setconstraint!(estim; C_x̂min, C_x̂max, C_ŵmin, C_ŵmax, C_v̂min, C_v̂max) (construct.jl:937-1020):
 * one softness per channel AND stage, same shapes as the window-long bounds (C_xmin / C_xmax (nx̂ (He+1), B): arrival
 * state first; C_wmin / C_wmax (nx̂ He, B); C_vmin / C_vmax (nym He, B)); >= 0 and finite, NULL = 0 (hard).  Cwt (B) must
 * be given and finite (MPCQP_ERR_ARG otherwise).  The softness is a column of the reference's constraint matrices, which
 * are not truncated while the window grows: the rows of window entry j use block j whatever Nk (the BOUNDS of a growing
 * window use the last Nk blocks; transcription.jl:737-752).  Replaces the softness of mpcqp_mhe_set_softness (and vice versa). */
int mpcqp_mhe_set_softness_window(mpcqp_mhe h, const double* Cwt, const double* C_xmin, const double* C_xmax, const double* C_wmin,
                                  const double* C_wmax, const double* C_vmin, const double* C_vmax);

/* init_estimate_cov!: empties the data windows (Nk = 0), x̂0 <- xhat0 (nx̂,B; NULL: zeros), arrival
 * covariance P̄ <- P0 (nx̂,nx̂,B; required), d0(-1) <- d0_prev (nd,B; NULL: zeros), lastu0 (nu,B; NULL: zeros). */
int mpcqp_mhe_init(mpcqp_mhe h, const double* xhat0, const double* P0, const double* d0_prev, const double* lastu0);

/* setstate!(estim, x̂) (src/estimator/execute.jl:424-429): x̂0 <- xhat0 (nx̂,B) and nothing else -- the data windows,
 * the window length and the arrival covariance stay (the reference raises when a covariance is passed to a
 * MovingHorizonEstimator, mhe/execute.jl:938-941: callers must not offer one).                                    */
int mpcqp_mhe_set_state(mpcqp_mhe h, const double* xhat0);

/* setmodel!(estim, model) (src/estimator/execute.jl:483-497, mhe/execute.jl:943-1046) in two calls: mpcqp_mhe_set_model
 * with the new augmented model (and covariances), and this one for the operating points -- the data windows, lastu0,
 * x̂0 and x̂0arr are deviation variables, so a change of (yop, uop, dop, x̂op) moves them by
 * dy0m = yop_old - yop_new (nym,B), du0 = uop_old - uop_new (nu,B), dd0 (nd,B), dx0 = x̂op_old - x̂op_new (nx̂,B);
 * NULL = unchanged.  Bounds on x̂ are deviation variables too: send them again (mpcqp_mhe_set_bounds).            */
int mpcqp_mhe_shift_windows(mpcqp_mhe h, const double* dy0m, const double* du0, const double* dd0, const double* dx0);

/* preparestate!: current form: add (y0m, d0, lastu0) to the windows, correct the arrival covariance when
 * the window moves, solve the QP, x̂0 <- estimate.  Predictor form: nothing to do (returns MPCQP_OK).
 * y0m (nym,B), d0 (nd,B or NULL).  Returns the number of estimators whose solve failed (status != 0),
 * or a negative MPCQP_ERR_*.                                                                        */
int mpcqp_mhe_prepare(mpcqp_mhe h, const double* y0m, const double* d0);
/* updatestate!: predictor form: add (y0m, d0, u0) to the windows and solve; both forms: arrival
 * covariance update once the window is full; lastu0 <- u0.                                          */
int mpcqp_mhe_update(mpcqp_mhe h, const double* u0, const double* y0m, const double* d0);
/* same, inputs already on the handle's device (and outputs left there): nothing crosses PCIe;
 * the calls are asynchronous on the handle's stream and return MPCQP_OK or an error.               */
int mpcqp_mhe_prepare_device(mpcqp_mhe h, const double* y0m_dev, const double* d0_dev);
int mpcqp_mhe_update_device(mpcqp_mhe h, const double* u0_dev, const double* y0m_dev, const double* d0_dev);
int mpcqp_mhe_sync(mpcqp_mhe h);

#define MPCQP_MHE_XHAT0    0   /* (nx̂,B)          current estimate x̂0                                  */
#define MPCQP_MHE_ZTILDE   1   /* (nx̂+He nx̂,B)    [x̂0arr; Ŵ] of the last solve (zero beyond Nk)          */
#define MPCQP_MHE_STATUS   2   /* (B) int32        0 solved, 1 iteration limit, 2 failed (open-loop kept) */
#define MPCQP_MHE_ITERS    3   /* (B) int32                                                              */
#define MPCQP_MHE_PBAR     4   /* (nx̂,nx̂,B)       arrival covariance P̄                                  */
#define MPCQP_MHE_VHAT     5   /* (He nym,B)       V̂ of the last solve  (MPCQP_MHE_KEEP_WINDOWS)          */
#define MPCQP_MHE_XHATWIN  6   /* (He nx̂,B)        X̂0 of the last solve (MPCQP_MHE_KEEP_WINDOWS)          */
#define MPCQP_MHE_EPSILON  7   /* (B)              slack ε of the last solve (after mpcqp_mhe_set_softness)     */
int mpcqp_mhe_get(mpcqp_mhe h, int what, void* out);
/* device address of one of the arrays above (NULL if not kept) */
void* mpcqp_mhe_device_ptr(mpcqp_mhe h, int what);
int mpcqp_mhe_nk(mpcqp_mhe h);                 /* current window length Nk                           */
double mpcqp_mhe_last_ms(mpcqp_mhe h);          /* device time of the last solve kernel (HIP events)  */
int mpcqp_mhe_register_columns(mpcqp_mhe h);    /* NX: register columns of the kernel that runs       */

#ifdef __cplusplus
}
#endif
#endif
