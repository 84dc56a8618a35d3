/* CPU ORACLE (test infrastructure, NOT a product path) -- C restatement of one period of the linear
 * MovingHorizonEstimator (BASELINE configs[4]), the `cpu_baseline` leg of bench.py --config C5
 * ("kind": "port") and a second checker of oracle/mhe.py at batch sizes NumPy is too slow for.
 *
 * What a period does (current form, `direct = true`; /root/reference/src/estimator/mhe/execute.jl):
 *   add_data_windows!   :497-548   y0m(k), u0(k-1), x̂0(k-1) pushed into the windows (growing, then moving)
 *   correct_cov!        :727-745   arrival covariance P̄ corrected with the oldest measurement (moving window only)
 *   initpred! + linconstraint! + optim_objective!  :419-457, transcription.jl:732-782, execute.jl:576-618
 *                                  the QP   min (x(0)-x̄)'P̄⁻¹(x(0)-x̄) + Σ ŵ'Q̂⁻¹ŵ + Σ v̂'R̂⁻¹v̂   s.t. bounds on x̂
 *   getstate!           :629-643   x̂0(k) = last state of the window
 *   update_cov!         :755-781   P̄ <- Â P̄ Â' + Q̂ once the window is full
 * The reference condenses the window into Z̃ = [x̂0arr; Ŵ] (a dense (nx̂ + nx̂ He)² Hessian).  Like the GPU kernel
 * (csrc/mhe_bodies.h) this port solves the SAME QP in the state sequence X = (x(0) .. x(N)),
 *     ŵ(j) = x(j+1) - Â x(j) - B̂u u0(j),     v̂(i) = y0m(i) - Ĉm x(i+1),
 * a bijection of Z̃, where the Hessian is block tridiagonal -- diagonal blocks 2P̄⁻¹ + T1 | T1 + T2 + T3 | T2 + T3,
 * sub-diagonal -2Q̂⁻¹Â, T1 = Â'2Q̂⁻¹Â, T2 = 2Q̂⁻¹, T3 = Ĉm'2R̂⁻¹Ĉm -- and a bound on a state is a bound on a variable.
 * Solver: Mehrotra predictor-corrector on the normal equations (bounds only: Φ = H + diag), block-tridiagonal
 * Cholesky per iteration.  A fair structure-exploiting CPU baseline: ~2.4 Mflop per solve instead of the ~100 Mflop
 * of a dense 252-variable interior-point solve.
 * Scope: nd = 0, operating points 0, per-channel hard bounds |x̂| <= xabs (the C5 workload); OpenMP over estimators.
 * Parity pin: tests/test_oracle_mhe.py::test_c_port_matches_the_numpy_oracle (x̂0 after every period, 1e-7).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NXMAX 24
#define AT(M, i, j, n) ((M)[(i) + (n) * (j)]) /* column-major n x n */

/* in-place Cholesky (lower) of the n x n column-major SPD matrix; 0 ok */
static int chol(double* M, int n) {
    for (int j = 0; j < n; ++j) {
        double d = AT(M, j, j, n);
        for (int k = 0; k < j; ++k) d -= AT(M, j, k, n) * AT(M, j, k, n);
        if (!(d > 0)) return 1;
        d = sqrt(d);
        AT(M, j, j, n) = d;
        for (int i = j + 1; i < n; ++i) {
            double s = AT(M, i, j, n);
            for (int k = 0; k < j; ++k) s -= AT(M, i, k, n) * AT(M, j, k, n);
            AT(M, i, j, n) = s / d;
        }
    }
    return 0;
}
static void fwd(const double* L, double* x, int n) { /* L y = x */
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= AT(L, i, k, n) * x[k];
        x[i] = s / AT(L, i, i, n);
    }
}
static void bwd(const double* L, double* x, int n) { /* L' y = x */
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= AT(L, k, i, n) * x[k];
        x[i] = s / AT(L, i, i, n);
    }
}
static void inv_spd(const double* M, double* Mi, int n) {
    double L[NXMAX * NXMAX], e[NXMAX];
    memcpy(L, M, sizeof(double) * n * n);
    chol(L, n);
    for (int j = 0; j < n; ++j) {
        memset(e, 0, sizeof(double) * n);
        e[j] = 1.0;
        fwd(L, e, n);
        bwd(L, e, n);
        for (int i = 0; i < n; ++i) AT(Mi, i, j, n) = e[i];
    }
}

typedef struct {
    int nx, nu, nym, He;
    const double *A, *Bu, *Cm; /* column-major (nx,nx), (nx,nu), (nym,nx) */
    double Qi2[NXMAX * NXMAX]; /* 2 Q̂⁻¹ */
    double T1[NXMAX * NXMAX], T2T3[NXMAX * NXMAX], Oc[NXMAX * NXMAX], CtR2[NXMAX * NXMAX]; /* CtR2: Ĉm' 2R̂⁻¹ (nx x nym) */
    double Q[NXMAX * NXMAX], R[NXMAX * NXMAX];
} est_t;

/* one QP: stages 0..N (N = Nk), returns iterations; X (N+1)*nx in/out (start), status out */
static int solve_window(const est_t* e, int N, const double* Pbar_inv2 /* 2 P̄⁻¹ */, const double* xbar, const double* Yw,
                        const double* Uw, double xabs, double* X, int* status, double* work) {
    const int nx = e->nx, nu = e->nu, nym = e->nym, S = N + 1, n = S * nx;
    double* q = work;             /* n */
    double* D = q + n;            /* n  diag barrier */
    double* Ld = D + n;           /* S * nx*nx  diagonal Cholesky blocks */
    double* Lo = Ld + S * nx * nx;/* S * nx*nx  sub-diagonal blocks L_{s,s-1} */
    double* rhs = Lo + S * nx * nx;/* n */
    double* dz = rhs + n;         /* n */
    double* sl = dz + n, *su = sl + n, *ll = su + n, *lu = ll + n; /* slacks / multipliers of lower, upper bounds */
    double* dsl = lu + n, *dsu = dsl + n, *dll = dsu + n, *dlu = dll + n;
    double* hz = dlu + n;         /* n: H x */
    double g[NXMAX], t[NXMAX];
    /* linear term q: arrival + process + measurement parts */
    memset(q, 0, sizeof(double) * n);
    for (int i = 0; i < nx; ++i) {
        double s = 0;
        for (int k = 0; k < nx; ++k) s += AT(Pbar_inv2, i, k, nx) * xbar[k];
        q[i] = -s;
    }
    for (int j = 0; j < N; ++j) {
        for (int i = 0; i < nx; ++i) {
            double s = 0;
            for (int c = 0; c < nu; ++c) s += e->Bu[i + nx * c] * Uw[j * nu + c];
            g[i] = s;
        }
        for (int i = 0; i < nx; ++i) {                      /* t = 2Q̂⁻¹ g */
            double s = 0;
            for (int k = 0; k < nx; ++k) s += AT(e->Qi2, i, k, nx) * g[k];
            t[i] = s;
        }
        for (int i = 0; i < nx; ++i) {
            q[(j + 1) * nx + i] -= t[i];
            double s = 0;
            for (int k = 0; k < nx; ++k) s += AT(e->A, k, i, nx) * t[k];       /* Â' t */
            q[j * nx + i] += s;
            double m = 0;
            for (int a = 0; a < nym; ++a) m += e->CtR2[i + nx * a] * Yw[j * nym + a];
            q[(j + 1) * nx + i] -= m;
        }
    }
    const int m = 2 * n;
    const int bounded = isfinite(xabs);
    /* H x for the current X */
#define HX(x_, out_)                                                                                     \
    for (int s_ = 0; s_ < S; ++s_)                                                                       \
        for (int i = 0; i < nx; ++i) {                                                                   \
            double a_ = 0;                                                                               \
            const double* Db = s_ == 0 ? NULL : (s_ == N ? e->T2T3 : NULL);                              \
            for (int k = 0; k < nx; ++k) {                                                               \
                double d_ = s_ == 0 ? AT(Pbar_inv2, i, k, nx) + (N > 0 ? AT(e->T1, i, k, nx) : 0.0)      \
                            : s_ == N ? AT(e->T2T3, i, k, nx) : AT(e->T1, i, k, nx) + AT(e->T2T3, i, k, nx); \
                (void)Db;                                                                                \
                a_ += d_ * (x_)[s_ * nx + k];                                                            \
                if (s_ > 0) a_ += AT(e->Oc, i, k, nx) * (x_)[(s_ - 1) * nx + k];                         \
                if (s_ < N) a_ += AT(e->Oc, k, i, nx) * (x_)[(s_ + 1) * nx + k];                         \
            }                                                                                            \
            (out_)[s_ * nx + i] = a_;                                                                    \
        }
    if (!bounded) {      /* unconstrained: one block-tridiagonal solve */
        for (int i = 0; i < n; ++i) D[i] = 0.0;
    }
    /* starting point */
    for (int i = 0; i < n; ++i) {
        sl[i] = fmax(X[i] + xabs, 1.0); su[i] = fmax(xabs - X[i], 1.0);
        ll[i] = 10.0 / sl[i]; lu[i] = 10.0 / su[i];
    }
    int it = 0;
    *status = 1;
    double laststep = 1e300;
    for (; it < 100; ++it) {
        HX(X, hz);
        double mu = 0, rdn = 0, rpn = 0, ndd = 1;
        if (bounded)
            for (int i = 0; i < n; ++i) mu += sl[i] * ll[i] + su[i] * lu[i];
        mu /= m;
        for (int i = 0; i < n; ++i) {
            const double gl = bounded ? lu[i] - ll[i] : 0.0;
            const double r = hz[i] + q[i] + gl;
            rhs[i] = r;                                  /* r_d */
            rdn = fmax(rdn, fabs(r));
            ndd = fmax(ndd, fmax(fabs(hz[i]), fmax(fabs(q[i]), fabs(gl))));
            if (bounded) rpn = fmax(rpn, fmax(fabs(-X[i] + sl[i] - xabs), fabs(X[i] + su[i] - xabs)));
        }
        if (!(mu == mu) || !(rdn == rdn)) { *status = 2; break; }
        if ((!bounded && it > 0) || (bounded && mu <= 1e-12 && rdn <= 1e-10 * ndd && rpn <= 1e-10 * (1 + xabs) && laststep <= 1e-7)) {
            *status = 0;
            break;
        }
        /* Phi = H + diag(D), D = ll/sl + lu/su; block-tridiagonal Cholesky */
        if (bounded)
            for (int i = 0; i < n; ++i) D[i] = ll[i] / sl[i] + lu[i] / su[i];
        int bad = 0;
        for (int s = 0; s <= N && !bad; ++s) {
            double* Ls = Ld + s * nx * nx;
            for (int j = 0; j < nx; ++j)
                for (int i = 0; i < nx; ++i) {
                    double d = s == 0 ? AT(Pbar_inv2, i, j, nx) + (N > 0 ? AT(e->T1, i, j, nx) : 0.0)
                               : s == N ? AT(e->T2T3, i, j, nx) : AT(e->T1, i, j, nx) + AT(e->T2T3, i, j, nx);
                    AT(Ls, i, j, nx) = d + (i == j ? D[s * nx + i] : 0.0);
                }
            if (s > 0) {
                /* Lo_s = Oc Ld_{s-1}^{-T}: solve Ld_{s-1} Lo_s' = Oc'  row by row */
                double* Lp = Ld + (s - 1) * nx * nx;
                double* Los = Lo + s * nx * nx;
                for (int i = 0; i < nx; ++i) {
                    double row[NXMAX];
                    for (int k = 0; k < nx; ++k) row[k] = AT(e->Oc, i, k, nx);
                    fwd(Lp, row, nx);
                    for (int k = 0; k < nx; ++k) AT(Los, i, k, nx) = row[k];
                }
                for (int j = 0; j < nx; ++j)
                    for (int i = j; i < nx; ++i) {
                        double a = 0;
                        for (int k = 0; k < nx; ++k) a += AT(Los, i, k, nx) * AT(Los, j, k, nx);
                        AT(Ls, i, j, nx) -= a;
                        AT(Ls, j, i, nx) = AT(Ls, i, j, nx);
                    }
            }
            bad = chol(Ls, nx);
        }
        if (bad) { *status = 2; break; }
#define SOLVE(v_)                                                                        \
    for (int s_ = 0; s_ <= N; ++s_) {                                                    \
        if (s_ > 0) {                                                                    \
            const double* Los = Lo + s_ * nx * nx;                                       \
            for (int i = 0; i < nx; ++i) {                                               \
                double a_ = 0;                                                           \
                for (int k = 0; k < nx; ++k) a_ += AT(Los, i, k, nx) * (v_)[(s_ - 1) * nx + k]; \
                (v_)[s_ * nx + i] -= a_;                                                 \
            }                                                                            \
        }                                                                                \
        fwd(Ld + s_ * nx * nx, (v_) + s_ * nx, nx);                                      \
    }                                                                                    \
    for (int s_ = N; s_ >= 0; --s_) {                                                    \
        if (s_ < N) {                                                                    \
            const double* Los = Lo + (s_ + 1) * nx * nx;                                 \
            for (int i = 0; i < nx; ++i) {                                               \
                double a_ = 0;                                                           \
                for (int k = 0; k < nx; ++k) a_ += AT(Los, k, i, nx) * (v_)[(s_ + 1) * nx + k]; \
                (v_)[s_ * nx + i] -= a_;                                                 \
            }                                                                            \
        }                                                                                \
        bwd(Ld + s_ * nx * nx, (v_) + s_ * nx, nx);                                      \
    }
        if (!bounded) {
            for (int i = 0; i < n; ++i) dz[i] = -rhs[i];
            SOLVE(dz);
            for (int i = 0; i < n; ++i) X[i] += dz[i];
            continue;
        }
        /* predictor (sigma = 0) and corrector share the factor.  Rows: lower  -x + sl = xabs, upper  x + su = xabs */
        double sigmu = 0.0, alpha = 1.0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < n; ++i) {
                const double rpl = -X[i] + sl[i] - xabs, rpu = X[i] + su[i] - xabs;
                const double rcl = sl[i] * ll[i] - sigmu + (pass ? dsl[i] * dll[i] : 0.0);
                const double rcu = su[i] * lu[i] - sigmu + (pass ? dsu[i] * dlu[i] : 0.0);
                /* (H + D) dz = -rd - G'(D rp - rc/s), G = -1 (lower), +1 (upper) */
                dz[i] = -rhs[i] + (ll[i] / sl[i] * rpl - rcl / sl[i]) - (lu[i] / su[i] * rpu - rcu / su[i]);
            }
            SOLVE(dz);
            double amax = 1e300;
            for (int i = 0; i < n; ++i) {
                const double rpl = -X[i] + sl[i] - xabs, rpu = X[i] + su[i] - xabs;
                const double rcl = sl[i] * ll[i] - sigmu + (pass ? dsl[i] * dll[i] : 0.0);
                const double rcu = su[i] * lu[i] - sigmu + (pass ? dsu[i] * dlu[i] : 0.0);
                const double nsl = -rpl + dz[i], nsu = -rpu - dz[i];            /* ds = -rp - G dz */
                const double nll = -(rcl + ll[i] * nsl) / sl[i], nlu = -(rcu + lu[i] * nsu) / su[i];
                dsl[i] = nsl; dsu[i] = nsu; dll[i] = nll; dlu[i] = nlu;
                if (nsl < 0) amax = fmin(amax, -sl[i] / nsl);
                if (nsu < 0) amax = fmin(amax, -su[i] / nsu);
                if (nll < 0) amax = fmin(amax, -ll[i] / nll);
                if (nlu < 0) amax = fmin(amax, -lu[i] / nlu);
            }
            if (pass == 0) {
                const double aaff = fmin(1.0, amax);
                double muaff = 0;
                for (int i = 0; i < n; ++i) muaff += (sl[i] + aaff * dsl[i]) * (ll[i] + aaff * dll[i]) + (su[i] + aaff * dsu[i]) * (lu[i] + aaff * dlu[i]);
                muaff /= m;
                const double sg = muaff / mu;
                sigmu = sg * sg * sg * mu;
            } else {
                alpha = fmin(1.0, 0.995 * amax);
            }
        }
        double zm = 1.0, dm = 0.0;
        for (int i = 0; i < n; ++i) {
            zm = fmax(zm, fabs(X[i])); dm = fmax(dm, fabs(alpha * dz[i]));
            X[i] += alpha * dz[i];
            sl[i] += alpha * dsl[i]; su[i] += alpha * dsu[i]; ll[i] += alpha * dll[i]; lu[i] += alpha * dlu[i];
        }
        laststep = alpha >= 0.5 ? dm / zm : 1e300;
    }
    return it;
}

/* Runs `periods` estimator periods of B estimators (preparestate! + updatestate! each), OpenMP over estimators.
 * Model arrays problem-major, column-major inside a problem (the C-ABI layout): A (nx,nx,B), Bu (nx,nu,B), Cm (nym,nx,B),
 * Q (nx,nx,B), R (nym,nym,B), P0 (nx,nx,B); data Y (nym,B,periods), U (nu,B,periods) = u0 applied during period k.
 * Outputs: xhat (nx,B,periods) the estimate x̂0(k) of every period, iters / status (B,periods). */
int mhe_ref_run(int B, int nx, int nu, int nym, int He, const double* A, const double* Bu, const double* Cm, const double* Q,
                const double* R, const double* P0, const double* Y, const double* U, int periods, double xabs, double* xhat,
                int* iters, int* status, int nthreads) {
    if (nx > NXMAX || nym > NXMAX) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int b = 0; b < B; ++b) {
        est_t e;
        e.nx = nx; e.nu = nu; e.nym = nym; e.He = He;
        e.A = A + (size_t)b * nx * nx; e.Bu = Bu + (size_t)b * nx * nu; e.Cm = Cm + (size_t)b * nym * nx;
        memcpy(e.Q, Q + (size_t)b * nx * nx, sizeof(double) * nx * nx);
        memcpy(e.R, R + (size_t)b * nym * nym, sizeof(double) * nym * nym);
        double Qi[NXMAX * NXMAX], Ri[NXMAX * NXMAX], tmp[NXMAX * NXMAX];
        inv_spd(e.Q, Qi, nx);
        inv_spd(e.R, Ri, nym);
        for (int i = 0; i < nx * nx; ++i) e.Qi2[i] = 2.0 * Qi[i];
        /* Oc = -2Q̂⁻¹Â ; T1 = Â' 2Q̂⁻¹ Â ; T2 + T3 = 2Q̂⁻¹ + Ĉm' 2R̂⁻¹ Ĉm */
        for (int j = 0; j < nx; ++j)
            for (int i = 0; i < nx; ++i) {
                double s = 0;
                for (int k = 0; k < nx; ++k) s += AT(e.Qi2, i, k, nx) * AT(e.A, k, j, nx);
                AT(tmp, i, j, nx) = s;
                AT(e.Oc, i, j, nx) = -s;
            }
        for (int j = 0; j < nx; ++j)
            for (int i = 0; i < nx; ++i) {
                double s = 0;
                for (int k = 0; k < nx; ++k) s += AT(e.A, k, i, nx) * AT(tmp, k, j, nx);
                AT(e.T1, i, j, nx) = s;
            }
        for (int a = 0; a < nym; ++a)
            for (int i = 0; i < nx; ++i) {
                double s = 0;
                for (int c = 0; c < nym; ++c) s += e.Cm[c + nym * i] * 2.0 * Ri[c + nym * a];
                e.CtR2[i + nx * a] = s;
            }
        for (int j = 0; j < nx; ++j)
            for (int i = 0; i < nx; ++i) {
                double s = AT(e.Qi2, i, j, nx);
                for (int a = 0; a < nym; ++a) s += e.CtR2[i + nx * a] * e.Cm[a + nym * j];
                AT(e.T2T3, i, j, nx) = s;
            }
        const int S = He + 1, n = S * nx;
        double* work = (double*)malloc(sizeof(double) * ((size_t)16 * n + 2 * (size_t)S * nx * nx));
        double* Yw = (double*)calloc((size_t)He * nym, sizeof(double));
        double* Uw = (double*)calloc((size_t)He * nu, sizeof(double));
        double* Xold = (double*)calloc((size_t)He * nx, sizeof(double));
        double* X = (double*)calloc((size_t)n, sizeof(double));
        double P[NXMAX * NXMAX], Pi2[NXMAX * NXMAX], x0[NXMAX], lastu[NXMAX], xbar[NXMAX];
        memcpy(P, P0 + (size_t)b * nx * nx, sizeof(double) * nx * nx);
        memset(x0, 0, sizeof x0);
        memset(lastu, 0, sizeof lastu);
        int Nk = 0;
        for (int k = 0; k < periods; ++k) {
            const double* yk = Y + ((size_t)k * B + b) * nym;
            const double* uk = U + ((size_t)k * B + b) * nu;
            /* add_data_windows! */
            int moving = 0;
            if (Nk == He) {
                moving = 1;
                memmove(Yw, Yw + nym, sizeof(double) * (He - 1) * nym);
                memmove(Uw, Uw + nu, sizeof(double) * (He - 1) * nu);
                memmove(Xold, Xold + nx, sizeof(double) * (He - 1) * nx);
            } else {
                ++Nk;
            }
            memcpy(Yw + (size_t)(Nk - 1) * nym, yk, sizeof(double) * nym);
            memcpy(Uw + (size_t)(Nk - 1) * nu, lastu, sizeof(double) * nu);
            memcpy(Xold + (size_t)(Nk - 1) * nx, x0, sizeof(double) * nx);
            memcpy(xbar, Xold, sizeof(double) * nx);
            /* correct_cov! (moving window): KalmanFilter correction of P̄ with the oldest measurement */
            if (moving) {
                double M[NXMAX * NXMAX], Mi[NXMAX * NXMAX], PC[NXMAX * NXMAX], K[NXMAX * NXMAX], Pn[NXMAX * NXMAX];
                for (int a = 0; a < nym; ++a)
                    for (int i = 0; i < nx; ++i) {
                        double s = 0;
                        for (int kk = 0; kk < nx; ++kk) s += AT(P, i, kk, nx) * e.Cm[a + nym * kk];
                        PC[i + nx * a] = s;                       /* P Ĉm' */
                    }
                for (int a = 0; a < nym; ++a)
                    for (int c = 0; c < nym; ++c) {
                        double s = e.R[c + nym * a];
                        for (int i = 0; i < nx; ++i) s += e.Cm[c + nym * i] * PC[i + nx * a];
                        M[c + nym * a] = s;
                    }
                inv_spd(M, Mi, nym);
                for (int a = 0; a < nym; ++a)
                    for (int i = 0; i < nx; ++i) {
                        double s = 0;
                        for (int c = 0; c < nym; ++c) s += PC[i + nx * c] * Mi[c + nym * a];
                        K[i + nx * a] = s;
                    }
                for (int j = 0; j < nx; ++j)
                    for (int i = 0; i < nx; ++i) {
                        double s = AT(P, i, j, nx);
                        for (int a = 0; a < nym; ++a) s -= K[i + nx * a] * PC[j + nx * a];   /* (I - K Ĉm) P, P symmetric */
                        AT(Pn, i, j, nx) = s;
                    }
                for (int j = 0; j < nx; ++j)
                    for (int i = 0; i < nx; ++i) AT(P, i, j, nx) = 0.5 * (AT(Pn, i, j, nx) + AT(Pn, j, i, nx));
            }
            inv_spd(P, Pi2, nx);
            for (int i = 0; i < nx * nx; ++i) Pi2[i] *= 2.0;
            /* the QP of the window: cold start at zero like the oracle */
            memset(X, 0, sizeof(double) * (size_t)(Nk + 1) * nx);
            int st = 0;
            const int it = solve_window(&e, Nk, Pi2, xbar, Yw, Uw, xabs, X, &st, work);
            memcpy(x0, X + (size_t)Nk * nx, sizeof(double) * nx);
            memcpy(xhat + ((size_t)k * B + b) * nx, x0, sizeof(double) * nx);
            iters[(size_t)k * B + b] = it;
            status[(size_t)k * B + b] = st;
            /* updatestate!: update_cov! once the window is full, keep u */
            if (Nk == He) {
                double AP[NXMAX * NXMAX];
                for (int j = 0; j < nx; ++j)
                    for (int i = 0; i < nx; ++i) {
                        double s = 0;
                        for (int kk = 0; kk < nx; ++kk) s += AT(e.A, i, kk, nx) * AT(P, kk, j, nx);
                        AT(AP, i, j, nx) = s;
                    }
                for (int j = 0; j < nx; ++j)
                    for (int i = 0; i < nx; ++i) {
                        double s = AT(e.Q, i, j, nx);
                        for (int kk = 0; kk < nx; ++kk) s += AT(AP, i, kk, nx) * AT(e.A, j, kk, nx);
                        AT(P, i, j, nx) = s;
                    }
            }
            memcpy(lastu, uk, sizeof(double) * nu);
        }
        free(work); free(Yw); free(Uw); free(Xold); free(X);
    }
    return 0;
}

int mhe_ref_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
